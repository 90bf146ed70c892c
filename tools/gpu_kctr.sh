#!/bin/bash
# developer helper (one gpurun call): instruction counters per wavefront of the kernels named in $2 (grep pattern), 4K stream, one picture in flight
out=gpurun_out/${1:-kctr}; pat=${2:-k_mc}; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="--steps 6 --warmup 2 --verify 0 --no-cpu-baseline --streams 1 --host-threads 0 --repeats 1"
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$out/ctr_$tag -o c -- python $R/bench.py $A > /dev/null 2> $R/$out/ctr_$tag.err) || echo "   (set '$set' failed)"
done
python - "$out" "$pat" <<'PY'
import csv, glob, sys, collections, re
out, pat = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict); dur = collections.defaultdict(list)
for f in glob.glob(out + "/ctr_*/**/*counter_collection.csv", recursive=True):
    acc = {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if not re.search(pat, k): continue
        key = (k, row["Counter_Name"]); n, s = acc.get(key, (set(), 0.0)); n.add(row["Dispatch_Id"]); acc[key] = (n, s + float(row["Counter_Value"]))
    for (k, c), (n, s) in acc.items(): res[k][c] = s / max(1, len(n))
for f in glob.glob(out + "/ctr_*/**/*kernel_trace.csv", recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if re.search(pat, k): dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(res.items()):
    w = v.get('SQ_WAVES', 1); d = sorted(dur.get(k, [0]))
    print("%-24s %6.1f us  waves %7.0f  VALU/w %6.0f SALU/w %5.0f LDS/w %5.0f VMEMRD/w %5.1f VMEMWR/w %5.1f wait/w %6.0f active/w %6.0f" % (k[:24], d[len(d) // 2], w, v.get('SQ_INSTS_VALU', 0) / w, v.get('SQ_INSTS_SALU', 0) / w, v.get('SQ_INSTS_LDS', 0) / w, v.get('SQ_INSTS_VMEM_RD', 0) / w, v.get('SQ_INSTS_VMEM_WR', 0) / w, v.get('SQ_WAIT_INST_ANY', 0) / w, v.get('SQ_ACTIVE_INST_ANY', 0) / w))
PY
