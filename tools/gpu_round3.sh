#!/bin/bash
# developer helper (one gpurun call): the round-3 evidence - rocprofv3 kernel statistics of the driver's command, PMC traffic passes per configuration,
# the counters of k_deblock the round-2 verdict asked for, the bench lines of the three configurations.  Everything lands in gpurun_out/$1/
out=gpurun_out/${1:-r3ev}; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== rocprof stats (driver arguments)"; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --verify 0 > $R/$out/bench_under_rocprof.json 2> $R/$out/rocprof.err); find $out/prof -name "*kernel_stats.csv" | head -2
PMCARGS="--steps 8 --warmup 4 --verify 0 --no-cpu-baseline --streams 1 --host-threads 0 --repeats 1"
for cfg in 4k allintra 8k; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $cfg $ctr"; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/$out/pmc_${cfg}_$ctr -o pmc -- python $R/bench.py --config $cfg $PMCARGS > $R/$out/bench_pmc_${cfg}_$ctr.json 2> $R/$out/pmc_${cfg}_$ctr.err)
  done
  f=$(find $out/pmc_${cfg}_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_${cfg}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python tools/pmc_summary.py "$f" "$w" $out/pmc_traffic_$cfg.json "python bench.py --config $cfg $PMCARGS" | grep -E "k_intra|k_mc |k_alf|k_deblock|k_sao|k_itrans" | head -12
done
echo "== k_deblock counters"
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$out/ctr_$tag -o c -- python $R/bench.py --steps 6 --warmup 2 --verify 0 --no-cpu-baseline --streams 1 --host-threads 0 --repeats 1 > /dev/null 2> $R/$out/ctr_$tag.err) || echo "   (set '$set' failed)"
done
python - <<'PY' $out
import csv, glob, sys, json
out = sys.argv[1]
res = {}
for f in glob.glob(out + "/ctr_*/**/*counter_collection.csv", recursive=True):
    acc = {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if not k.startswith("k_deblock") and not k.startswith("k_sao") and not k.startswith("k_lmcs"):
            continue
        key = (k, row["Counter_Name"]); n, s = acc.get(key, (set(), 0.0)); n.add(row["Dispatch_Id"]); acc[key] = (n, s + float(row["Counter_Value"]))
    for (k, c), (n, s) in acc.items():
        res.setdefault(k, {})[c] = round(s / max(1, len(n)), 1)
json.dump({"_note": "rocprofv3 --pmc counter sets, per-dispatch averages (summed over the dimensions rocprofv3 reports), 4K RA stream, one picture in flight", "kernels": res}, open(out + "/deblock_counters.json", "w"), indent=1)
for k, v in sorted(res.items()): print(k, v)
PY
echo "== bench lines"
timeout 400 python bench.py --steps 20 --warmup 5 > $out/bench_4k_steps20_warmup5.json 2> $out/bench_4k.err; tail -c 300 $out/bench_4k_steps20_warmup5.json
timeout 400 python bench.py > $out/bench_4k_steps64_warmup16.json 2> $out/bench_4k_64.err
timeout 500 python bench.py --config allintra --verify 2 > $out/bench_allintra.json 2> $out/bench_allintra.err
timeout 600 python bench.py --config 8k --steps 32 --warmup 8 --verify 1 > $out/bench_8k.json 2> $out/bench_8k.err
for f in $out/bench_4k_steps20_warmup5.json $out/bench_4k_steps64_warmup16.json $out/bench_allintra.json $out/bench_8k.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']; r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'value', d['value'], c.get('value_samples_fps'), 'la0', c.get('value_irap_lookahead_0'), 'dev', c['device_only_fps'], 'verified', c['verified_timed_pictures_vs_oracle'], 'cpu', d.get('cpu_baseline',{}).get('value'), 'dom', r['kernel'], r['frac'], r.get('traffic'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
nproc > $out/host.txt; lscpu | head -20 >> $out/host.txt
