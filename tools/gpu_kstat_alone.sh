#!/bin/bash
# developer helper (one gpurun call): rocprofv3 kernel trace of the 4K stream with ONE picture in flight - what every kernel takes when it has the GPU to itself
out=gpurun_out/${1:-kalone}; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o k -- python $R/bench.py --steps 16 --warmup 4 --verify 0 --no-cpu-baseline --streams 1 --host-threads 0 --repeats 1 > $R/$out/bench.json 2> $R/$out/err.txt)
python - "$out" <<'PY'
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/prof/**/*kernel_trace.csv',recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].replace('void ','')
    acc[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1])):
    v2=sorted(v); med=v2[len(v2)//2]
    print("%-28s n=%4d median %8.1f us  mean %8.1f  max %8.1f"%(k[:28],len(v),med,sum(v)/len(v),v2[-1]))
PY
