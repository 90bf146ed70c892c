#!/bin/bash
# developer helper: one batch of GPU work per gpurun call (tests, bench lines, rocprof summaries), each step under its own timeout;
# everything lands in gpurun_out/$1/
out=gpurun_out/${1:-run}; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest" ; timeout 420 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
echo "== bench driver args"; timeout 300 python bench.py --steps 20 --warmup 5 > $out/bench_20_5.json 2> $out/bench_20_5.err; tail -c 400 $out/bench_20_5.json
echo "== bench default"; timeout 300 python bench.py --no-cpu-baseline > $out/bench_64_16.json 2> $out/bench_64_16.err; tail -c 200 $out/bench_64_16.json
if [ "$2" == "mid" ]; then
echo "== allintra"; timeout 300 python bench.py --config allintra --steps 32 --warmup 8 --no-cpu-baseline --verify 2 > $out/bench_allintra.json 2> $out/bench_allintra.err
echo "== allintra 24 streams"; timeout 300 python bench.py --config allintra --steps 48 --warmup 24 --streams 24 --slots 48 --no-cpu-baseline --verify 0 > $out/bench_allintra_s24.json 2> $out/bench_allintra_s24.err
fi
if [ "$2" == "full" ]; then
echo "== bench 8 host threads"; timeout 300 python bench.py --no-cpu-baseline --verify 0 --host-threads 8 > $out/bench_64_16_ht8.json 2> $out/bench_64_16_ht8.err; echo "== driver args 8 threads"; timeout 300 python bench.py --no-cpu-baseline --verify 0 --steps 20 --warmup 5 --host-threads 8 > $out/bench_20_5_ht8.json 2> $out/bench_20_5_ht8.err
echo "== bench pageable records"; timeout 300 python bench.py --no-cpu-baseline --verify 0 --pageable-records > $out/bench_64_16_pageable.json 2> $out/bench_64_16_pageable.err
echo "== rocprof stats"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --no-cpu-baseline --verify 0 > $R/$out/bench_rocprof.json 2> $R/$out/rocprof.err); ls $out/prof | head
for ctr in FETCH_SIZE WRITE_SIZE; do
echo "== pmc $ctr"; (cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/$out/pmc_$ctr -o pmc -- python $R/bench.py --steps 8 --warmup 4 --verify 0 --no-cpu-baseline --streams 1 --host-threads 0 > $R/$out/bench_pmc_$ctr.json 2> $R/$out/pmc_$ctr.err); find $out/pmc_$ctr -name "*counter_collection.csv" | head -2
done
echo "== allintra"; timeout 300 python bench.py --config allintra --no-cpu-baseline --verify 2 > $out/bench_allintra.json 2> $out/bench_allintra.err
echo "== allintra 24 streams"; timeout 300 python bench.py --config allintra --steps 48 --warmup 24 --streams 24 --slots 48 --no-cpu-baseline --verify 0 > $out/bench_allintra_s24.json 2> $out/bench_allintra_s24.err
echo "== 8k"; timeout 420 python bench.py --config 8k --steps 32 --warmup 8 --no-cpu-baseline --verify 1 > $out/bench_8k.json 2> $out/bench_8k.err
fi
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(d["config"].get("submit_loop_ms"))
    print(sys.argv[1].split('/')[-1], 'value',d['value'],'dev_only',d['config']['device_only_fps'],'irap',d['config']['irap_in_window'],'verified',d['config']['verified_timed_pictures_vs_oracle'],'peak_meas',r['peak_measured'], 'cpu', d.get('cpu_baseline',{}).get('value'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
nproc > $out/nproc.txt; lscpu | head -20 >> $out/nproc.txt; free -g >> $out/nproc.txt
