#!/bin/bash
# developer helper (one gpurun call): GPU parity suite, then every kernel alone on the device (rocprofv3 kernel trace, one picture in flight) and the SQ counters per kernel
out=gpurun_out/${1:-r6mc}; mkdir -p $out
R=$GRAFT_REPO_ROOT
if [ -z "$SKIP_SUITE" ]; then echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -q ${SUITE_ARGS} 2>&1 | tail -15 | tee $out/gpu_parity_suite.log; fi
echo "== kernels alone"; bash tools/gpu_kstat_alone.sh $(basename $out)/alone 2>&1 | head -24 | tee $out/kernels_alone_rocprof.txt
echo "== counters"; bash tools/gpu_r5_counters.sh $(basename $out) 4k 2>&1 | tail -25
if [ -n "$WITH_BENCH" ]; then echo "== bench, driver arguments"; timeout 900 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > $out/bench_4k_steps20_warmup5.json 2> $out/bench.err; cut -c1-400 $out/bench_4k_steps20_warmup5.json; fi
