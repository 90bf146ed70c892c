#!/bin/bash
# developer helper (one gpurun call each): the round-5 closing runs with the final library.
#   part a: the GPU parity suite, smoke, the bench lines (driver's arguments with other_configs and the CPU baseline; the default arguments)
#   part b: the evidence of tools/gpu_r5_evidence.sh (rocprof statistics, PMC traffic, kernels alone), SQ counters, the leaf timeline, drop-in decoder rates
out=gpurun_out/${2:-r5final}; mkdir -p $out
R=$GRAFT_REPO_ROOT
if [ "$1" = a ]; then
  [ -n "$SKIP_SUITE" ] || { echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $out/gpu_parity_suite.log
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2; }
  echo "== bench, driver arguments"; t0=$SECONDS; timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_4k_steps20_warmup5.json 2> $out/bench_4k_steps20_warmup5.err; echo "$((SECONDS-t0)) s wall"; cut -c1-600 $out/bench_4k_steps20_warmup5.json
  echo "== bench, default arguments"; t0=$SECONDS; timeout 900 python bench.py > $out/bench_4k_default.json 2> $out/bench_4k_default.err; echo "$((SECONDS-t0)) s wall"; cut -c1-300 $out/bench_4k_default.json
  nproc > $out/host.txt; lscpu | head -20 >> $out/host.txt
else
  bash tools/gpu_r5_evidence.sh $(basename $out) 2>&1 | tail -60
  bash tools/gpu_r5_counters.sh $(basename $out) 4k 2>&1 | tail -25
  bash tools/gpu_r5_leaf.sh $(basename $out) 2>&1 | tail -30
  echo "== drop-in decoder on the 4K stream"; for t in 16 32; do timeout 600 python tools/dropin_4k_rate.py $t 6 > $out/dropin_4k_rate_t$t.json 2>&1; python -c "import json,sys; d=json.load(open('$out/dropin_4k_rate_t$t.json')); print($t, {k:(v or {}).get('pictures_per_s') for k,v in d.items()}, (d['with_read_back'] or {}).get('host_ms_per_picture'))"; done
fi
