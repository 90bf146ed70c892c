#!/bin/bash
# developer helper (one gpurun call) at the end of the round: the GPU suite, kernel statistics of the all-intra and 8K configurations, two more lines at the driver's arguments
out=gpurun_out/${1:-final}; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $out/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for cfg in allintra 8k; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$cfg -o bench -- python $R/bench.py --config $cfg --no-cpu-baseline --verify 0 --repeats 2 > $R/$out/bench_under_rocprof_$cfg.json 2> $R/$out/rocprof_$cfg.err)
  f=$(find $out/prof_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_$cfg.csv && head -6 $out/kernel_stats_$cfg.csv | cut -c1-120
done
for k in 1 2; do timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_k20_run$k.json 2>/dev/null; python -c "
import json,sys; d=json.load(open('$out/bench_k20_run$k.json')); print('K20 run $k', d['value'], d['config']['value_samples_fps'], d['config']['device_only_fps'])"; done
