#!/bin/bash
# developer helper: one batch of GPU work per gpurun call (tests, bench lines, rocprof summaries), each step under its own timeout;
# everything lands in gpurun_out/$1/
out=gpurun_out/${1:-run}; mkdir -p $out
export TMPDIR=/tmp
echo "== pytest" ; timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
echo "== bench driver args"; timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_20_5.json 2> $out/bench_20_5.err; tail -c 600 $out/bench_20_5.json
echo "== bench default"; timeout 600 python bench.py --no-cpu-baseline > $out/bench_64_16.json 2> $out/bench_64_16.err; tail -c 300 $out/bench_64_16.json
if [ "$2" != "quick" ]; then
echo "== rocprof stats"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --verify 0 > $GRAFT_REPO_ROOT/$out/bench_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rocprof.err); ls $out/prof | head
echo "== allintra"; timeout 600 python bench.py --config allintra --steps 16 --warmup 4 --no-cpu-baseline --verify 2 > $out/bench_allintra.json 2> $out/bench_allintra.err; tail -c 300 $out/bench_allintra.json
echo "== 8k"; timeout 900 python bench.py --config 8k --steps 16 --warmup 4 --no-cpu-baseline --verify 1 > $out/bench_8k.json 2> $out/bench_8k.err; tail -c 300 $out/bench_8k.json
fi
nproc > $out/nproc.txt; lscpu | head -20 >> $out/nproc.txt; free -g >> $out/nproc.txt
