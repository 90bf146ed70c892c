#!/bin/bash
out=gpurun_out/${1:-probe}; mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; shift; timeout 240 "$@" > $out/$name.json 2> $out/$name.err; grep "vvr\]" $out/$name.err; python - $out/$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1],'value',d['value'],'dev_only',d['config']['device_only_fps'],d['config'].get('submit_loop_ms',{}).get('vvr_submit'))
except Exception as e: print(sys.argv[1],'ERR',e)
PY
}
for r in 12 18 24 36 64; do run ra_ring$r python bench.py --no-cpu-baseline --verify 0 --ring $r; done
run ra_ring36_ht12 python bench.py --no-cpu-baseline --verify 0 --ring 36 --host-threads 12
run ra_ring24_ht4 python bench.py --no-cpu-baseline --verify 0 --ring 24 --host-threads 4
timeout 300 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
