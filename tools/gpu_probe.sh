#!/bin/bash
out=gpurun_out/${1:-probe}; mkdir -p $out
export TMPDIR=/tmp
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; python - $out/$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1],'value',d['value'],'dev_only',d['config']['device_only_fps'],'threads',d['config']['host_threads'],'verified',d['config']['verified_timed_pictures_vs_oracle'])
except Exception as e: print(sys.argv[1],'ERR',e)
PY
}
run k20_t8 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --host-threads 8
run k20_t16 python bench.py --no-cpu-baseline --steps 20 --warmup 5
run k64_t16 python bench.py --no-cpu-baseline
run ai python bench.py --no-cpu-baseline --config allintra --verify 2
PROBE_PICTURES=2 timeout 120 python tools/intra_probe.py 2>&1 | grep -v "vvr\]" | head -3
