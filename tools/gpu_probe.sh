#!/bin/bash
# developer helper: parity suite, then device-only throughput with the chain ended after SAO / after deblocking (what SAO costs)
out=gpurun_out/${1:-probe}; mkdir -p $out
export TMPDIR=/tmp
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for sa in 0 3 2; do
  timeout 240 python bench.py --no-cpu-baseline --verify 0 --stop-after $sa > $out/stop$sa.json 2> $out/stop$sa.err
  python - $out/stop$sa.json $sa <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print('stop_after',sys.argv[2],'value',d['value'],'dev_only',d['config']['device_only_fps'])
except Exception as e: print(sys.argv[1],'ERR',e)
PY
done
PROBE_PICTURES=2 timeout 120 python tools/intra_probe.py 2>&1 | grep -v "vvr\]" | head -3
