#!/bin/bash
export TMPDIR=/tmp
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
PROBE_PICTURES=2 timeout 120 python tools/intra_probe.py 2>&1 | grep -v "vvr\]" | head -3
timeout 240 python bench.py --no-cpu-baseline --verify 0 --host-threads 8 | python -c "import json,sys; d=json.load(sys.stdin); print('value',d['value'],'dev_only',d['config']['device_only_fps'])"
