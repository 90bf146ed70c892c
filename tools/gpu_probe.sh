#!/bin/bash
export TMPDIR=/tmp
PROBE_PICTURES=2 timeout 120 python tools/intra_probe.py 2>&1 | grep "^POC" | head -3
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --verify 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K64', d['value'], d['config']['device_only_fps'])"
