#!/bin/bash
export TMPDIR=/tmp
for sl in 24 32 48 32 24; do
timeout 300 python bench.py --no-cpu-baseline --verify 0 --slots $sl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slots $sl K64', d['value'], d['config']['device_only_fps'])"
done
timeout 300 python bench.py --no-cpu-baseline --verify 0 --slots 32 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slots 32 K20', d['value'], d['config']['device_only_fps'])"
