#!/bin/bash
export TMPDIR=/tmp
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --verify 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K64', d['value'], d['config']['device_only_fps'])"
done
timeout 300 python bench.py --no-cpu-baseline --verify 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K20', d['value'], d['config']['device_only_fps'])"
