#!/bin/bash
# developer scratch: the quick GPU check between changes (one gpurun call): parity subset, the driver's bench line, the all-intra line
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "stream or golden" 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --verify 2 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K20', d['value'], d['config']['device_only_fps'], d['config']['verified_timed_pictures_vs_oracle'])"
timeout 300 python bench.py --no-cpu-baseline --verify 0 --config allintra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('allintra', d['value'], d['config']['device_only_fps'])"
