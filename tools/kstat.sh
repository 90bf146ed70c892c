#!/bin/bash
# developer helper (GPU box): run the short bench with the given environment and print fps + per-kernel averages
python bench.py --no-cpu-baseline --verify 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fps', d['value'], 'ms/frame', d['ms_per_step']); print('  '+'  '.join('%s %.1f' % (k, v['avg_us']) for k, v in d['roofline']['all_kernels'].items()))"
