#!/bin/bash
# developer helper (one gpurun call): GPU parity suite, per-kernel times alone, bench at the driver's arguments
out=gpurun_out/${1:-r4f}; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > $out/pytest_gpu.log 2>&1; tail -15 $out/pytest_gpu.log
PROBE_PICTURES=3 timeout 300 python tools/intra_probe.py > $out/probe.txt 2>&1; cat $out/probe.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench20.json 2> $out/bench20.err; python - $out/bench20.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d['config']; r=d['roofline']
print('value', d['value'], c.get('value_samples_fps'), 'dev', c['device_only_fps'], 'verified', c['verified_timed_pictures_vs_oracle'], 'dom', r['kernel'], r['frac'])
print({k:(v['avg_us'],v['launches']) for k,v in r['all_kernels'].items()})
PY
