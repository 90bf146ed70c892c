#!/bin/bash
# developer helper: the GPU suite and the driver's bench line of the final state
out=gpurun_out/${1:-r4last}; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $out/gpu_parity_suite.log 2>&1; tail -3 $out/gpu_parity_suite.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_4k_steps20.json 2> $out/bench.err
python - "$out/bench_4k_steps20.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d['config']
print('value', d['value'], c.get('value_samples_fps'), 'la0', c.get('value_irap_lookahead_0'), 'dev', c['device_only_fps'], 'verified', c['verified_timed_pictures_vs_oracle'])
PY
