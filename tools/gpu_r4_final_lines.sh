#!/bin/bash
# developer helper: the bench lines of the final defaults (edge parameters and affine sub-block vectors derived on the device)
out=gpurun_out/${1:-r4fin}; mkdir -p $out; export TMPDIR=/tmp
timeout 400 python bench.py --steps 20 --warmup 5 > $out/bench_4k_steps20_warmup5.json 2> $out/bench_4k.err
timeout 400 python bench.py --no-cpu-baseline > $out/bench_4k_steps64_warmup16.json 2> $out/bench_4k_64.err
timeout 600 python bench.py --config 8k --steps 32 --warmup 8 --verify 1 --no-cpu-baseline > $out/bench_8k.json 2> $out/bench_8k.err
for f in $out/bench_4k_steps20_warmup5.json $out/bench_4k_steps64_warmup16.json $out/bench_8k.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']; r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'value', d['value'], c.get('value_samples_fps'), 'la0', c.get('value_irap_lookahead_0'), 'dev', c['device_only_fps'], 'verified', c['verified_timed_pictures_vs_oracle'], 'cpu', d.get('cpu_baseline',{}).get('value'), 'dropin', d.get('cpu_baseline',{}).get('dropin_host_ms_per_picture',{}) and {k: d['cpu_baseline']['dropin_host_ms_per_picture'].get(k) for k in ('lf_init','flatten','planes_back','with_the_reference_lf_init_instead')})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
