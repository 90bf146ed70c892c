#!/bin/bash
# developer helper (one gpurun call): device-only throughput against the number of pictures in flight (lanes), and the host stage alone (kernels skipped)
out=gpurun_out/${1:-lanes}; mkdir -p $out
A="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --verify 0"
for s in 3 4 6 8; do
  timeout 300 python bench.py $A --streams $s > $out/bench_s$s.json 2> $out/bench_s$s.err
done
VVDEC_AMD_LIB=$PWD/vvdec_amd/libvvdec_amd_wd.so VVR_SKIP_KERNELS=4087 timeout 300 python bench.py $A --host-threads 8 > $out/bench_nokernels_t8.json 2> $out/bench_nokernels_t8.err
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']
    print(sys.argv[1].split('/')[-1], 'value', d['value'], c.get('value_samples_fps'), 'la0', c.get('value_irap_lookahead_0'), 'dev', c['device_only_fps'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
grep "per streamed picture" $out/bench_nokernels_t8.err | tail -1
