#!/bin/bash
# developer helper (one gpurun call): the driver's 20-picture window against the number of library threads (20 pictures = 2.5 / 2 / 1 rounds of 8 / 10 / 20
# workers), and with every kernel launch but k_lmcs skipped (watchdog build, VVR_SKIP_KERNELS): what the host stage alone sustains.  gpurun_out/$1/
out=gpurun_out/${1:-hostprobe}; mkdir -p $out
A="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --verify 0"
for t in 8 10 20; do
  timeout 300 python bench.py $A --host-threads $t > $out/bench_t$t.json 2> $out/bench_t$t.err
done
VVDEC_AMD_LIB=$PWD/vvdec_amd/libvvdec_amd_wd.so VVR_SKIP_KERNELS=8175 timeout 300 python bench.py $A --host-threads 8 > $out/bench_nokernels_t8.json 2> $out/bench_nokernels_t8.err
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']
    print(sys.argv[1].split('/')[-1], 'value', d['value'], c.get('value_samples_fps'), 'la0', c.get('value_irap_lookahead_0'), 'dev', c['device_only_fps'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
grep "per streamed picture" $out/bench_nokernels_t8.err | tail -1
