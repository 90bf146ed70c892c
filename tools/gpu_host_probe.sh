#!/bin/bash
# developer helper (one gpurun call): where the headline `value` is bound - the same bench window with 4 / 8 / 12 / 16 library threads, and with
# every kernel launch skipped (watchdog build, VVR_SKIP_KERNELS): what the host stage alone sustains.  Lands in gpurun_out/$1/
out=gpurun_out/${1:-hostprobe}; mkdir -p $out
A="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --verify 0"
for t in 8 16; do
  timeout 300 python bench.py $A --host-threads $t > $out/bench_t$t.json 2> $out/bench_t$t.err
done
for t in 8 16; do
  VVDEC_AMD_LIB=$PWD/vvdec_amd/libvvdec_amd_wd.so VVR_SKIP_KERNELS=4087 timeout 300 python bench.py $A --host-threads $t > $out/bench_nokernels_t$t.json 2> $out/bench_nokernels_t$t.err
done
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']
    print(sys.argv[1].split('/')[-1], 'value', d['value'], c.get('value_samples_fps'), 'la0', c.get('value_irap_lookahead_0'), 'dev', c['device_only_fps'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
