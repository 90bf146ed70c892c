#!/bin/bash
# developer helper: LF_INIT on the device (VVR_TOOL_LFP_ON_DEVICE) - its parity tests, the whole GPU suite, the driver's bench line with the edge parameters
# derived on the device and with the host's tables, the parser-fed streams through the drop-in (which leaves LF_INIT to the back-end by default)
out=gpurun_out/${1:-r4lf}; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "edge_parameters_derived" > $out/lf_tests.log 2>&1; tail -5 $out/lf_tests.log
if [ -z "$SKIP_SUITE" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_suite.log 2>&1; tail -4 $out/gpu_suite.log; fi
for m in device host; do
  timeout 600 python bench.py --steps 20 --warmup 5 --lf-init $m > $out/bench_lf_$m.json 2> $out/bench_lf_$m.err
  python - <<PY
import json
try:
    l = json.loads(open("$out/bench_lf_$m.json").read().strip().splitlines()[-1]); c = l["config"]
    print("lf-init $m value", l["value"], c.get("value_samples_fps"), "dev", c.get("device_only_fps"), "K64?", c.get("workload", "")[-120:])
    ks = l["roofline"].get("all_kernels") or {}
    print("   ", {k: (v.get("avg_us"), v.get("launches"), v.get("algo_GBps")) for k, v in ks.items() if k in ("k_lf_init", "k_deblock_v", "k_deblock_h")}, "MB/pic", l["roofline"]["frame_level"]["algorithmic_MB_per_picture"])
except Exception as e:
    print("bench $m:", e); print(open("$out/bench_lf_$m.err").read()[-1500:])
PY
done
timeout 600 python bench.py --steps 64 --warmup 16 --lf-init device > $out/bench_lf_device_k64.json 2>/dev/null; python -c "
import json; l=json.loads(open('$out/bench_lf_device_k64.json').read().strip().splitlines()[-1]); print('K64 device lf', l['value'], l['config'].get('value_samples_fps'), 'dev', l['config'].get('device_only_fps'))"
timeout 900 python tools/dropin_decode.py --dir tests/bitstreams --json $out/dropin_decode.json > $out/dropin_decode.txt 2>&1; tail -6 $out/dropin_decode.txt
