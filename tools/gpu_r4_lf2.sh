out=gpurun_out/r4lf2; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "edge_parameters_derived" > $out/lf_tests.log 2>&1; tail -2 $out/lf_tests.log
for m in device host; do
timeout 600 python bench.py --steps 20 --warmup 5 --lf-init $m --no-cpu-baseline > $out/bench_$m.json 2>$out/bench_$m.err
python - <<PY
import json
l = json.loads(open("$out/bench_$m.json").read().strip().splitlines()[-1]); c = l["config"]
print("$m value", l["value"], c.get("value_samples_fps"), "dev", c.get("device_only_fps"))
ks = l["roofline"].get("all_kernels") or {}
print("   ", {k: (v.get("avg_us"), v.get("launches")) for k, v in ks.items()})
PY
done
