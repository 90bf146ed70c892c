#!/bin/bash
# developer helper: the driver's window and K = 64 with the sub-block vectors of affine CUs spanned on the device or supplied by the host (edge parameters on the device in both)
out=gpurun_out/${1:-r4aff}; mkdir -p $out; export TMPDIR=/tmp
for rep in 1 2; do for m in host device; do for k in "20 5" "64 16"; do set -- $k
timeout 600 python bench.py --steps $1 --warmup $2 --affine-mv $m --no-cpu-baseline --verify 2 > $out/b_${m}_$1_$rep.json 2>/dev/null
python - <<PY
import json
l = json.loads(open("$out/b_${m}_$1_$rep.json").read().strip().splitlines()[-1]); c = l["config"]
ks = l["roofline"].get("all_kernels") or {}
print("affine-mv $m K $1 value", l["value"], c.get("value_samples_fps"), "dev", c.get("device_only_fps"), "verified", c.get("verified_timed_pictures_vs_oracle"), {k: v.get("avg_us") for k, v in ks.items() if k in ("k_mc_affine", "k_lf_init")})
PY
done; done; done
