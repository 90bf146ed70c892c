#!/bin/bash
# developer helper (one gpurun call): library variants side by side on one box (built with other -D switches: vvdec_amd/libvvdec_amd_<tag>.so) - the kernels alone, then the
# 64-picture window and the device pipeline alone.   usage: tools/gpu_r6_variants.sh <out> "<tag> <tag> ..." ["<kernel name prefix to print>"]
out=gpurun_out/${1:-r6var}; mkdir -p $out
R=$GRAFT_REPO_ROOT
for tag in "" $2; do
  lib=$R/vvdec_amd/libvvdec_amd${tag:+_$tag}.so; name=${tag:-product}
  echo "== $name"
  VVDEC_AMD_LIB=$lib bash tools/gpu_kstat_alone.sh $(basename $out)/alone_$name 2>&1 | grep -i "${3:-k_}" | head -${4:-6}
  VVDEC_AMD_LIB=$lib timeout 300 python bench.py --config 4k --steps 64 --warmup 16 --verify 0 --no-cpu-baseline --no-other-configs --repeats 3 > $out/bench_k64_$name.json 2>/dev/null
  python - $out/bench_k64_$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print("   K=64: value %.1f %s device only %.1f" % (d["value"], c["value_samples_fps"], c["device_only_fps"]))
PY
done 2>&1 | tee $out/variants.txt
