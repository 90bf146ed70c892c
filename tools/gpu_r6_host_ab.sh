#!/bin/bash
# developer helper (one gpurun call): library variants of the HOST stage side by side on one box - the driver's 20-picture window (9 windows) and the all-intra configuration
out=gpurun_out/${1:-r6hostab}; mkdir -p $out
R=$GRAFT_REPO_ROOT
for round in 1 2; do for tag in $2; do
  lib=$R/vvdec_amd/libvvdec_amd_$tag.so
  VVDEC_AMD_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --repeats 9 --verify 0 --no-cpu-baseline --no-other-configs > $out/b20_$tag.json 2>/dev/null
  VVDEC_AMD_LIB=$lib timeout 300 python bench.py --config allintra --steps 64 --warmup 16 --repeats 5 --verify 0 --no-cpu-baseline --no-other-configs > $out/bai_$tag.json 2>/dev/null
  python - $out/b20_$tag.json $out/bai_$tag.json $tag <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
e=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); f=e["config"]
print("%s: K=20 value %7.1f %s dev %7.1f | all-intra %7.1f %s dev %7.1f" % (sys.argv[3], d["value"], c["value_samples_fps"], c["device_only_fps"], e["value"], f["value_samples_fps"], f["device_only_fps"]))
PY
done; done 2>&1 | tee $out/host_ab.txt
