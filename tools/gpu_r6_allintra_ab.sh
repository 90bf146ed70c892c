#!/bin/bash
# developer helper (one gpurun call): the all-intra configuration with the libraries of round 4, round 5 and this tree side by side on ONE box (vvdec_amd/libvvdec_amd_r4.so / _r5.so:
# built from `git archive 9691667^` / `git archive d7b1710`, not tracked), K = 64 and the K = 32 window of config.other_configs: where the round-5 regression of this configuration came from
out=gpurun_out/${1:-r6ab}; mkdir -p $out
for k in "64 16" "32 8"; do set -- $k
  for l in r4 r5 cur; do
    lib=vvdec_amd/libvvdec_amd_$l.so; [ $l = cur ] && lib=vvdec_amd/libvvdec_amd.so
    [ -f $lib ] || continue
    VVDEC_AMD_LIB=$PWD/$lib timeout 300 python bench.py --config allintra --steps $1 --warmup $2 --repeats 3 --verify 0 --no-cpu-baseline --no-other-configs > $out/allintra_${l}_k$1.json 2> $out/allintra_${l}_k$1.err
    python - $out/allintra_${l}_k$1.json $l $1 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("all-intra K=%s library %-3s: through vvr_submit %7.1f %s  device only %7.1f  k_intra avg %s us" % (sys.argv[3], sys.argv[2], d["value"], c.get("value_samples_fps"), c.get("device_only_fps"), (d.get("roofline") or {}).get("avg_launch_us")))
except Exception as e: print(sys.argv[2], "unreadable", e)
PY
  done
done 2>&1 | tee $out/allintra_ab.txt
