#!/bin/bash
# developer helper (one gpurun call): where the driver's 20-picture window goes with the round-6 kernels - host and device timeline (watchdog build), then the window
# over host threads and lanes, then the kernels alone
out=gpurun_out/${1:-r6win}; mkdir -p $out
export TMPDIR=/tmp
bash tools/gpu_r4_timeline.sh $(basename $out)/tl
bash tools/gpu_r6_sweep.sh $(basename $out)/sweep
timeout 300 python bench.py --config 4k --steps 64 --warmup 16 --verify 0 --no-cpu-baseline --no-other-configs --repeats 3 > $out/bench_k64.json 2>/dev/null
python - $out/bench_k64.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print("K=64: value %.1f %s device only %.1f" % (d["value"], c["value_samples_fps"], c["device_only_fps"]))
PY
