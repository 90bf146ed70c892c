#!/bin/bash
# developer helper (one gpurun call): the driver's 20-picture window over host threads and lanes (the device got faster in round 6: does the balance move?)
out=gpurun_out/${1:-r6sweep}; mkdir -p $out
for cfg in "8 4" "12 4" "16 4" "8 5" "12 5" "8 3"; do set -- $cfg
  timeout 300 python bench.py --steps 20 --warmup 5 --repeats 7 --verify 0 --no-cpu-baseline --no-other-configs --host-threads $1 --streams $2 > $out/b_t$1_s$2.json 2>/dev/null
  python - $out/b_t$1_s$2.json $1 $2 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print("host threads %2s lanes %s: value %7.1f %s  device only %7.1f" % (sys.argv[2], sys.argv[3], d["value"], c["value_samples_fps"], c["device_only_fps"]))
PY
done 2>&1 | tee $out/sweep.txt
