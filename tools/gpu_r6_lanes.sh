#!/bin/bash
# developer helper (one gpurun call): three against four lanes on one box, the 20- and the 64-picture window
out=gpurun_out/${1:-r6lanes}; mkdir -p $out
for cfg in "20 5 4" "20 5 3" "64 16 4" "64 16 3" "20 5 4" "20 5 3" "64 16 4" "64 16 3"; do set -- $cfg
  timeout 300 python bench.py --steps $1 --warmup $2 --repeats 7 --verify 0 --no-cpu-baseline --no-other-configs --streams $3 > $out/b.json 2>/dev/null
  python - $out/b.json $1 $3 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print("K %2s lanes %s: value %7.1f %s  device only %7.1f" % (sys.argv[2], sys.argv[3], d["value"], c["value_samples_fps"], c["device_only_fps"]))
PY
done 2>&1 | tee $out/lanes.txt
