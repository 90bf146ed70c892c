#!/bin/bash
# developer helper (one gpurun call): the bench lines - the driver's arguments (with other_configs unless NO_OTHER is set) and the default arguments
out=gpurun_out/${1:-r6bench}; mkdir -p $out
echo "== bench, driver arguments"; t0=$SECONDS; timeout 900 python bench.py --steps 20 --warmup 5 ${NO_OTHER:+--no-other-configs} > $out/bench_4k_steps20_warmup5.json 2> $out/bench_4k_steps20_warmup5.err; echo "$((SECONDS-t0)) s wall"; cut -c1-700 $out/bench_4k_steps20_warmup5.json; tail -3 $out/bench_4k_steps20_warmup5.err
echo "== bench, default arguments"; t0=$SECONDS; timeout 900 python bench.py --no-other-configs --no-cpu-baseline > $out/bench_4k_default.json 2> $out/bench_4k_default.err; echo "$((SECONDS-t0)) s wall"; cut -c1-400 $out/bench_4k_default.json
python - $out <<'PY'
import json,sys
for f in ("bench_4k_steps20_warmup5.json","bench_4k_default.json"):
    try:
        d=json.loads(open(sys.argv[1]+"/"+f).read().strip().splitlines()[-1]); c=d["config"]
        print(f, "value", d["value"], "device_only", c.get("device_only_fps"), "samples", c.get("samples"), "other", {k:(v.get("fps"),v.get("device_only_fps")) for k,v in (c.get("other_configs") or {}).items()})
    except Exception as e: print(f, "unreadable", e)
PY
