#!/bin/bash
# developer helper: host and device timeline of the driver's 20-picture window (watchdog build, VVR_TIMELINE)
out=gpurun_out/${1:-r4tl}; mkdir -p $out
export TMPDIR=/tmp
VVDEC_AMD_LIB=$PWD/vvdec_amd/libvvdec_amd_wd.so VVR_TIMELINE=1 timeout 400 python bench.py --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-other-configs --host-threads ${HT:-8} > $out/bench.json 2> $out/timeline.txt
grep -c "vvr timeline" $out/timeline.txt; python - $out/bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d['config']
print('value', d['value'], c.get('value_samples_fps'), 'dev', c['device_only_fps'])
PY
