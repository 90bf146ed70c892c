#!/bin/bash
# developer helper: the driver's window against the number of host worker threads and the band-building policy (VVR_PARTS)
out=gpurun_out/${1:-r4ht}; mkdir -p $out
export TMPDIR=/tmp
for parts in ${PARTS:-0 2 1}; do for ht in ${HTS:-8 16 24}; do for k in ${KS:-20}; do
VVR_PARTS=$parts timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --host-threads $ht > $out/b${parts}_${ht}_$k.json 2> $out/b${parts}_${ht}_$k.err; python - $out/b${parts}_${ht}_$k.json $parts $ht $k <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d['config']
print('parts', sys.argv[2], 'host threads', sys.argv[3], 'K', sys.argv[4], 'value', d['value'], c.get('value_samples_fps'), 'dev', c['device_only_fps'], 'la0', c.get('value_irap_lookahead_0'), 'verified', c['verified_timed_pictures_vs_oracle'])
PY
done; done; done | tee $out/ht.txt
