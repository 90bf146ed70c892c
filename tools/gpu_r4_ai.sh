#!/bin/bash
# developer helper: the all-intra configuration (product build) and the driver's window after a change of the intra launch
out=gpurun_out/${1:-r4ai}; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "intra or allintra or all_intra or config5 or baseline" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
for st in ${STS:-12}; do
timeout 300 python bench.py --config allintra --no-cpu-baseline --verify 2 --streams $st > $out/ai_$st.json 2> $out/ai_$st.err; python - $out/ai_$st.json $st <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d['config']; r=d['roofline']
print('lanes', sys.argv[2], 'value', d['value'], c.get('value_samples_fps'), 'dev', c['device_only_fps'], 'verified', c['verified_timed_pictures_vs_oracle'], {k:(v['avg_us'],v['launches']) for k,v in r['all_kernels'].items()})
PY
done | tee $out/ai.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/b20.json 2> $out/b20.err; python - $out/b20.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d['config']
print('K20 value', d['value'], c.get('value_samples_fps'), 'dev', c['device_only_fps'], 'verified', c['verified_timed_pictures_vs_oracle'])
PY
