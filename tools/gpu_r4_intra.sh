#!/bin/bash
# developer helper (one gpurun call): GPU parity suite, then the per-block timeline of k_intra and the per-kernel times alone
out=gpurun_out/${1:-r4b}; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log; fi
echo "== intra probe (dev build, trace)"
VVDEC_AMD_LIB=$R/vvdec_amd/libvvdec_amd_dev.so VVR_INTRA_TRACE=1 PROBE_PICTURES=2 timeout 300 python tools/intra_probe.py > $out/probe_trace.txt 2>&1
for p in 0 16; do [ -f gpurun_out/intra_btrace_poc$p.bin ] && { echo "-- POC $p"; python tools/intra_btrace.py $p gpurun_out; } ; done > $out/btrace.txt 2>&1
python tools/intra_trace.py >> $out/btrace.txt 2>&1
rm -f gpurun_out/intra_*poc*.bin
grep -E "POC|blocks [0-9]+:|regular blocks|per unit|kernel span|alive" $out/btrace.txt
echo "== intra probe (product build)"
PROBE_PICTURES=3 timeout 300 python tools/intra_probe.py > $out/probe.txt 2>&1; cat $out/probe.txt
if [ -n "$BENCH" ]; then timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench20.json 2> $out/bench20.err; python - $out/bench20.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d['config']; r=d['roofline']
print('value', d['value'], c.get('value_samples_fps'), 'dev', c['device_only_fps'], 'verified', c['verified_timed_pictures_vs_oracle'], 'dom', r['kernel'], r['frac'])
PY
fi
