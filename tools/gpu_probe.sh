#!/bin/bash
# developer helper: experiments of one gpurun call
out=gpurun_out/${1:-probe}; mkdir -p $out
export TMPDIR=/tmp
echo "== pytest (new tests)"; timeout 400 python -m pytest tests -m gpu -q -k "slices or golden or ladf or luma_adaptive or declibrecon" > $out/pytest_new.log 2>&1; tail -4 $out/pytest_new.log
echo "== H2D"; timeout 100 python tools/h2d_probe.py 2>&1 | tee $out/h2d.txt
run() { name=$1; shift; timeout 240 "$@" > $out/$name.json 2> $out/$name.err; grep "vvr\]" $out/$name.err; python - $out/$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1],'value',d['value'],'dev_only',d['config']['device_only_fps'],d['config'].get('submit_loop_ms',{}).get('vvr_submit'))
except Exception as e: print(sys.argv[1],'ERR',e)
PY
}
VVDEC_AMD_LIB=$GRAFT_REPO_ROOT/vvdec_amd/libvvdec_amd_wd.so run ra_wd python bench.py --no-cpu-baseline --verify 0
VVDEC_AMD_LIB=$GRAFT_REPO_ROOT/vvdec_amd/libvvdec_amd_wd.so run ra_wd_pageable python bench.py --no-cpu-baseline --verify 0 --pageable-records
run ra_ht12 python bench.py --no-cpu-baseline --verify 0 --host-threads 12
run ra_s6 python bench.py --no-cpu-baseline --verify 0 --streams 6
