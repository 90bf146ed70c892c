#!/bin/bash
out=gpurun_out/${1:-probe}; mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; shift; timeout 240 "$@" > $out/$name.json 2> $out/$name.err; grep "vvr\]" $out/$name.err | grep -v "host time"; python - $out/$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1],'value',d['value'],'dev_only',d['config']['device_only_fps'],d['config'].get('submit_loop_ms',{}).get('vvr_submit'))
except Exception as e: print(sys.argv[1],'ERR',e)
PY
}
export VVDEC_AMD_LIB=$GRAFT_REPO_ROOT/vvdec_amd/libvvdec_amd_wd.so
run ra_wd python bench.py --no-cpu-baseline --verify 0


unset VVDEC_AMD_LIB
for t in 6 8 12; do run ra_ht$t python bench.py --no-cpu-baseline --verify 0 --host-threads $t; done
run ra_driver python bench.py --no-cpu-baseline --verify 0 --steps 20 --warmup 5
run ai python bench.py --no-cpu-baseline --verify 0 --config allintra
