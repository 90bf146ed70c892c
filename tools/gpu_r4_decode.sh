#!/bin/bash
# developer helper: the parser-fed bitstreams through the drop-in libvvdec.so on the GPU
out=gpurun_out/${1:-r4dec}; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python tools/dropin_decode.py --dir tests/bitstreams --json $out/dropin_decode.json ${ONLY:+--only $ONLY} > $out/dropin_decode.txt 2>&1; tail -${TAILN:-30} $out/dropin_decode.txt
