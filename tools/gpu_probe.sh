#!/bin/bash
# developer helper: intra piece length on content with smaller coding units
out=gpurun_out/${1:-probe}; mkdir -p $out
export TMPDIR=/tmp
export VVDEC_AMD_LIB=$GRAFT_REPO_ROOT/vvdec_amd/libvvdec_amd_dev.so
for s in 1.5 2.0; do for c in 100000 64 32 16; do
  echo "== split $s chunk $c"
  PROBE_SPLIT=$s VVR_INTRA_CHUNK=$c PROBE_PICTURES=2 timeout 120 python tools/intra_probe.py 2>&1 | grep -v "vvr\]" | head -3
done; done
