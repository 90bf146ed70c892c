#!/bin/bash
# developer helper: how much of the intra stage's block latency is cross-wavefront barrier time (timing experiment, results wrong)
export TMPDIR=/tmp
for lib in libvvdec_amd.so libvvdec_amd_nobar.so; do
  echo "== $lib"
  VVDEC_AMD_LIB=$GRAFT_REPO_ROOT/vvdec_amd/$lib PROBE_PICTURES=2 timeout 120 python tools/intra_probe.py 2>&1 | grep -v "vvr\]" | head -3
done
