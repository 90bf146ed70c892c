#!/bin/bash
# developer helper: where the time of the tiled deblocking passes goes (timing switches of the developer build)
out=gpurun_out/${1:-r4dbv}; mkdir -p $out
export TMPDIR=/tmp VVDEC_AMD_LIB=$PWD/vvdec_amd/libvvdec_amd_dev.so
for m in ${DBV_MODES:-0 32 64 96}; do echo "== VVR_DBV_DBG=$m"; VVR_DBV_DBG=$m PROBE_PICTURES=2 timeout 200 python tools/intra_probe.py 2>&1 | grep "POC 16"; done | tee $out/dbv.txt
