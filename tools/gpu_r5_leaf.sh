#!/bin/bash
# developer helper (one gpurun call): k_intra_leaf alone on the device - product build, the timing switches of the developer build, its per-item timeline
out=gpurun_out/${1:-r5leaf}; mkdir -p $out
R=$GRAFT_REPO_ROOT
echo "== product build"; PROBE_PICTURES=3 timeout 300 python tools/intra_probe.py 2>&1 | tail -2
for dbg in 1 2 3; do echo "== developer build, VVR_LEAF_DBG=$dbg (1: no waits, 2: no ticket)"; VVDEC_AMD_LIB=$R/vvdec_amd/libvvdec_amd_dev.so VVR_LEAF_DBG=$dbg PROBE_PICTURES=2 timeout 300 python tools/intra_probe.py 2>&1 | tail -1; done
echo "== timeline"; VVDEC_AMD_LIB=$R/vvdec_amd/libvvdec_amd_dev.so VVR_INTRA_TRACE=1 PROBE_PICTURES=2 timeout 300 python tools/intra_probe.py > $out/probe_trace.txt 2>&1
python tools/leaf_trace.py 16 gpurun_out | tee $out/leaf_timeline.txt
mv gpurun_out/leaf_*poc16.bin $out/ 2>/dev/null; rm -f gpurun_out/leaf_*poc*.bin gpurun_out/intra_*poc*.bin
