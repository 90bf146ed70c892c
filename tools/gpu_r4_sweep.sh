#!/bin/bash
# developer helper (one gpurun call): k_intra of B pictures alone on the device - wavefronts per workgroup x workgroups per launch (dev build)
out=gpurun_out/${1:-r4c}; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for waves in 4 8; do for mult in 2 4 8; do
  echo "== waves $waves mult $mult"
  VVDEC_AMD_LIB=$R/vvdec_amd/libvvdec_amd_dev.so VVR_INTRA_WAVES=$waves VVR_INTRA_WG_MULT=$mult PROBE_PICTURES=3 timeout 200 python tools/intra_probe.py 2>&1 | grep POC | sed 's/mc .*itrans [0-9]*//; s/deblock.*//'
done; done | tee $out/sweep.txt
