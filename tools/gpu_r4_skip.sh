#!/bin/bash
# developer helper: the fused SAO + ALF pass with phases left out (timing only)
for m in 1 2 4 8 16 31; do echo "== skip $m"; VVDEC_AMD_LIB=$GRAFT_REPO_ROOT/vvdec_amd/libvvdec_amd_x$m.so PROBE_PICTURES=2 timeout 200 python tools/intra_probe.py 2>&1 | grep "POC 16" | sed 's/mc .*deblock_h [0-9]*//'; done
