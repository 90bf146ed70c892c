#!/bin/bash
# developer helper (one gpurun call): poll back-off cap of the leaf kernel's waits (LEAF_SLEEP_MAX 4 / 12 / 32 x 64 cycles) - kernels alone and the window.
# The variants are builds of the library with -DLEAF_SLEEP_MAX=4 / 12 (the Makefile's command line with that define, output vvdec_amd/libvvdec_amd_s4.so / _s12.so); not kept in the tree.
R=$GRAFT_REPO_ROOT
for v in "" _s12 _s4; do
  echo "== libvvdec_amd$v.so: alone"; VVDEC_AMD_LIB=$R/vvdec_amd/libvvdec_amd$v.so PROBE_PICTURES=3 timeout 300 python tools/intra_probe.py 2>&1 | tail -2
  for k in 20 64; do
    echo "== libvvdec_amd$v.so K=$k"; VVDEC_AMD_LIB=$R/vvdec_amd/libvvdec_amd$v.so timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-other-configs --verify 1 --repeats 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], [ (k,v) for k,v in d['config'].items() if 'device_only' in k or 'repeat' in k])"
  done
done
