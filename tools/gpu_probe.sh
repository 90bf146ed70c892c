#!/bin/bash
# developer helper: device-only throughput with one kernel family left out (developer library, VVR_SKIP_KERNELS): what each costs when the device is busy
out=gpurun_out/${1:-probe}; mkdir -p $out
export TMPDIR=/tmp
export VVDEC_AMD_LIB=$GRAFT_REPO_ROOT/vvdec_amd/libvvdec_amd_wd.so
# kernel ids: 0 mc, 1 mc_dmvr, 2 mc_affine, 3 lmcs, 4 itrans, 5 intra, 6 deblock_v, 7 deblock_h, 8 sao, 9 alf
for m in 0 1 2 4 16 32 192 256 512 7 8; do
  VVR_SKIP_KERNELS=$m timeout 240 python bench.py --no-cpu-baseline --verify 0 --host-threads 8 > $out/skip$m.json 2> $out/skip$m.err
  python - $out/skip$m.json $m <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print('skip',sys.argv[2],'value',d['value'],'dev_only',d['config']['device_only_fps'],'us/pic %.0f'%(1e6/d['config']['device_only_fps']))
except Exception as e: print(sys.argv[1],'ERR',e)
PY
done
