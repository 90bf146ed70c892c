#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/tl6
VVDEC_AMD_LIB=$PWD/vvdec_amd/libvvdec_amd_wd.so VVR_TIMELINE=1 timeout 300 python bench.py --no-cpu-baseline --verify 0 > gpurun_out/tl6/k64.json 2> gpurun_out/tl6/k64.err
python -c "
import json; d=json.load(open('gpurun_out/tl6/k64.json')); print('K64', d['value'], d['config']['device_only_fps'], d['config']['submit_loop_ms'])"
