#!/bin/bash
# developer helper (one gpurun call): the leaf item list by level against decoding order - k_intra_leaf alone, the window at K=20 / K=64, parity of the leaf paths
out=gpurun_out/${1:-r5lev}; mkdir -p $out
R=$GRAFT_REPO_ROOT
for lv in 0 1; do
  echo "== VVR_LEAF_BY_LEVEL=$lv: kernels alone"; VVR_LEAF_BY_LEVEL=$lv PROBE_PICTURES=3 timeout 300 python tools/intra_probe.py 2>&1 | tail -2
done
echo "== parity (fuzz, leaf tests)"; timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for lv in 0 1; do
  for k in 20 64; do
    echo "== VVR_LEAF_BY_LEVEL=$lv K=$k"; VVR_LEAF_BY_LEVEL=$lv timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-other-configs --verify 0 --repeats 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config'].get('device_only_pictures_per_s'), d['config'].get('repeat_values'))"
  done
done
echo "== timeline by level"; VVDEC_AMD_LIB=$R/vvdec_amd/libvvdec_amd_dev.so VVR_INTRA_TRACE=1 PROBE_PICTURES=2 timeout 300 python tools/intra_probe.py > $out/probe_trace.txt 2>&1
python tools/leaf_trace.py 16 gpurun_out | tee $out/leaf_timeline.txt
rm -f gpurun_out/leaf_*poc*.bin gpurun_out/intra_*poc*.bin
