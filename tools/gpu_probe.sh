#!/bin/bash
# developer helper: experiments of one gpurun call
out=gpurun_out/${1:-probe}; mkdir -p $out
export TMPDIR=/tmp
echo "== host path on this machine"; timeout 200 python tools/host_path_probe.py 2>&1 | tee $out/host_probe.txt
run() { name=$1; shift; timeout 240 "$@" > $out/$name.json 2> $out/$name.err; python - $out/$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1],'value',d['value'],'dev_only',d['config']['device_only_fps'],d['config'].get('submit_loop_ms',{}).get('vvr_submit'))
except Exception as e: print(sys.argv[1],'ERR',e)
PY
}
for q in 4 8 16; do GPU_MAX_HW_QUEUES=$q run ra_q${q}_s8 python bench.py --no-cpu-baseline --verify 0; done
GPU_MAX_HW_QUEUES=16 run ra_q16_s16 python bench.py --no-cpu-baseline --verify 0 --streams 16 --slots 32
GPU_MAX_HW_QUEUES=24 run ra_q24_s24 python bench.py --no-cpu-baseline --verify 0 --streams 24 --slots 48
GPU_MAX_HW_QUEUES=16 run ra_q16_s8_ht16 python bench.py --no-cpu-baseline --verify 0 --host-threads 16
GPU_MAX_HW_QUEUES=16 run ra_q16_s8_20_5 python bench.py --no-cpu-baseline --verify 0 --steps 20 --warmup 5
GPU_MAX_HW_QUEUES=24 run ai_q24_s24 python bench.py --config allintra --steps 48 --warmup 24 --streams 24 --slots 48 --no-cpu-baseline --verify 0
GPU_MAX_HW_QUEUES=32 run ai_q32_s32 python bench.py --config allintra --steps 64 --warmup 32 --streams 32 --slots 64 --no-cpu-baseline --verify 0
GPU_MAX_HW_QUEUES=16 run ai_q16_s16 python bench.py --config allintra --steps 48 --warmup 16 --streams 16 --slots 32 --no-cpu-baseline --verify 2
