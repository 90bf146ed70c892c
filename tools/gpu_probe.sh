#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/tl5; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q -k "stream or golden" 2>&1 | tail -2
for rep in 1 2 3; do for t in 8 16; do
VVDEC_AMD_LIB=$PWD/vvdec_amd/libvvdec_amd_wd.so VVR_TIMELINE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --verify 0 --host-threads $t > $out/k20_t${t}_$rep.json 2> $out/k20_t${t}_$rep.err
python -c "
import json; d=json.load(open('$out/k20_t${t}_$rep.json')); print($t, 'K20', d['value'], d['config']['device_only_fps'])"
done; done
for t in 8 16; do
timeout 300 python bench.py --no-cpu-baseline --verify 0 --host-threads $t > $out/k64_t$t.json 2> $out/k64_t$t.err
python -c "
import json; d=json.load(open('$out/k64_t$t.json')); print($t, 'K64', d['value'], d['config']['device_only_fps'])"
done
timeout 300 python bench.py --config allintra --no-cpu-baseline --verify 0 > $out/ai.json 2> $out/ai.err
python -c "
import json; d=json.load(open('$out/ai.json')); print('allintra', d['value'], d['config']['device_only_fps'])"
