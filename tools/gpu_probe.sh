#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_line.json 2> gpurun_out/driver_line.err; tail -c 300 gpurun_out/driver_line.json; python -c "
import json; d=json.load(open('gpurun_out/driver_line.json')); print(); print('value', d['value'], d['ms_per_step'], d['cpu_baseline']['value'], d['roofline']['frac'])"
