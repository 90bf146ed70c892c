#!/bin/bash
# developer helper (one gpurun call): round-4 starting point - per-block timeline of k_intra (I picture and a B picture), per-kernel times alone, the driver's bench line
out=gpurun_out/${1:-r4a}; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== intra probe (dev build, trace)"
VVDEC_AMD_LIB=$R/vvdec_amd/libvvdec_amd_dev.so VVR_INTRA_TRACE=1 PROBE_PICTURES=3 timeout 300 python tools/intra_probe.py > $out/probe_trace.txt 2>&1
for p in 0 16 8; do [ -f gpurun_out/intra_btrace_poc$p.bin ] && { echo "-- POC $p"; python tools/intra_btrace.py $p gpurun_out; } ; done > $out/btrace.txt 2>&1
python tools/intra_trace.py >> $out/btrace.txt 2>&1
rm -f gpurun_out/intra_*poc*.bin
echo "== intra probe (product build)"
PROBE_PICTURES=5 timeout 300 python tools/intra_probe.py > $out/probe.txt 2>&1; cat $out/probe.txt
echo "== bench"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench20.json 2> $out/bench20.err; tail -c 600 $out/bench20.json
