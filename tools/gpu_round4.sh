#!/bin/bash
# developer helper (one gpurun call): the round-4 evidence - rocprofv3 kernel statistics of the driver's command, PMC traffic passes per configuration,
# the counters of k_deblock the round-2 verdict asked for, the bench lines of the three configurations.  Everything lands in gpurun_out/$1/
out=gpurun_out/${1:-r4ev}; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== rocprof stats (driver arguments)"; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --verify 0 > $R/$out/bench_under_rocprof.json 2> $R/$out/rocprof.err); find $out/prof -name "*kernel_stats.csv" | head -2
PMCARGS="--steps 8 --warmup 4 --verify 0 --no-cpu-baseline --streams 1 --host-threads 0 --repeats 1"
for cfg in 4k allintra 8k; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $cfg $ctr"; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/$out/pmc_${cfg}_$ctr -o pmc -- python $R/bench.py --config $cfg $PMCARGS > $R/$out/bench_pmc_${cfg}_$ctr.json 2> $R/$out/pmc_${cfg}_$ctr.err)
  done
  f=$(find $out/pmc_${cfg}_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_${cfg}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python tools/pmc_summary.py "$f" "$w" $out/pmc_traffic_$cfg.json "python bench.py --config $cfg $PMCARGS" | grep -E "k_intra|k_mc |k_alf|k_deblock|k_sao|k_itrans|k_lf" | head -14
done
echo "== GPU suite"; timeout 900 python -m pytest tests -m gpu -q > $out/gpu_parity_suite.log 2>&1; tail -3 $out/gpu_parity_suite.log
echo "== kernels alone"; PROBE_PICTURES=3 timeout 300 python tools/intra_probe.py > $out/kernels_alone.txt 2>&1; cat $out/kernels_alone.txt
echo "== intra block phases (developer build, in-kernel timeline)"
VVDEC_AMD_LIB=$R/vvdec_amd/libvvdec_amd_dev.so VVR_INTRA_TRACE=1 PROBE_PICTURES=2 timeout 300 python tools/intra_probe.py > $out/probe_trace.txt 2>&1
for p in 0 16; do [ -f gpurun_out/intra_btrace_poc$p.bin ] && { echo "-- POC $p"; python tools/intra_btrace.py $p gpurun_out; } ; done > $out/intra_block_phases.txt 2>&1
python tools/intra_trace.py >> $out/intra_block_phases.txt 2>&1; rm -f gpurun_out/intra_*poc*.bin; grep -E "POC|blocks [0-9]+:|regular blocks|kernel span" $out/intra_block_phases.txt | head -12
echo "== drop-in libvvdec.so on the parser-fed bitstreams"; timeout 600 python tools/dropin_decode.py --dir tests/bitstreams --json $out/dropin_decode.json > $out/dropin_decode.txt 2>&1; tail -4 $out/dropin_decode.txt
echo "== bench lines"
timeout 400 python bench.py --steps 20 --warmup 5 > $out/bench_4k_steps20_warmup5.json 2> $out/bench_4k.err; tail -c 300 $out/bench_4k_steps20_warmup5.json
timeout 400 python bench.py > $out/bench_4k_steps64_warmup16.json 2> $out/bench_4k_64.err
timeout 400 python bench.py --steps 20 --warmup 5 --lf-init host --no-cpu-baseline > $out/bench_4k_lf_init_host.json 2> $out/bench_4k_lfhost.err
timeout 500 python bench.py --config allintra --verify 2 > $out/bench_allintra.json 2> $out/bench_allintra.err
timeout 600 python bench.py --config 8k --steps 32 --warmup 8 --verify 1 > $out/bench_8k.json 2> $out/bench_8k.err
for f in $out/bench_4k_steps20_warmup5.json $out/bench_4k_steps64_warmup16.json $out/bench_4k_lf_init_host.json $out/bench_allintra.json $out/bench_8k.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']; r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'value', d['value'], c.get('value_samples_fps'), 'la0', c.get('value_irap_lookahead_0'), 'dev', c['device_only_fps'], 'verified', c['verified_timed_pictures_vs_oracle'], 'cpu', d.get('cpu_baseline',{}).get('value'), 'dom', r['kernel'], r['frac'], r.get('traffic'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
nproc > $out/host.txt; lscpu | head -20 >> $out/host.txt
