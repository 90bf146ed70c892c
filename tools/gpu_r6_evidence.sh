#!/bin/bash
# developer helper (one gpurun call): the round-6 evidence - rocprofv3 kernel statistics of the driver's command, PMC traffic passes (4K; all-intra and 8K with ALL_CONFIGS=1),
# SQ counters and the kernels alone.  Everything lands in gpurun_out/$1/; tools/collect_profiles.py copies it to profiles/round6_* and stamps the counter summaries with the
# hash of the kernel sources (bench.py says "stale" when the library was built from other sources).
out=gpurun_out/${1:-r6ev}; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== rocprof stats (driver arguments)"; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --verify 0 > $R/$out/bench_under_rocprof.json 2> $R/$out/rocprof.err); find $out/prof -name "*kernel_stats.csv" | head -2
PMCARGS="--steps 8 --warmup 4 --verify 0 --no-cpu-baseline --no-other-configs --streams 1 --host-threads 0 --repeats 1"
CFGS="4k"; [ -n "$ALL_CONFIGS" ] && CFGS="4k allintra 8k"
for cfg in $CFGS; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $cfg $ctr"; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/$out/pmc_${cfg}_$ctr -o pmc -- python $R/bench.py --config $cfg $PMCARGS > $R/$out/bench_pmc_${cfg}_$ctr.json 2> $R/$out/pmc_${cfg}_$ctr.err)
  done
  f=$(find $out/pmc_${cfg}_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_${cfg}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python tools/pmc_summary.py "$f" "$w" $out/pmc_traffic_$cfg.json "python bench.py --config $cfg $PMCARGS" | grep -E "k_intra|k_mc|k_alf|k_deblock|k_sao|k_itrans|k_lf|k_resi" | head -20
done
echo "== counters"; bash tools/gpu_r5_counters.sh $(basename $out) 4k 2>&1 | tail -22
echo "== kernels alone"; bash tools/gpu_kstat_alone.sh $(basename $out)/alone > $out/kernels_alone_rocprof.txt 2>&1; head -20 $out/kernels_alone_rocprof.txt
nproc > $out/host.txt; lscpu | head -20 >> $out/host.txt
