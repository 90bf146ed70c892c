#!/bin/bash
# developer helper (one gpurun call): HBM-side traffic and duration of k_intra_leaf (PMC FETCH_SIZE / WRITE_SIZE passes, one picture in flight), kernels alone, parity of the leaf paths
out=gpurun_out/${1:-r5lt}; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PMCARGS="--steps 8 --warmup 4 --verify 0 --no-cpu-baseline --no-other-configs --streams 1 --host-threads 0 --repeats 1"
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/$out/pmc_4k_$ctr -o pmc -- python $R/bench.py --config 4k $PMCARGS > $R/$out/bench_pmc_4k_$ctr.json 2> $R/$out/pmc_4k_$ctr.err)
done
f=$(find $out/pmc_4k_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_4k_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py "$f" "$w" $out/pmc_traffic_4k.json "python bench.py --config 4k $PMCARGS" | grep -E "k_intra"
echo "== kernels alone"; PROBE_PICTURES=3 timeout 300 python tools/intra_probe.py 2>&1 | tail -2
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for k in 20 64; do echo "== K=$k"; timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-other-configs --verify 1 --repeats 7 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config'].get('device_only_fps'), d['config']['value_samples_fps'])"; done
