"""developer helper: copies what tools/gpu_round3.sh / gpu_round4.sh left in gpurun_out/<tag>/ into profiles/ under the names the documents cite
   (kernel statistics of the driver's command, PMC traffic per configuration, counters of the streaming kernels, the bench lines).
   usage: python tools/collect_profiles.py gpurun_out/r3ev2 round3"""
import glob, json, os, shutil, sys
src, rnd = sys.argv[1], sys.argv[2]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
stats = glob.glob(os.path.join(src, "prof", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(P, rnd + "_kernel_stats.csv"))
configs = {}
for cfg in ("4k", "allintra", "8k"):
    f = os.path.join(src, "pmc_traffic_%s.json" % cfg)
    if os.path.exists(f):
        configs[cfg] = json.load(open(f))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    SRC_HASH = bench.kernel_source_hash()          # the summaries say which kernel sources they were measured with (bench.py: "stale" when the library's differ)
except Exception:
    SRC_HASH = None
sq = os.path.join(src, "sq_counters_4k.json")
if os.path.exists(sq):
    d = json.load(open(sq)); d["kernel_source_hash"] = SRC_HASH
    json.dump(d, open(os.path.join(P, rnd + "_sq_counters.json"), "w"), indent=1)
if configs:
    json.dump({"kernel_source_hash": SRC_HASH, "_note": "HBM-side bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over bench.py (one picture in flight), "
                        "gfx950 correction of the MI355X guide applied (tools/pmc_summary.py); keyed by bench configuration and kernel", "configs": configs},
              open(os.path.join(P, rnd + "_pmc_traffic.json"), "w"), indent=1)
for a, b in (("deblock_counters.json", "_deblock_counters.json"), ("host.txt", "_gpu_box_host.txt"), ("bench_under_rocprof.json", "_bench_under_rocprof.json"),
             ("bench_4k_steps20_warmup5.json", "_bench_steps20_warmup5.json"), ("bench_4k_steps64_warmup16.json", "_bench_steps64_warmup16.json"), ("bench_4k_lf_init_host.json", "_bench_steps20_lf_init_host.json"),
             ("bench_allintra.json", "_bench_allintra.json"), ("bench_8k.json", "_bench_8k.json"), ("gpu_parity_suite.log", "_gpu_parity_suite.log"),
             ("kernels_alone.txt", "_kernels_alone.txt"), ("kernels_alone_rocprof.txt", "_kernels_alone_rocprof.txt"), ("bench_4k_default.json", "_bench_default_steps64.json"), ("intra_block_phases.txt", "_intra_block_phases.txt"), ("dropin_decode.json", "_dropin_decode.json")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(P, rnd + b))
print(sorted(f for f in os.listdir(P) if f.startswith(rnd)))
