#!/usr/bin/env python3
"""bench.py — decoded frames/s of the MI355X-native VVC reconstruction back-end on a synthetic PRE-PARSED stream.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N > 1: launched through torch.distributed.run)
  * a "step" is one pass of the hot path over one picture of the stream (all kernels: MC, residual, deblock, SAO, ALF);
  * workload at N = 1: BASELINE.json configs[1], "3840x2160 10-bit random-access QP32, single MI355X" — a hierarchical-B
    GOP-16 stream of synthetic pre-parsed pictures (tool subset listed in config.tools);
  * inputs (CU/TU records, packed levels, motion field, edge parameters, filter controls) are resident in HBM before the
    timed region starts (vvr_prepare); the timed region is K x vvr_submit_prepared + one sync, bracketed by a barrier and
    torch.cuda.synchronize() on both sides; value = pictures of all ranks / max-over-ranks time;
  * N > 1: the stream shards by closed-GOP segment (each rank reconstructs its own independently decodable segment with its
    own DPB): no data-path collective, "scaling": "weak".
  * roofline: per-kernel durations come from HIP events recorded on the launch streams in a second, identical pass
    (vvr_enable_stats); achieved = algorithmic bytes / duration for the kernel with the largest total time;
  * cpu_baseline: the reference decoder's own reconstruction classes (oracle/_ref, SIMD enabled) when that build is present,
    else the plain-C restatement (oracle/), timed on a bounded sample of the same pictures, one picture per process on all
    host cores (frame-parallel, the same sharding the GPU path uses).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np


# SURVEY.md §8(d) config 2 tool mix (fractions of inter CUs; BDOF / DMVR follow from the reference's own conditions:
# bi-predicted, mirrored POC distances, >= 8x8 and >= 128 samples, merge mode for DMVR)
MIX = dict(p_intra=0.15, p_bi=0.6, p_affine=0.06, p_geo=0.03, p_ciip=0.03, p_sbtmvp=0.03, p_bcw=0.05, p_jccr=0.1, p_cclm=0.10, p_mip=0.05, p_isp=0.05)


def _cpu_worker(args):
    kind, W, H, seed, tools, gop, idx = args
    import refdrv
    from vvdec_amd import synth, stream
    plans, _ = stream.ra_plan(gop + 1, gop=gop, seed_poc0_is_external=False)
    pl = plans[idx % len(plans)]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, **MIX)
    refs = {}
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            if slot not in refs:
                refs[slot] = synth.natural_picture(W, H, seed + 100 + poc)
    t0 = time.perf_counter()
    if kind == "reference":
        r = refdrv.reconstruct(d, refs, flags=refdrv.SIMD)
        dt = r["ms"][7] / 1e3          # stage time only (excludes building the reference's object graph)
    else:
        refdrv.oracle_reconstruct(d, refs)
        dt = time.perf_counter() - t0
    return dt


def cpu_baseline(W, H, seed, tools, gop, budget_s=20.0):
    import refdrv
    from concurrent.futures import ProcessPoolExecutor
    kind = "reference" if refdrv.available() else "port"
    # bounded sample: at most 32 worker processes (more only adds memory-bandwidth contention on the GPU box's host and
    # burns wall-clock), a few pictures each, sized to ~budget_s of wall time after a one-picture calibration
    cores = min(os.cpu_count() or 1, 32)
    t1 = _cpu_worker((kind, W, H, seed, tools, gop, 0))
    per_core = max(1, min(4, int(budget_s / max(4 * t1, 1e-3))))
    n = cores * per_core
    t0 = time.perf_counter()
    import multiprocessing
    with ProcessPoolExecutor(max_workers=cores, mp_context=multiprocessing.get_context("spawn")) as ex:
        times = list(ex.map(_cpu_worker, [(kind, W, H, seed, tools, gop, i) for i in range(n)]))
    wall = time.perf_counter() - t0
    # throughput of the reconstruction stage itself: pictures / (sum of stage times / cores)
    fps = n / (sum(times) / cores)
    return {"value": round(fps, 2), "unit": "frames/s", "cores": cores, "kind": kind,
            "sample": "%d pictures of the same %dx%d stream, one picture per process on %d cores (stage time %.0f ms/picture/core, wall %.1f s)%s" %
                      (n, W, H, cores, 1e3 * sum(times) / n, wall, ", reference classes with SIMD" if kind == "reference" else ", plain-C restatement")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--gop", type=int, default=16)
    ap.add_argument("--streams", type=int, default=8, help="pictures in flight per GPU")
    ap.add_argument("--slots", type=int, default=24, help="DPB slots used round-robin (0: smallest DPB, slots reused at once)")
    ap.add_argument("--intra-period", type=int, default=64, help="an IRAP picture every N pictures (multiple of --gop; 0: only POC 0)")
    ap.add_argument("--irap-lookahead", type=int, default=8, help="IRAP pictures are submitted N pictures ahead of their decoding-order position (they depend on nothing)")
    ap.add_argument("--cold", action="store_true", help="time the first K pictures of the stream (from the IRAP at POC 0, nothing to overlap it with) instead of K pictures of the running stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", type=int, default=2, help="number of pictures re-checked against the CPU oracle after the run")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    backend = os.environ.get("VVR_BENCH_BACKEND", "nccl")     # "gloo": control-flow test of the multi-rank path on a box with fewer GPUs than ranks
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    import vvdec_amd
    from vvdec_amd import abi, synth, stream
    vvdec_amd.lib()
    W, H = a.width, a.height
    tools = (abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST |
             abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE)
    K, Wm = a.steps, a.warmup
    # the stream: W warm-up pictures, then the K timed ones; --cold: the K timed pictures are the first K again (start of a stream)
    total = max(K, Wm) if a.cold else Wm + K
    nframes = ((total - 1 + a.gop - 1) // a.gop) * a.gop + 1
    plans, nslots = stream.ra_plan(nframes, gop=a.gop, seed_poc0_is_external=False, pool=a.slots, intra_period=a.intra_period,
                                   irap_lookahead=0 if a.cold else a.irap_lookahead)   # POC 0 is an I picture
    nslots = max(nslots, a.slots)
    from vvdec_amd import parallel
    seed = parallel.segment_seed(1234, rank)          # every rank reconstructs its own closed-GOP segment (no data-path collective)
    rec = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=a.streams, device=local_rank)
    descs = [synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, **MIX) for pl in plans[:total]]
    prepared = [rec.prepare(d) for d in descs]        # everything resident in HBM from here on
    first = 0 if a.cold else Wm                       # first timed picture (submission order)
    n_irap = sum(1 for pl in plans[first:first + K] if pl.slice_type == abi.SLICE_I)

    def one_pass(i0, n, all_ranks=True):
        # all_ranks=False: a pass that only rank 0 runs (statistics): no collective inside
        rec.sync()
        torch.cuda.synchronize()
        if world > 1 and all_ranks:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(i0, i0 + n):
            rec.submit_prepared(prepared[i])
        rec.sync()
        torch.cuda.synchronize()
        if world > 1 and all_ranks:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    one_pass(0, Wm)                                    # warm-up (untimed): the first W pictures of the stream
    dt = one_pass(first, K)                            # timed: exactly K steps
    if world > 1:
        t = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        # ---- correctness of what was just timed: re-check pictures against the CPU oracle (checker only)
        verified = 0
        if a.verify:
            import refdrv
            cpu = {}
            rec2 = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=1, device=local_rank)
            tail = {pl.slot: pl for pl in plans[first + K - 8:first + K]}           # the last timed pictures still sit in their slots
            timed_out = {slot: rec.read_picture(slot) for slot in tail}
            for i, (pl, d) in enumerate(zip(plans[:first + K], descs)):
                rec2.wait(rec2.decompress_picture(d))
                if i < a.verify:                                                    # against the CPU oracle
                    got = rec2.read_picture(pl.slot)
                    want = refdrv.oracle_reconstruct(d, cpu)
                    assert all(np.array_equal(g, w) for g, w in zip(got, want)), "bench: POC %d differs from the oracle" % pl.poc
                    cpu[pl.slot] = want
                    verified += 1
            for slot, pl in tail.items():                                           # pipelined timed run == one picture at a time
                got = rec2.read_picture(slot)
                assert all(np.array_equal(g, w) for g, w in zip(got, timed_out[slot])), "bench: POC %d differs between the timed run and a serial run" % pl.poc
            rec2.close()
        # ---- roofline of the dominant kernel: second identical pass with HIP-event timing on the launch streams
        rec.enable_stats(True)
        if not a.cold:
            one_pass(0, Wm, all_ranks=False)
        one_pass(first, K, all_ranks=False)
        st = rec.stats()
        rec.enable_stats(False)
        st.sort(key=lambda s: -s["total_ms"])
        dom = st[0]
        for x in st:
            x["total_ms"] = max(x["total_ms"], 1e-6)          # (a kernel that never ran has no time)
        achieved = dom["algo_bytes"] / (dom["total_ms"] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": None,
                "avg_launch_us": round(1e3 * dom["total_ms"] / dom["launches"], 2),
                "all_kernels": {s["name"]: {"avg_us": round(1e3 * s["total_ms"] / s["launches"], 2), "launches": s["launches"],
                                            "algo_GBps": round(s["algo_bytes"] / (s["total_ms"] * 1e-3) / 1e9, 1)} for s in st}}
        # HBM-side traffic of that kernel from the separate PMC passes (profiles/round1_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE /
        # --pmc WRITE_SIZE, gfx950 correction applied); counters cannot be collected inside this run
        try:
            pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round1_pmc_traffic.json")))["kernels"]
            ent = next(v for k, v in pmc.items() if k.split("<")[0] == dom["name"])
            roof["traffic"] = ent["hbm_side_bytes_per_launch_corrected"]
            roof["traffic_source"] = "profiles/round1_pmc_traffic.json (separate rocprofv3 --pmc passes, bytes per launch)"
            roof["algorithmic_bytes_per_launch"] = int(dom["algo_bytes"] / dom["launches"])
        except Exception:
            pass
        fps = world * K / dt
        out = {"metric": "decoded frames/sec (4K 10-bit RA) on MI355X, synthetic pre-parsed stream, bit-exact vs ref",
               "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
               "ms_per_step": round(1e3 * dt / K, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "int16", "data": "synthetic",
               "config": {"workload": "%dx%d 10-bit 4:2:0 random-access QP32 (hierarchical-B GOP %d, IRAP every %d pictures), CTU 128, pre-parsed records resident in HBM; %s"
                                      % (W, H, a.gop, a.intra_period, "the first K pictures of the stream (cold start at the IRAP)" if a.cold else
                                         "K pictures of the running stream after W warm-up pictures (%d IRAP in the timed pictures, submitted %d pictures ahead of its decoding-order position)" % (n_irap, a.irap_lookahead)),
                          "tools": "I picture + hierarchical-B pictures with 15 % intra CUs: intra planar/DC/angular/wide-angle + PDPC + MRL + reference smoothing + BDPCM + ISP, LFNST, inter uni/bi MC (8/4-tap DCTIF, alt half-pel, BCW), BDOF, DMVR, affine 4/6-parameter + PROF, GPM, CIIP, SbTMVP, CCLM/MDLM, MIP, LMCS luma mapping + chroma residual scaling, dequant + dep-quant, DCT2/DST7/DCT8 + transform skip, joint Cb-Cr, deblocking, SAO, ALF + CC-ALF",
                          "mix": MIX,
                          "not_in_mix": "SBT, explicit weighted prediction, scaling lists (implemented and tested, not part of the configuration of SURVEY.md 8(d)); dual-tree I pictures, CUs down to 4x4, local dual trees and IBC are implemented and tested too",
                          "pictures_in_flight": a.streams, "sharding": "closed-GOP segment per GPU, no data-path collective",
                          "verified_pictures_vs_oracle": verified, "timed_run_equals_serial_run": bool(a.verify)},
               "roofline": roof}
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H, seed, tools, a.gop)
    for h in prepared:
        rec.free_prepared(h)
    rec.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
