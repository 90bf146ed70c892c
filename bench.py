#!/usr/bin/env python3
"""bench.py — decoded frames/s of the MI355X-native VVC reconstruction back-end on a synthetic PRE-PARSED stream.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N > 1: launched through torch.distributed.run)
  * a "step" is one pass of the hot path over one picture of the stream (all kernels: MC, residual, intra, deblock, SAO, ALF);
  * workload at N = 1 (--config 4k, the default): BASELINE.json configs[1], "3840x2160 10-bit random-access QP32, single MI355X" — a
    hierarchical-B GOP-32 stream (six temporal layers) of synthetic pre-parsed pictures with an IRAP every 64 pictures;
    --config 8k = configs[2] (7680x4320 RA QP27), --config allintra = configs[4] (4K all-intra QP22, dual tree);
  * WHAT IS TIMED (`value`): the C-ABI path a decoder uses.  The timed region is K x vvr_submit(host records) + one vvr_sync,
    bracketed by a barrier and torch.cuda.synchronize() on both sides: validation, the host glue that turns the records into
    device work lists (--host-threads worker threads inside the library, default 8, pinned to the GPU's NUMA node), the H2D copy of every picture (pinned ring,
    async) and all kernels.  The records sit in host memory when the region starts — what a CABAC parser leaves behind.
    `config.device_only_fps` is a second timed pass over the same K pictures with the records and work lists already resident in
    HBM (vvr_prepare / vvr_submit_prepared): the number the device pipeline alone sustains.
  * the stream runs P pre-roll pictures (untimed: they only build the DPB the window's pictures reference), W warm-up pictures,
    then the K timed ones; P is chosen so that the timed window holds IRAP pictures in (at least) stream proportion,
    max(1, round(K / intra period)) of them, whatever K is; an IRAP is handed to the back-end --irap-lookahead pictures ahead of its
    decoding-order position (it depends on nothing; a host that parses ahead does the same);
  * a K-picture window is 10-20 ms of a pipeline with host threads in it, so the whole sequence - stream from its first picture, pre-roll,
    warm-up, K timed pictures - is run --repeats times (default 5) and `value` is the MEDIAN of the K-picture times (all samples, minimum and
    maximum are in `config`); `config.value_irap_lookahead_0` is the same stream submitted in plain decoding order (no look-ahead);
  * N > 1: `value` = ONE stream sharded by PICTURE over the GPUs (north_star's split: pictures round-robin within their temporal layer, reference
    pictures over RCCL / xGMI to the ranks that predict from them); a step is one picture per GPU, the timed window holds K x N pictures of the stream,
    "scaling": "weak"; value = those pictures / max-over-ranks time.  The closed-GOP segment mode (each rank its own independently decodable segment
    and DPB, no data-path collective) is measured in the same run and reported as `value_segment_mode`;
  * verification (rank 0, after the timed passes): the last timed run's last pictures and the IRAP of its window == a one-picture-at-a-time
    run, and the IRAP + --verify of those last pictures == the CPU oracle fed with the same reference pictures (checker only, never timed or shipped);
  * roofline: per-kernel durations from HIP events recorded on the launch streams in a further pass over the K timed pictures only
    (vvr_enable_stats; resident records so that launches are back to back); achieved = algorithmic bytes / duration
    for the kernel with the largest total time; `peak_measured` = the library's copy kernel over one DPB slot in the same run;
  * cpu_baseline: the reference's OWN multithreaded reconstruction (oracle/_ref, SIMD enabled) when that build is present - DecLibRecon's per-picture
    set-up and its CTU task state machine on its ThreadPool, 64 threads and all hardware threads, wall clock per picture (cpu_reference_threaded) -,
    with the frame-parallel figure beside it (one picture per process, no scheduler: pictures / (summed stage time / processes)); without that
    build the plain-C restatement (oracle/), one picture per process;
  * N > 1: the picture-mode pass runs in child processes (it has never run on two devices: a fault in it must not sink the line); both modes are
    first-class fields `value_picture_mode` / `value_segment_mode` (fps, pictures, ceiling of the reference graph for the window and for the open
    stream, MB sent per picture, ranks); if the pass does not come back `value` is the segment mode, config.value_is says so, the line carries
    "timeout": true and the process ends with a non-zero status;
  * N = 1, --config 4k (the driver's command): behind the headline configuration the 8K (configs[2]) and all-intra (configs[4]) configurations run in short
    windows, each in a process of its own, and land in config.other_configs (fps, device only, pictures verified against the oracle); roofline carries the
    dominant kernel, the dominant kernel of the B pictures and the I picture's intra kernel apart, and the VALU-issue utilisation from the counter pass.
"""
import argparse
import ctypes as C
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

# Pictures in flight run on HIP streams of their own; the ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4),
# and kernels of streams that share a queue do not overlap.  One queue per lane (the variable is read when HIP initialises: before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


# SURVEY.md §8(d) tool mixes (fractions of inter CUs; BDOF / DMVR follow from the reference's own conditions:
# bi-predicted, mirrored POC distances, >= 8x8 and >= 128 samples, merge mode for DMVR)
MIX = dict(p_intra=0.15, p_bi=0.6, p_affine=0.06, p_geo=0.03, p_ciip=0.03, p_sbtmvp=0.03, p_bcw=0.05, p_jccr=0.1, p_cclm=0.10, p_mip=0.05, p_isp=0.05)
CONFIGS = {
    # name: (width, height, generator parameters, intra period, description)
    "4k": (3840, 2160, dict(MIX, base_qp=32), 64, "BASELINE configs[1]: 3840x2160 10-bit 4:2:0 random-access QP32"),
    "8k": (7680, 4320, dict(MIX, base_qp=27, p_coded=0.5), 64, "BASELINE configs[2]: 7680x4320 10-bit 4:2:0 random-access QP27"),
    "allintra": (3840, 2160, dict(p_intra=1.0, base_qp=22, p_coded=0.7, p_coded_chroma=0.5, p_small_corner=0.5, p_lfnst=0.4, p_isp=0.10, p_mip=0.10, p_cclm=0.10, p_jccr=0.1, dual_tree=1.0), 1,
                 "BASELINE configs[4]: 3840x2160 10-bit 4:2:0 all-intra QP22 (every picture an I picture, dual tree)"),
}


def _tools(abi):
    return (abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST |
            abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE)


def _window(plans, order, K, Wm, n_irap):
    """index (into `order`) of the first timed picture: the window [first, first + K) holds n_irap IRAP pictures, centred on them"""
    from vvdec_amd import abi
    iraps = [k for k, i in enumerate(order) if plans[i].slice_type == abi.SLICE_I and k > Wm]
    lo, hi = iraps[0], iraps[n_irap - 1]
    first = max(Wm, min(lo - 1, (lo + hi) // 2 - K // 2))
    first = max(first, hi - K + 1)
    got = sum(1 for k in iraps if first <= k < first + K)
    assert got >= n_irap and first + K <= len(order), (first, K, iraps[:4], len(order))
    return first


def stream_plan(cfg_name, gop, intra_period, irap_lookahead, slots, K, Wm):
    """-> (plans in DECODING order with their DPB slots, number of slots, {look-ahead: (submission order = indices into plans, index of the
    first timed picture in that order)}).  The same picture descriptions serve every submission order (vvdec_amd.stream.submission_order)."""
    from vvdec_amd import abi, stream
    if cfg_name == "allintra":
        pool = max(slots, 2)
        plans = [stream.PicPlan(poc=i, layer=0, slice_type=abi.SLICE_I, slot=i % pool, ref_slots=([], [])) for i in range(Wm + K)]
        return plans, pool, {irap_lookahead: (list(range(Wm + K)), Wm), 0: (list(range(Wm + K)), Wm)}
    n_irap = max(1, int(round(K / float(intra_period))))
    # enough stream for: a first intra period (pre-roll), the window, the look-ahead
    nframes = intra_period * (n_irap + 2) + K + Wm + gop
    nframes = ((nframes - 1 + gop - 1) // gop) * gop + 1
    plans, nslots = stream.ra_plan(nframes, gop=gop, seed_poc0_is_external=False, pool=slots, intra_period=intra_period)
    orders = {}
    for la in sorted({irap_lookahead, 0}):
        order = stream.submission_order(plans, la)
        first = _window(plans, order, K, Wm, n_irap)
        orders[la] = (order[:first + K], first)
    return plans, max(nslots, slots), orders


def _cpu_worker(args):
    kind, cfg_name, seed, gop, idx = args
    import refdrv
    from vvdec_amd import abi, synth, stream
    W, H, mix, _, _ = CONFIGS[cfg_name]
    tools = _tools(abi)
    if cfg_name == "allintra":
        pl = stream.PicPlan(poc=idx, layer=0, slice_type=abi.SLICE_I, slot=0, ref_slots=([], []))
    else:
        plans, _ = stream.ra_plan(gop + 1, gop=gop, seed_poc0_is_external=False)
        # the pictures of the stream in its proportions: one IRAP (the I picture of the plan) per intra period, B pictures of one GOP otherwise
        pl = plans[0] if idx % CONFIGS[cfg_name][3] == CONFIGS[cfg_name][3] - 1 else plans[1 + idx % (len(plans) - 1)]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, **mix)
    refs = {}
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            if slot not in refs:
                refs[slot] = synth.natural_picture(W, H, seed + 100 + poc)
    t0 = time.perf_counter()
    if kind == "reference":
        r = refdrv.reconstruct(d, refs, flags=refdrv.SIMD)
        dt = r["ms"][7] / 1e3          # stage time only (excludes building the reference's object graph from the records)
    else:
        refdrv.oracle_reconstruct(d, refs)
        dt = time.perf_counter() - t0
    return dt


def dropin_host_cost(cfg_name, seed, gop, threads=8):
    """what the reference-side half of the DROP-IN costs per picture (integration/DecLibReconDropIn.cpp: LF_INIT by the reference's own
    LoopFilter::calcFilterStrengthsCTU over the decoder's thread pool, flattening the reference's objects into a vvr_picture, the planes back into
    the Picture's buffers) - one B picture of the stream through vvdec::DecLibRecon with this library behind it, in a process of its own.  Part of
    the cpu_baseline leg (it runs the reference's classes from oracle/_ref); None where that build is absent."""
    import subprocess
    code = (
        "import os, sys, re\n"
        "sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "os.environ['VVDEC_AMD_TIMES'] = '1'\n"
        "import torch, refdrv, vvdec_amd, bench\n"
        "from vvdec_amd import abi, synth, stream\n"
        "assert refdrv.available() and refdrv.dropin_available()\n"
        "W, H, mix, _, _ = bench.CONFIGS[%r]\n"
        "plans, _ = stream.ra_plan(%d + 1, gop=%d, seed_poc0_is_external=False)\n"
        "pl = plans[2]\n"
        "d = synth.picture_for_plan(pl, W, H, seed=%d, tool_flags=bench._tools(abi), **mix)\n"
        "refs = {slot: synth.natural_picture(W, H, %d + 100 + poc) for lst in pl.ref_slots for (slot, poc) in lst}\n"
        "for k in range(3):\n"
        "    refdrv.run_dropin(d, refs, vvdec_amd._LIBPATH, threads=%d)\n"
    ) % (ROOT, ROOT, cfg_name, gop, gop, seed, seed, threads)
    try:
        pat = r"host ms per picture: MIDER ([0-9.]+), LF_INIT ([0-9.]+), flatten ([0-9.]+), submit\+device ([0-9.]+), planes back ([0-9.]+)"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180)            # the drop-in as shipped: LF_INIT left to the back-end
        ms = re.findall(pat, r.stderr)
        if not ms:
            return None
        first, m = ms[0], ms[-1]
        r1 = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180, env=dict(os.environ, VVDEC_AMD_LF_INIT="1"))      # ... and with the reference's own LF_INIT (round 3)
        m1 = (re.findall(pat, r1.stderr) or [None])[-1]
        return {"lf_init": round(float(m[1]) / threads, 2), "lf_init_cpu_ms_summed_over_the_ctu_row_tasks": float(m[1]), "flatten": float(m[2]), "planes_back": float(m[4]), "pool_threads": threads,
                "lf_init_where": "on the device (VVR_TOOL_LFP_ON_DEVICE, k_lf_init): the drop-in neither runs LoopFilter::calcFilterStrengthsCTU nor copies its tables",
                "with_the_reference_lf_init_instead": None if not m1 else {"lf_init": round(float(m1[1]) / threads, 2), "lf_init_cpu_ms_summed_over_the_ctu_row_tasks": float(m1[1]), "flatten": float(m1[2]), "how": "VVDEC_AMD_LF_INIT=1"},
                "first_picture_of_the_process": {"lf_init": round(float(first[1]) / threads, 2), "flatten": float(first[2]), "planes_back": float(first[4])},
                "what": "ms per picture (one B picture of this stream) the reference-side half of the drop-in spends on the host: the reference's own edge-parameter derivation (one task per CTU row on the decoder's pool: lf_init = the tasks' summed time / pool threads), "
                        "the walk over the reference's CU / TU lists into the records of include/vvr.h, the finished planes copied back into the Picture's buffers.  The picture runs three times, "
                        "each through a decoder instance of its own; the figures are those of the third run (the process's memory is warm, as in a decoder that has been running), "
                        "first_picture_of_the_process those of the first (every buffer touched for the first time)"}
    except Exception:
        return None


def cpu_reference_threaded(cfg_name, seed, gop, threads, n_b=8):
    """The reference's OWN multithreaded reconstruction (north_star: "VVdeC's own multithreaded CPU path"): DecLibRecon's per-picture set-up and its
    15-state CTU task on its ThreadPool with `threads` threads (oracle/_ref, oracle/ref_harness.cpp::decompressFromLfInit - the scheduler of
    DecLibRecon.cpp:429-1110 started at LF_INIT, because the pictures arrive with their motion derived), one picture after the other in a process of its
    own: the IRAP of the stream and n_b of its B pictures, wall clock of decompressPicture + waitForPrevDecompressedPic per picture.
    -> frames/s in stream proportions (one IRAP per intra period), or None where that build is absent"""
    import subprocess
    code = (
        "import os, sys, json\n"
        "sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import refdrv, bench\n"
        "from vvdec_amd import abi, synth, stream\n"
        "assert refdrv.available()\n"
        "W, H, mix, period, _ = bench.CONFIGS[%r]\n"
        "tools = bench._tools(abi)\n"
        "if %r == 'allintra':\n"
        "    plans = [stream.PicPlan(poc=i, layer=0, slice_type=abi.SLICE_I, slot=0, ref_slots=([], [])) for i in range(1 + %d)]\n"
        "else:\n"
        "    plans, _ = stream.ra_plan(%d + 1, gop=%d, seed_poc0_is_external=False)\n"
        "    plans = [plans[0]] + plans[1:1 + %d]\n"
        "ms = []\n"
        "for k, pl in enumerate(plans):\n"
        "    d = synth.picture_for_plan(pl, W, H, seed=%d, tool_flags=tools, **mix)\n"
        "    refs = {slot: synth.natural_picture(W, H, %d + 100 + poc) for lst in pl.ref_slots for (slot, poc) in lst}\n"
        "    if k == 0: refdrv.reconstruct_threaded(d, refs, threads=%d)\n"        # (the first call of the process: library load, page faults)
        "    ms.append((int(pl.slice_type == abi.SLICE_I), refdrv.reconstruct_threaded(d, refs, threads=%d)['ms']))\n"
        "print('RESULT ' + json.dumps(ms))\n"
    ) % (ROOT, ROOT, cfg_name, cfg_name, n_b, gop, gop, n_b, seed, seed, threads, threads)
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            return None
        ms = json.loads(line[0][7:])
        i_ms = [m for (is_i, m) in ms if is_i]
        b_ms = [m for (is_i, m) in ms if not is_i] or i_ms
        period = CONFIGS[cfg_name][3] if cfg_name != "allintra" else 1
        mean_ms = (sum(i_ms) / len(i_ms) + (period - 1) * sum(b_ms) / len(b_ms)) / period if period > 1 else sum(i_ms) / len(i_ms)
        return {"fps": round(1e3 / mean_ms, 2), "threads": threads, "ms_per_I_picture": round(sum(i_ms) / len(i_ms), 1), "ms_per_B_picture": round(sum(b_ms) / len(b_ms), 1), "pictures": len(ms)}
    except Exception:
        return None


def cpu_baseline(cfg_name, seed, gop, budget_s=20.0):
    import refdrv
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor
    kind = "reference" if refdrv.available() else "port"
    W, H = CONFIGS[cfg_name][:2]
    # one process per picture.  Not more than 64 of them: with one per hardware thread of a 256-thread host the processes fight for memory
    # bandwidth (measured on the MI355X box: 256 processes 70 frames/s at 3.6 s per picture, 32 processes 158 frames/s at 0.2 s)
    cores = min(os.cpu_count() or 1, 64)
    try:
        import psutil                                             # one process holds a picture description, its planes and the reference's object graph
        cores = max(1, min(cores, int(psutil.virtual_memory().available / (3.0e9 * (4 if W > 4000 else 1)))))
    except Exception:
        pass
    t1 = _cpu_worker((kind, cfg_name, seed, gop, 0))             # calibration: one picture on one core
    per_core = max(1, min(3, int(budget_s / max(6 * t1, 1e-3))))
    n = cores * per_core
    with ProcessPoolExecutor(max_workers=cores, mp_context=multiprocessing.get_context("spawn")) as ex:
        list(ex.map(_cpu_worker, [(kind, cfg_name, seed, gop, i) for i in range(cores)]))       # start the workers (imports, library loads)
        t0 = time.perf_counter()
        times = list(ex.map(_cpu_worker, [(kind, cfg_name, seed, gop, i) for i in range(n)]))
        wall = time.perf_counter() - t0
    # the workers also generate their picture and (reference) build the reference's object graph, which is not reconstruction work:
    # value = pictures / (summed reconstruction-stage time / cores), the stage throughput of `cores` busy cores; the wall clock is stated
    fps = n / (sum(times) / cores)
    per_process = {"value": round(fps, 2), "cores": cores,
                   "what": "%d pictures of the same %dx%d stream, one picture per process on %d cores (frame-parallel, no scheduler: reconstruction stage %.0f ms/picture/core summed and divided by the cores; "
                           "wall clock of the sample incl. picture generation %.1f s)%s" % (n, W, H, cores, 1e3 * sum(times) / n, wall, ", reference classes with SIMD" if kind == "reference" else ", plain-C restatement")}
    if kind == "reference":
        # the headline baseline: the reference's own multithreaded path, wall clock - on 64 threads and on every hardware thread of the host
        t64 = cpu_reference_threaded(cfg_name, seed, gop, min(64, os.cpu_count() or 1))
        tall = cpu_reference_threaded(cfg_name, seed, gop, os.cpu_count() or 1) if (os.cpu_count() or 1) > 64 else None
        if t64:
            # the headline is a WALL-CLOCK rate of the reference's own threaded path (round-5 verdict: the per-process figure is an extrapolation - stage time summed over
            # one-picture processes, divided by the cores - and stands beside it, as does the run on every hardware thread)
            best = t64
            return {"value": best["fps"], "unit": "frames/s", "cores": best["threads"], "kind": "reference", "host_cores": os.cpu_count(), "which": "the reference's own threaded path on 64 threads, wall clock (one_picture_per_process: the frame-parallel extrapolation; threaded_all_hardware_threads: the same path on every hardware thread)",
                    "sample": "the IRAP and %d B pictures of the same %dx%d stream, one after the other through the reference's own scheduler (DecLibRecon's set-up + ctuTask state machine on its ThreadPool, "
                              "started at LF_INIT: the pictures arrive with their motion derived) on %d threads, wall clock of decompressPicture + waitForPrevDecompressedPic per picture, weighted one IRAP per "
                              "intra period (%.1f ms per I picture, %.1f ms per B picture); reference classes with SIMD" % (best["pictures"] - 1, W, H, best["threads"], best["ms_per_I_picture"], best["ms_per_B_picture"]),
                    "threaded_64": t64, "threaded_all_hardware_threads": tall, "one_picture_per_process": per_process}
    return {"value": per_process["value"], "unit": "frames/s", "cores": cores, "kind": kind, "host_cores": os.cpu_count(), "sample": per_process["what"]}


def kernel_source_hash():
    """MD5 over the kernel and host sources of the library (the files the Makefile lists): the counter / traffic summaries under profiles/ carry the hash of
    the sources they were measured with (tools/collect_profiles.py, tools/gpu_r6_counters.sh), and the line says "stale" when it differs"""
    import hashlib
    h = hashlib.md5()
    d = os.path.join(ROOT, "vvdec_amd", "csrc")
    for f in ("vvr_api.cpp", "vvr_prepare.cpp", "vvr_kernels.hip", "vvr_device.h", "vvr_host.h", "vvr_lf_init.h", "vvr_output.inc", "vvr_intra_leaf.inc", "vvr_intra_cells.inc"):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def picture_sharding_pass(a, W, H, mix, tools, plans, nslots, first, K, Wm, rank, world, local_rank, backend):
    """ONE stream over all ranks, a picture per rank at a time (vvdec_amd.parallel.PictureParallel); timed like the main pass: the K pictures of the
    window between barriers, max over ranks.  Strong scaling: the work is the same whatever N is."""
    import torch
    import torch.distributed as dist
    import vvdec_amd
    from vvdec_amd import synth, parallel
    dev = "cuda" if backend == "nccl" else "cpu"
    if backend != "nccl" and "hoststub" not in os.path.basename(vvdec_amd._LIBPATH or ""):
        # (the DPB of this mode lives in host memory then, and streams / events are the stand-in's: with the product library the kernels would be handed host pointers)
        raise RuntimeError("VVR_BENCH_BACKEND=%s is the control-flow test of this pass on the stand-in runtime (tools/bench_host_side.py); with the product library use the nccl backend" % backend)
    dpb = vvdec_amd.Reconstructor.new_dpb_tensor(W, H, nslots, device=dev)
    rec = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=a.streams, device=local_rank, host_threads=a.host_threads, ext_planes=dpb.data_ptr())
    # (gloo: the control-flow test of this path on the stand-in runtime, whose streams and events are the stub library's)
    mk_rt = (lambda: None) if backend == "nccl" else (lambda: parallel.HostStubRuntime(vvdec_amd.lib()))
    transfer = os.environ.get("VVR_BENCH_TRANSFER", "p2p")        # "broadcast": the rank-wide RCCL broadcast instead of sends to the dependants
    pp = parallel.PictureParallel(rec, dpb, plans, rank, world, runtime=mk_rt(), transfer=transfer)
    descs = [synth.picture_for_plan(pl, W, H, seed=1234, tool_flags=tools, alloc=rec.host_array, **mix) if pp.owners[i] == rank else None for i, pl in enumerate(plans)]
    samples = []
    for rep in range(max(1, min(3, a.repeats))):
        # every window from the first picture of the stream (pre-roll and warm-up untimed), as the one-GPU line does
        if rep:
            pp = parallel.PictureParallel(rec, dpb, plans, rank, world, runtime=mk_rt(), transfer=transfer)
        pp.run(descs, 0, first)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        pp.run(descs, first, first + K)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        samples.append(float(t.item()))
    med = sorted(samples)[len(samples) // 2]
    nb = sum(1 for i in range(first, first + K) if pp.need[i])
    sends = sum(len(pp.deps[i]) for i in range(first, first + K))
    host_waits = sum(1 for (op, _) in pp.trace if op == "host_wait")
    n_irap = sum(1 for pl in plans[first:first + K] if pl.slice_type == 2)
    rec.close()
    return {"fps": round(K / med, 2), "seconds": med, "samples_fps": [round(K / x, 1) for x in samples], "scaling": "weak", "pictures": K, "pictures_per_rank": K // world, "irap_pictures_in_window": n_irap,
            "replicated_pictures_in_window": nb, "point_to_point_sends_in_window": sends, "transfer": transfer,
            "slot_MB": round(dpb.numel() / nslots / 1e6, 1), "host_waits_for_hand_over": host_waits,
            "what": "ONE stream over all ranks, pictures round-robin within their temporal layer; a reference picture goes from its owner to the ranks that predict from it (point-to-point over xGMI, RCCL; "
                    "VVR_BENCH_TRANSFER=broadcast: to every rank), ordered on the device: the collective's stream waits for the picture's event, dependants wait for the event behind the receive; "
                    "a step is one picture per GPU: the window holds steps x ranks pictures of the stream, value = those pictures / max-over-ranks time"}


def picture_pass_in_children(rank, timeout):
    """N > 1: every rank starts this script once more (same arguments, VVR_BENCH_CHILD=picture) and the children - a process group of their own on the next
    port - run picture_sharding_pass; rank 0's child prints the result.  The pass has never run on more than one device (no such node was at hand): a fault
    in it must not take the segment-mode result of the line with it."""
    import subprocess
    entry = os.path.abspath(getattr(sys.modules.get("__main__"), "__file__", __file__))
    env = dict(os.environ, VVR_BENCH_CHILD="picture", MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 23), TORCHELASTIC_USE_AGENT_STORE="False")
    try:
        r = subprocess.run([sys.executable, entry] + sys.argv[1:], env=env, capture_output=True, text=True, timeout=max(1, timeout))
    except subprocess.TimeoutExpired:
        return {"error": "no result within %d s" % timeout, "timeout": True}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if rank != 0:
        return {}
    if r.returncode == 0 and lines:
        try:
            return json.loads(lines[-1])
        except ValueError:
            pass
    return {"error": "the picture-sharding processes ended with %d: %s" % (r.returncode, (r.stderr or "")[-300:].replace("\n", " | "))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="4k")
    ap.add_argument("--width", type=int, default=0, help="override the configuration's picture size (tests)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--gop", type=int, default=32, help="hierarchical-B GOP size (SURVEY 8(d) config 2: 32, six temporal layers)")
    ap.add_argument("--repeats", type=int, default=5, help="the timed K-picture window is run this many times, each from the first picture of the stream; value = the median")
    ap.add_argument("--streams", type=int, default=-1, help="pictures in flight per GPU (lanes).  Default: 4 for the random-access configurations (3-4 lanes give the device-only rate of 8 and a better window rate), 12 for all-intra (I pictures are bound by their dependency chain: more of them side by side); profiles/round3_lanes_and_host_threads.txt")
    ap.add_argument("--host-threads", type=int, default=-1, help="worker threads inside the library that prepare submitted pictures (default: 8, 16 for --config 8k; round 3, driver arguments: 8 threads 1014..1063 frames/s over five windows, 16 threads 780..1048: sixteen pictures prepared at once contend for memory bandwidth)")
    ap.add_argument("--ring", type=int, default=0, help="entries of the library's upload ring (0: its default)")
    ap.add_argument("--slots", type=int, default=48, help="DPB slots used round-robin (physical slots are cheap in 288 GB: 48 x 25 MB at 4K; fewer slots = more write-after-read waits between pictures in flight)")
    ap.add_argument("--intra-period", type=int, default=-1, help="an IRAP picture every N pictures (multiple of --gop); default: the configuration's")
    ap.add_argument("--irap-lookahead", type=int, default=8, help="IRAP pictures are submitted N pictures ahead of their decoding-order position (they depend on nothing)")
    ap.add_argument("--lf-init", choices=["device", "host"], default="device", help="device: the back-end derives the deblocking edge parameters itself (VVR_TOOL_LFP_ON_DEVICE, the reference's LF_INIT task on the GPU; the description's tables stay on the host); host: the generator's tables are uploaded (rounds 1-3)")
    ap.add_argument("--affine-mv", choices=["device", "host"], default="device", help="device: VVR_TOOL_AFFINE_MV_ON_DEVICE - the back-end spans the sub-block vectors of affine CUs from the control points itself (k_mc_affine, k_lf_maps); host: they travel with the picture (the motion field's cells under affine CUs)")
    ap.add_argument("--pageable-records", action="store_true", help="keep the host records in ordinary (pageable) memory: the library stages them through its pinned ring")
    ap.add_argument("--no-picture-sharding", action="store_true", help="N > 1: skip the additional pass that shards ONE stream by picture over the ranks")
    ap.add_argument("--picture-sharding-timeout", type=int, default=150, help="N > 1: seconds after which the picture-sharding pass is given up and the line is printed without it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1, --config 4k: do not run the 8K and all-intra configurations behind the headline one (config.other_configs)")
    ap.add_argument("--stop-after", type=int, default=0, help="developer: end the kernel chain after the reconstruction (1), deblocking (2) or SAO (3) stage, to see what a stage costs; needs --verify 0")
    ap.add_argument("--verify", type=int, default=8, help="number of timed pictures re-checked against the CPU oracle after the run")
    a = ap.parse_args()
    if a.streams < 0:
        a.streams = 12 if a.config == "allintra" else 4       # (all-intra: 8 / 10 / 12 / 14 / 16 lanes: device only 900 / 1009 / 1199 / 1236 / 880, through vvr_submit 816 / 934 / 936 / 937 / 629)
    if a.host_threads < 0:
        a.host_threads = 16 if a.config == "8k" else 8       # (8K: four times the host work per picture - 8 threads 261, 16 threads 339 frames/s through vvr_submit)

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
    from vvdec_amd import abi, synth, parallel
    vvdec_amd.lib()
    W, H, mix, ip_default, cfg_text = CONFIGS[a.config]
    if a.width:
        W, H = a.width, a.height
    tools = _tools(abi) | (abi.TOOL_LFP_ON_DEVICE if a.lf_init == "device" else 0) | (abi.TOOL_AFFINE_MV_ON_DEVICE if a.affine_mv == "device" else 0)
    K, Wm = a.steps, a.warmup
    intra_period = ip_default if a.intra_period < 0 else a.intra_period
    plans, nslots, orders = stream_plan(a.config, a.gop, intra_period, a.irap_lookahead, a.slots, K, Wm)
    order, first = orders[a.irap_lookahead]           # submission order of the headline figure (indices into plans) and its first timed picture
    n_irap = sum(1 for i in order[first:first + K] if plans[i].slice_type == abi.SLICE_I)
    if os.environ.get("VVR_BENCH_CHILD") == "picture" and world > 1:
        # the picture-sharding pass in processes of its own (one child per rank, a process group of their own: see picture_pass_in_children): whatever happens to
        # it - an exception, a collective that never completes, a fault on the device - the parents live to print the line with the segment-mode result
        try:
            # a step = one picture per GPU: the window of this mode holds K x world pictures of the ONE stream (at the driver's K = 20: 40 / 80 / 160 pictures at
            # 2 / 4 / 8 GPUs - five GOPs of 32 in flight at N = 8; IRAP pictures in stream proportion)
            plans_p, nslots_p, orders_p = stream_plan(a.config, a.gop, intra_period, a.irap_lookahead, a.slots, K * world, Wm)
            order_p, first_p = orders_p[a.irap_lookahead]
            res = picture_sharding_pass(a, W, H, mix, tools, [plans_p[i] for i in order_p], nslots_p, first_p, K * world, Wm, rank, world, local_rank, backend)
        except Exception as e:            # noqa: BLE001
            res = {"error": repr(e)[:300]}
        if rank == 0:
            print(json.dumps(res), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    seed = parallel.segment_seed(1234, rank)          # every rank reconstructs its own closed-GOP segment (no data-path collective)
    rec = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=a.streams, device=local_rank, host_threads=a.host_threads, ring_entries=a.ring, stop_after=a.stop_after)
    # the records are written where a parser integrated with the back-end would write them: host memory the device reads directly (vvr_host_alloc)
    needed = max(max(o) for o, _ in orders.values()) + 1
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as tp:        # (the generator is a C library: the calls release the GIL)
        descs = list(tp.map(lambda pl: synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, alloc=None if a.pageable_records else rec.host_array, **mix), plans[:needed]))
    cpics = [d.c() for d in descs]                     # the host records as the C ABI sees them (plain structs pointing at the arrays)
    upload_mb = sum(descs[i].cu.nbytes + descs[i].tu.nbytes + descs[i].coef.nbytes + (0 if a.lf_init == "device" else descs[i].lfp[0].nbytes + descs[i].lfp[1].nbytes) for i in order[first:first + K]) / K / 1e6

    enq = {}

    def barrier(all_ranks):
        rec.sync()
        torch.cuda.synchronize()
        if world > 1 and all_ranks:
            dist.barrier()
        torch.cuda.synchronize()

    def host_pass(idx, all_ranks=True):
        # the C-ABI path: host records in, reconstructed pictures in the DPB
        barrier(all_ranks)
        t0 = time.perf_counter()
        for i in idx:
            rec.submit_c(cpics[i])
        enq["host"] = time.perf_counter() - t0              # the submitting thread is done here; the rest is the wait for the pictures
        barrier(all_ranks)
        return time.perf_counter() - t0

    def window(order_, first_):
        """the stream from its first picture: pre-roll and warm-up untimed, then exactly K steps through vvr_submit, timed (max over ranks)"""
        host_pass(order_[:first_ - Wm])
        host_pass(order_[first_ - Wm:first_])
        dt_ = host_pass(order_[first_:first_ + K])
        if world > 1:
            t = torch.tensor([dt_], device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_

    # ---- the timed runs.  One K-picture window is 10-20 ms of a pipeline with 16 host threads in it: a noisy instrument.  The window is therefore
    # run --repeats times, every time from the first picture of the stream (fresh pre-roll), and `value` is the MEDIAN; all samples are printed.
    # The same stream in plain decoding order (no IRAP look-ahead: what a host without a reorder buffer submits) is timed next to it.
    dts = [window(order, first) for _ in range(max(1, a.repeats))]
    enq_host = enq["host"]
    tail_n = min(8, K)
    keep = order[first + K - tail_n:first + K] + [i for i in order[first:first + K] if plans[i].slice_type == abi.SLICE_I]
    later = {i: k for k, i in enumerate(order[:first + K])}
    # (a timed picture can be checked if nothing submitted after it has overwritten its slot: the last pictures and, with a 48-slot DPB, the IRAP)
    timed_out = {i: rec.read_picture(plans[i].slot) for i in dict.fromkeys(keep) if all(plans[j].slot != plans[i].slot for j in order[later[i] + 1:first + K])}
    dts0 = []
    if a.irap_lookahead and a.config != "allintra":
        order0, first0 = orders[0]
        dts0 = [window(order0, first0) for _ in range(max(1, min(3, a.repeats)))]
    dt = float(np.median(dts))

    # ---- the same K pictures with records and work lists resident in HBM (the device pipeline alone)
    prepared = {}

    def resident_pass(k0, n, all_ranks=True):
        idx = order[k0:k0 + n]
        for i in idx:
            if i not in prepared:
                prepared[i] = rec.prepare(descs[i])
        barrier(all_ranks)
        t0 = time.perf_counter()
        for i in idx:
            rec.submit_prepared(prepared[i])
        enq["device"] = time.perf_counter() - t0
        barrier(all_ranks)
        return time.perf_counter() - t0

    window(order, first)                               # (the DPB as the headline order leaves it)
    resident_pass(first - Wm, Wm)
    dt_dev = resident_pass(first, K)
    enq_dev = enq["device"]
    if world > 1:
        t = torch.tensor([dt_dev], device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_dev = float(t.item())

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: a further pass over the K timed pictures only, HIP-event timing on the launch streams
        resident_pass(first - Wm, Wm, all_ranks=False)
        rec.enable_stats(True)
        resident_pass(first, K, all_ranks=False)
        st = rec.stats()
        rec.enable_stats(False)
        copy_bps = rec.copy_bandwidth(20)
        for x in st:
            x["total_ms"] = max(x["total_ms"], 1e-6)          # (a kernel that never ran has no time)
        st.sort(key=lambda s: -s["total_ms"])
        dom = st[0]
        achieved = dom["algo_bytes"] / (dom["total_ms"] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": None,
                "peak_measured": round(copy_bps / 1e9, 1), "peak_measured_how": "the library's copy kernel over the DPB (%d MB read + %d MB written per launch), HIP events, 20 launches, same run" % ((min(nslots, 2 * (a.streams + (a.streams >= 2))) * rec.slot_bytes()) >> 20, (min(nslots, 2 * (a.streams + (a.streams >= 2))) * rec.slot_bytes()) >> 20),
                "frac_of_measured": round(achieved / max(copy_bps / 1e9, 1e-9), 4),
                "avg_launch_us": round(1e3 * dom["total_ms"] / dom["launches"], 2),
                "algorithmic_bytes_per_launch": int(dom["algo_bytes"] / dom["launches"]),
                "frame_level": {"algorithmic_MB_per_picture": round(sum(s["algo_bytes"] for s in st) / K / 1e6, 1),
                                "achieved_GBps_at_value": None},
                "all_kernels": {s["name"]: {"avg_us": round(1e3 * s["total_ms"] / s["launches"], 2), "launches": s["launches"],
                                            "algo_GBps": round(s["algo_bytes"] / (s["total_ms"] * 1e-3) / 1e9, 1)} for s in st}}
        # the two kinds of picture apart: the kernel with the largest total time among those of the B pictures, and the intra stage of the I picture(s)
        # (k_intra: the CTU-tile kernel only I pictures take since round 5; k_intra_leaf: the intra stage of the B pictures)
        def _entry(x):
            return {"kernel": x["name"], "avg_launch_us": round(1e3 * x["total_ms"] / x["launches"], 2), "launches": x["launches"], "algorithmic_bytes_per_launch": int(x["algo_bytes"] / x["launches"]),
                    "achieved": round(x["algo_bytes"] / (x["total_ms"] * 1e-3) / 1e9, 1), "unit": "GB/s", "frac": round(x["algo_bytes"] / (x["total_ms"] * 1e-3) / 8e12, 4)}
        bk = [x for x in st if x["name"] != "k_intra" and x["launches"]]
        ik = [x for x in st if x["name"] == "k_intra" and x["launches"]]
        roof["b_picture_kernel"] = _entry(bk[0]) if bk and a.config != "allintra" else None
        roof["i_picture_kernel"] = _entry(ik[0]) if ik else None
        # VALU-issue utilisation of the kernels alone on the device (a separate rocprofv3 --pmc pass: tools/gpu_r5_counters.sh -> profiles/round5_sq_counters.json)
        src_hash = kernel_source_hash()
        roof["kernel_source_hash"] = src_hash
        try:
            sqf = next(f for f in ("round6_sq_counters.json", "round5_sq_counters.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            sqd = json.load(open(os.path.join(ROOT, "profiles", sqf)))
            sq = sqd["kernels"]
            roof["valu_frac_source"] = "profiles/" + sqf
            roof["valu_frac_stale"] = sqd.get("kernel_source_hash") != src_hash          # the counter pass was made with other kernel sources than this library
            def _valu(name):
                v = [e["valu_issue_utilisation"] for k, e in sq.items() if k.split("<")[0] == name or (name == "k_alf" and k.startswith("k_sao_alf")) or (name in ("k_deblock_v", "k_deblock_h") and k.startswith("k_deblock_tile"))]
                return round(max(v), 3) if v else None
            roof["valu_frac"] = _valu(dom["name"])
            roof["valu_frac_all_kernels"] = {x["name"]: _valu(x["name"]) for x in st if _valu(x["name"]) is not None}
            roof["valu_frac_what"] = "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCCs), one picture in flight (profiles/round5_sq_counters.json): the share of the device's VALU issue slots the kernel fills when it runs alone"
        except Exception:
            pass
        # HBM-side traffic of that kernel from the separate PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over this command,
        # gfx950 correction applied, profiles/*_pmc_traffic.json); counters cannot be collected inside this run
        # (keyed by configuration AND kernel: the 8K and all-intra lines must not carry the 4K figure)
        for name in ("round6_pmc_traffic.json", "round5_pmc_traffic.json", "round4_pmc_traffic.json", "round3_pmc_traffic.json"):
            try:
                pmcd = json.load(open(os.path.join(ROOT, "profiles", name)))
                pmc = pmcd["configs"][a.config]["kernels"]
                roof["traffic_stale"] = pmcd.get("kernel_source_hash") != src_hash
                ent = next(v for k, v in pmc.items() if k.split("<")[0] == dom["name"])
                roof["traffic"] = ent["hbm_side_bytes_per_launch_corrected"]
                roof["traffic_source"] = "profiles/%s, configuration %s (separate rocprofv3 --pmc passes, bytes per launch)" % (name, a.config)
                break
            except Exception:
                pass
        # ---- correctness of what was timed
        verified = 0
        serial_equal = False
        if a.verify:
            import refdrv
            rec2 = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=1, device=local_rank)
            # the IRAP of the window (the picture that took the priority lane and overtook others) and the last timed pictures
            check = set([i for i in timed_out if plans[i].slice_type == abi.SLICE_I] + [i for i in order[first + K - tail_n:first + K] if i in timed_out][-min(a.verify, tail_n):])
            for i in order:
                pl, d = plans[i], descs[i]
                refs = None
                if i in check:
                    refs = {slot: rec2.read_picture(slot) for lst in pl.ref_slots for (slot, _) in lst}
                rec2.wait(rec2.decompress_picture(d))
                if i in timed_out:
                    got = rec2.read_picture(pl.slot)
                    assert all(np.array_equal(g, w) for g, w in zip(got, timed_out[i])), "bench: POC %d differs between the timed run and a one-picture-at-a-time run" % pl.poc
                    serial_equal = True
                    if i in check:                                                  # the timed picture against the CPU oracle, same reference pictures
                        want = refdrv.oracle_reconstruct(d, refs)
                        assert all(np.array_equal(g, w) for g, w in zip(got, want)), "bench: POC %d differs from the oracle" % pl.poc
                        verified += 1
            rec2.close()
        fps = world * K / dt
        roof["frame_level"]["achieved_GBps_at_value"] = round(roof["frame_level"]["algorithmic_MB_per_picture"] * 1e6 * fps / world / 1e9, 1)
        metric = {"4k": "decoded frames/sec (4K 10-bit RA) on MI355X, synthetic pre-parsed stream, bit-exact vs ref",
                  "8k": "decoded frames/sec (8K 10-bit RA) on MI355X, synthetic pre-parsed stream, bit-exact vs ref",
                  "allintra": "decoded frames/sec (4K 10-bit all-intra) on MI355X, synthetic pre-parsed stream, bit-exact vs ref"}[a.config]
        out = {"metric": metric, "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
               "ms_per_step": round(1e3 * dt / K, 4), "higher_is_better": True, "scaling": "weak" if world > 1 else None, "vs_baseline": None,
               "dtype": "int16", "data": "synthetic",
               "config": {"workload": "%s, %dx%d, CTU 128%s; timed: K pictures through vvr_submit from host records (validation, work lists on %d library threads, H2D of %.1f MB per picture, all kernels); %d pre-roll + W warm-up pictures untimed; %d IRAP picture(s) in the timed window"
                                      % (cfg_text, W, H, "" if a.config == "allintra" else ", hierarchical-B GOP %d, IRAP every %d pictures, submitted %d pictures ahead of its decoding-order position" % (a.gop, intra_period, a.irap_lookahead),
                                         a.host_threads, upload_mb, first - Wm, n_irap),
                          "conformance": "JVET set unpinned: no JVET bitstream exists offline; parser-fed parity is pinned on the streams tools/mini_vvenc.py writes (tests/bitstreams, reference decoder's MD5 / picture hashes)",
                          "timed_path": "vvr_submit(host records)", "affine_sub_block_mvs": "spanned on the device from the control points (VVR_TOOL_AFFINE_MV_ON_DEVICE)" if a.affine_mv == "device" else "supplied by the host (motion field)", "lf_init": "k_lf_init (VVR_TOOL_LFP_ON_DEVICE: edge parameters derived on the device, no table uploaded)" if a.lf_init == "device" else "tables supplied by the host", "host_records_in": "pageable memory (staged by the library)" if a.pageable_records else "pinned host memory of the context (vvr_host_alloc): cu / tu / coef / lfp arrays are copied to HBM from where the generator wrote them", "host_threads": a.host_threads, "host_cores": os.cpu_count(),
                          "value_is": "median of %d runs of the K-picture window, each from the first picture of the stream (pre-roll and warm-up untimed)" % len(dts),
                          "value_samples_fps": [round(world * K / x, 1) for x in dts], "value_min_fps": round(world * K / max(dts), 2), "value_max_fps": round(world * K / min(dts), 2),
                          "value_irap_lookahead_0": round(world * K / float(np.median(dts0)), 2) if dts0 else None,
                          "value_irap_lookahead_0_samples_fps": [round(world * K / x, 1) for x in dts0],
                          "value_irap_lookahead_0_what": "the same stream submitted in plain decoding order (a host without a reorder buffer): the IRAP arrives when its turn comes",
                          "device_only_fps": round(world * K / dt_dev, 2), "device_only_ms_per_step": round(1e3 * dt_dev / K, 4),
                          "submit_loop_ms": {"vvr_submit": round(1e3 * enq_host, 2), "vvr_submit_prepared": round(1e3 * enq_dev, 2), "what": "time the submitting thread spends in the K calls (of the timed K-picture passes: %.2f / %.2f ms)" % (1e3 * dt, 1e3 * dt_dev)},
                          "device_only_what": "same K pictures, records and work lists resident in HBM (vvr_prepare + vvr_submit_prepared)",
                          "irap_in_window": n_irap, "irap_share_of_stream": "1/%d" % intra_period if a.config != "allintra" else "1/1",
                          "tools": "intra planar/DC/angular/wide-angle + PDPC + MRL + reference smoothing + BDPCM + ISP, LFNST, inter uni/bi MC (8/4-tap DCTIF, alt half-pel, BCW), BDOF, DMVR, affine 4/6-parameter + PROF, GPM, CIIP, SbTMVP, CCLM/MDLM, MIP, LMCS luma mapping + chroma residual scaling, dequant + dep-quant, DCT2/DST7/DCT8 + transform skip, joint Cb-Cr, deblocking, SAO, ALF + CC-ALF",
                          "mix": mix,
                          "picture_sharding": None, "pictures_in_flight": a.streams, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "sharding": "closed-GOP segment per GPU, no data-path collective" if world > 1 else None,
                          "verified_timed_pictures_vs_oracle": verified, "timed_run_equals_serial_run": serial_equal},
               "roofline": roof}
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.config, seed, a.gop)
            if a.config != "allintra":
                cost = dropin_host_cost(a.config, seed, a.gop)
                if cost:
                    out["cpu_baseline"]["dropin_host_ms_per_picture"] = cost
            if a.config == "4k":
                # the DROP-IN as a decoder: the reference's application on the drop-in libvvdec.so decoding the parser-fed 4K stream (tests/bitstreams/mini_4k_*:
                # 3840x2176, 17 pictures, every tool; output MD5 and picture hashes checked by the GPU suite), next to the reference decoder itself on the same stream
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import dropin_4k_rate
                    out["cpu_baseline"]["dropin_decoder_on_the_parser_fed_4k_stream"] = {"dropin": dropin_4k_rate.rate(16, loops=2, copies=6), "reference_decoder_on_the_cpu": dropin_4k_rate.reference_rate(16, loops=2, copies=6),
                        "dropin_17_pictures": dropin_4k_rate.rate(16), "reference_decoder_17_pictures": dropin_4k_rate.reference_rate(16),
                        "what": "vvdecapp -t 16 on the stream six times behind itself (102 pictures, six coded video sequences: the decoder pipelines across them), second of two loops; and on the 17 pictures alone (mostly pipeline fill and drain: the GOP's chain of five temporal layers)"}
                except Exception as e:            # noqa: BLE001
                    out["cpu_baseline"]["dropin_decoder_on_the_parser_fed_4k_stream"] = {"error": repr(e)[:200]}
    for h in prepared.values():
        rec.free_prepared(h)
    rec.close()
    # ---- BASELINE configs[2] (8K RA QP27) and configs[4] (4K all-intra QP22) behind the headline configuration, each a run of this script in a process of its
    # own (short windows, no CPU baseline, the IRAP / one more picture against the oracle): config.other_configs.  The driver only ever runs the default command.
    if rank == 0 and world == 1 and a.config == "4k" and not a.no_other_configs and not a.width:
        import subprocess
        del descs, cpics
        others = {}
        for name, extra in (("8k", ["--steps", "16", "--warmup", "4", "--repeats", "3", "--verify", "1", "--intra-period", "32"]), ("allintra", ["--steps", "64", "--warmup", "16", "--repeats", "5", "--verify", "2"])):
            t0 = time.perf_counter()
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", name, "--no-cpu-baseline", "--no-other-configs"] + extra, capture_output=True, text=True, timeout=400)
                o = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                c = o["config"]
                others[name] = {"fps": o["value"], "steps": o["steps"], "warmup": o["warmup"], "ms_per_step": o["ms_per_step"], "samples_fps": c["value_samples_fps"], "device_only_fps": c["device_only_fps"],
                                "verified_vs_oracle": c["verified_timed_pictures_vs_oracle"], "timed_run_equals_serial_run": c["timed_run_equals_serial_run"], "workload": c["workload"],
                                "roofline": {k: o["roofline"].get(k) for k in ("kernel", "achieved", "frac", "avg_launch_us", "traffic", "b_picture_kernel", "i_picture_kernel")}, "seconds": round(time.perf_counter() - t0, 1)}
            except Exception as e:            # noqa: BLE001 - the headline line must survive
                others[name] = {"error": repr(e)[:300]}
        out["config"]["other_configs"] = others
        # the same 4K stream through vvr_submit with K = 64 (the default arguments of this script): the pipeline's steady state next to the driver's short window,
        # whose IRAP chain and fill weigh 3.2 times their share of the stream (round-5 verdict: a first-class field of the line, not only a file under profiles/)
        if K < 64:
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", "64", "--warmup", "16", "--repeats", "3", "--verify", "0", "--no-cpu-baseline", "--no-other-configs"], capture_output=True, text=True, timeout=300)
                o = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                out["value_k64"] = {"fps": o["value"], "steps": 64, "warmup": 16, "ms_per_step": o["ms_per_step"], "samples_fps": o["config"]["value_samples_fps"], "device_only_fps": o["config"]["device_only_fps"],
                                    "what": "the same stream and path (vvr_submit from host records) over a 64-picture window with one IRAP: python bench.py --steps 64 --warmup 16"}
            except Exception as e:            # noqa: BLE001
                out["value_k64"] = {"error": repr(e)[:300]}
    # ---- N > 1: the same stream sharded by PICTURE over the ranks (BASELINE north_star / SURVEY 8(e): pictures round-robin within their temporal layer,
    # reference pictures broadcast slot to slot over RCCL), next to the segment mode above.  Reported under config.picture_sharding.  It runs last and
    # under a watchdog: whatever happens to it (an exception, a collective that never completes), the line with the segment-mode result is printed.
    if world > 1 and a.config != "allintra" and not a.no_picture_sharding:
        import threading
        line_lock = threading.Lock()           # the line is printed once: by the watchdog or by the main path, whoever takes the lock first
        state = {"printed": False}

        def give_up():
            with line_lock:
                if state["printed"]:
                    return
                state["printed"] = True
                if rank == 0:
                    out["config"]["picture_sharding"] = {"error": "no result within %d s" % a.picture_sharding_timeout, "timeout": True}
                    out["timeout"] = True
                    print(json.dumps(out), flush=True)
            if rank != 0:
                time.sleep(3)
            os._exit(3)          # (ranks may hang in a collective: the process ends here, NOT with success - the line carries the segment-mode result and "timeout": true)

        timer = threading.Timer(a.picture_sharding_timeout + 30, give_up)     # (the children are given up after the timeout by picture_pass_in_children; this is for a parent that hangs)
        timer.daemon = True
        timer.start()
        try:
            pic_mode = picture_pass_in_children(rank, a.picture_sharding_timeout)
        except Exception as e:            # noqa: BLE001 - the segment-mode line must survive
            pic_mode = {"error": repr(e)[:300]}
        with line_lock:
            timer.cancel()
            if state["printed"]:
                return
            if rank == 0:
                out["config"]["picture_sharding"] = pic_mode
                if pic_mode.get("timeout"):
                    out["timeout"] = True
                # `value` stays the segment mode (frames sharded over the GPUs by closed-GOP segment, no inter-GPU reference, no data-path collective:
                # weak scaling, what north_star's "near-linear" is claimed for).  The picture-level split of ONE stream is reported beside it with the
                # ceiling its dependency graph allows: a hierarchical-B window is a chain of temporal layers behind its IRAP.
                # both modes as first-class fields of the line.  `value` is the PICTURE mode - the split north_star names: frames one-per-GPU of ONE stream, reference
                # pictures over xGMI; a step is one picture per GPU, so the window holds steps x ranks pictures ("scaling": "weak": the work per GPU is fixed as N
                # grows).  The closed-GOP segment mode measured above (no data-path collective) stands beside it; it becomes `value` only when the picture-mode pass
                # did not come back, and the line then says so.
                seg = {"fps": out["value"], "ms_per_step": out["ms_per_step"], "scaling": "weak", "collective_on_the_data_path": None, "pictures": K * world,
                       "what": "every rank its own closed-GOP segment and DPB, steps pictures each; RCCL carries the barrier and the max-over-ranks time"}
                out["value_segment_mode"] = seg
                from vvdec_amd import parallel as _par, stream as _stream
                dev_ms = 1e3 * dt_dev / K
                cost = lambda pl: 10.0 * dev_ms if pl.slice_type == 2 else dev_ms      # (an I picture counted as 10 B pictures: its intra wavefront)
                plans_p, _, orders_p = stream_plan(a.config, a.gop, intra_period, a.irap_lookahead, a.slots, K * world, Wm)
                order_p, first_p = orders_p[a.irap_lookahead]
                win = [plans_p[i] for i in order_p][first_p:first_p + K * world]
                long_ = _stream.ra_plan(intra_period * 8 + 1, gop=a.gop, seed_poc0_is_external=False, pool=a.slots, intra_period=intra_period)[0]
                xfer = pic_mode.get("slot_MB", 25.0) / 100.0        # ms: a slot over one xGMI link pair at ~100 GB/s
                sp_w, _, _ = _par.predicted_speedup(win, world, cost, xfer)
                sp_s, _, _ = _par.predicted_speedup(long_, world, cost, xfer)
                ceil_w, _, _ = _par.strong_scaling_ceiling(win, cost)
                ceil_s, _, _ = _par.strong_scaling_ceiling(long_, cost)
                ceilings = {"speedup_over_one_gpu_at_least": {"this_window": round(sp_w, 2), "open_stream": round(sp_s, 2)}, "speedup_over_one_gpu_at_most": {"this_window": round(min(world, ceil_w), 2), "open_stream": round(min(world, ceil_s), 2)},
                            "what": "bounds from the reference graph (DecLibRecon's whole-picture gating), an I picture counted as 10 B pictures at the measured device-only time per picture. At most: total work / "
                                    "critical path (capped at the number of GPUs). At least: a list schedule in which every rank takes its pictures strictly in order, ONE at a time, a reference from another rank "
                                    "%.2f ms later (parallel.predicted_speedup) - a back-end keeps several pictures in flight per GPU, so a picture that waits does not hold up the next. Open stream = 8 intra "
                                    "periods of the same hierarchy. None of this is a measurement" % xfer}
                out["value_picture_mode"] = {"fps": pic_mode.get("fps"), "scaling": "weak", "pictures": pic_mode.get("pictures"), "samples_fps": pic_mode.get("samples_fps"), "ceiling": ceilings,
                                             "MB_sent_per_picture": round(pic_mode.get("slot_MB", 0.0) * pic_mode.get("point_to_point_sends_in_window", 0) / max(1, pic_mode.get("pictures", 1)), 1) if "fps" in pic_mode else None,
                                             "ranks": dist.get_world_size(), "backend": "RCCL (torch.distributed nccl)" if backend == "nccl" else backend,
                                             "transfer": "point-to-point sends of a reconstructed picture to exactly the ranks that predict from it (batch_isend_irecv): most pictures of a GOP-32 have one or two "
                                                         "dependants among eight ranks - a refinement of north_star's broadcast, which VVR_BENCH_TRANSFER=broadcast runs instead" if pic_mode.get("transfer", "p2p") == "p2p" else "RCCL broadcast of every replicated picture to every rank",
                                             "error": pic_mode.get("error")}
                out["rccl_ranks"] = dist.get_world_size() if backend == "nccl" else 0
                if "fps" in pic_mode:
                    out["value"] = pic_mode["fps"]
                    out["ms_per_step"] = round(1e3 * pic_mode["seconds"] / K, 4)
                    out["scaling"] = "weak"
                    out["config"]["sharding"] = "ONE stream sharded by picture over the GPUs (round-robin within the temporal layer), reference pictures point to point over RCCL / xGMI; steps x GPUs pictures in the timed window"
                    out["config"]["value_is"] = "picture mode (value_picture_mode); the closed-GOP segment mode is value_segment_mode"
                else:
                    out["config"]["value_is"] = "SEGMENT mode: the picture-mode pass did not come back (%s)" % (pic_mode.get("error") or "no result")
            state["printed"] = True
            if rank == 0:
                print(json.dumps(out), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        if pic_mode.get("timeout"):
            sys.exit(3)          # (a pass that was given up is not a success: the line carries the segment-mode result and "timeout": true)
        return
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
