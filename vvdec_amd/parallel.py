"""Multi-GPU sharding of the reconstruction path: one process per GPU, one closed-GOP segment per process.

The path shards by independently decodable SEGMENT (an IRAP picture and everything that predicts from it, directly or
transitively, up to the next IRAP — in the reference that is the unit `DecLib` can start decoding at,
source/Lib/DecoderLib/DecLib.cpp:182-312).  Pictures inside a segment depend on each other through their reference lists
(whole-picture dependencies, DecLibRecon.cpp:460-489), so a segment stays on one GPU with its own DPB; segments never read
each other's pictures, hence there is NO data-path collective.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is
used only for the control plane: the barrier around the timed region, the max-over-ranks time and gathering per-picture
MD5s for verification.

Picture-level sharding inside one segment (frames round-robin over GPUs + one broadcast of the three
planes per reference picture) needs `vvr_config.ext_planes` DPB slots registered with RCCL; the slot pointers are already
exposed for that (`Reconstructor.plane_tensor`), the scheduler for it is a later round.
"""
import hashlib
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, timeout_s=600):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT (torch.distributed.run sets them).
    backend None -> "nccl" (= RCCL) when a GPU is visible, else "gloo"."""
    import datetime
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world == 1 or dist.is_initialized():
        return rank, world, local
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return rank, world, local


def segments_for_rank(num_segments, rank, world):
    """Segment indices this rank reconstructs: round-robin, so that a live stream (segments arriving in order) keeps all
    GPUs busy and every rank gets floor/ceil(num_segments / world) of them."""
    return list(range(rank, num_segments, world))


def segment_seed(base_seed, segment):
    """Generator seed of a segment of the synthetic stream (every segment is a different closed-GOP piece of content)."""
    return base_seed + 100000 * segment


def picture_md5(planes):
    """MD5 over the three planes in Y, Cb, Cr order, little-endian 16-bit samples, rows without padding — the layout
    the reference hashes for its decoded-picture-hash check (CommonLib/PicYuvMD5.cpp:197)."""
    h = hashlib.md5()
    for p in planes:
        h.update(p.astype("<u2").tobytes())
    return h.hexdigest()


def max_over_ranks(seconds):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def gather_results(local):
    """local: picklable per-rank result (e.g. [(segment, poc, md5), ...]); returns the concatenation over ranks on every rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return list(local)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(local))
    return [x for part in out for x in part]


def reconstruct_segments(num_segments, reconstruct_segment, rank=None, world=None):
    """Run `reconstruct_segment(segment) -> [(poc, md5), ...]` for this rank's share and gather [(segment, poc, md5)] from all
    ranks, sorted.  `reconstruct_segment` is the GPU back-end in production/bench and the CPU oracle in the gloo tests."""
    if rank is None:
        rank, world, _ = env_rank_world()
    mine = []
    for s in segments_for_rank(num_segments, rank, world):
        mine += [(s, poc, md5) for (poc, md5) in reconstruct_segment(s)]
    return sorted(gather_results(mine))
