"""Multi-GPU sharding of the reconstruction path: one process per GPU, one closed-GOP segment per process.

The path shards by independently decodable SEGMENT (an IRAP picture and everything that predicts from it, directly or
transitively, up to the next IRAP — in the reference that is the unit `DecLib` can start decoding at,
source/Lib/DecoderLib/DecLib.cpp:182-312).  Pictures inside a segment depend on each other through their reference lists
(whole-picture dependencies, DecLibRecon.cpp:460-489), so a segment stays on one GPU with its own DPB; segments never read
each other's pictures, hence there is NO data-path collective.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is
used only for the control plane: the barrier around the timed region, the max-over-ranks time and gathering per-picture
MD5s for verification.

Picture-level sharding of ONE stream (BASELINE north_star: "frames shard one-per-GPU, DPB replicated via RCCL
broadcast only for inter-GPU references") is `PictureParallel` below: pictures go to the ranks round-robin within their temporal
layer, every rank keeps a DPB of its own in one torch tensor (`vvr_config.ext_planes`), and a reference picture that a picture
on another rank predicts from is broadcast once, slot to slot, from the rank that reconstructed it.
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


# ---------------------------------------------------------------------------------------------------------------------
# picture-level sharding of one stream
# ---------------------------------------------------------------------------------------------------------------------
def assign_owners(plans, world):
    """rank that reconstructs each picture of a stream (submission order): round-robin within the temporal layer, so that the
    many independent pictures of the upper layers spread over all ranks (an RA GOP-16 holds 1, 1, 2, 4, 8 pictures per layer)"""
    nxt, owners = {}, []
    for pl in plans:
        r = nxt.get(pl.layer, 0)
        owners.append(r % world)
        nxt[pl.layer] = r + 1
    return owners


def broadcast_plan(plans, owners):
    """for every picture: True if a picture reconstructed on ANOTHER rank predicts from it, i.e. its slot has to be replicated"""
    need = [False] * len(plans)
    by_poc = {pl.poc: i for i, pl in enumerate(plans)}
    for j, pl in enumerate(plans):
        for poc in list(pl.l0) + list(pl.l1):
            i = by_poc.get(poc)
            if i is not None and owners[i] != owners[j]:
                need[i] = True
    return need


class PictureParallel:
    """One stream over `world` ranks, one picture per rank at a time (DecLibRecon's whole-picture reference gating,
    DecLibRecon.cpp:460-489, across GPUs).

    Every rank walks the same plan.  The owner of a picture submits it to its back-end; if a picture on another rank references it,
    all ranks then take part in one broadcast of its DPB slot (three planes, one contiguous range of the DPB tensor) rooted at the
    owner - RCCL over xGMI on GPUs, gloo in the CPU tests.  Ordering:
      * the owner waits for the picture (vvr_wait) before the broadcast reads the slot;
      * a receiver waits for its own pictures that still read the slot's previous content before the broadcast overwrites it;
      * before a rank submits a picture, the broadcasts into its reference slots have completed.
    `rec` is this rank's Reconstructor created with ext_planes = dpb.data_ptr(); `dpb` a uint8 tensor of num_slots * rec.slot_bytes()."""

    def __init__(self, rec, dpb, plans, rank, world, replicate=True):
        self.rec, self.dpb, self.plans, self.rank, self.world = rec, dpb, plans, rank, world
        self.owners = assign_owners(plans, world)
        self.need = broadcast_plan(plans, self.owners) if replicate else [False] * len(plans)
        self.slot_bytes = rec.slot_bytes()
        self.pending = {}           # slot -> broadcast still in flight into / out of it
        self.users = {}             # slot -> local jobs that read or write it and have not been waited for
        self.trace = []             # (op, picture index): "submit", "wait", "bcast_send", "bcast_recv" - what the tests look at
        self.n_bcast = 0

    def _slot_view(self, slot):
        return self.dpb[slot * self.slot_bytes:(slot + 1) * self.slot_bytes]

    def _settle(self, slot):
        """the broadcast touching `slot` is complete as far as this rank's device is concerned"""
        import torch
        w = self.pending.pop(slot, None)
        if w is not None:
            w.wait()
            if self.dpb.is_cuda:
                torch.cuda.current_stream().synchronize()      # (RCCL: wait() only orders the current torch stream; the back-end has streams of its own)

    def _drain_users(self, slot):
        for job in self.users.pop(slot, []):
            self.rec.wait(job)

    def run(self, descs, i0=0, i1=None):
        """pictures [i0, i1) of the plan (default: all).  descs[i]: description of plans[i] for the pictures this rank owns (None elsewhere is
        fine).  Returns {picture index: job} of the pictures reconstructed here; everything of the range is complete on return, the DPB state
        carries over to the next call."""
        import torch.distributed as dist
        jobs = {}
        for i in range(i0, len(self.plans) if i1 is None else i1):
            pl = self.plans[i]
            mine = self.owners[i] == self.rank
            ref_slots = [slot for lst in (pl.ref_slots or ([], [])) for (slot, _) in lst]
            if mine:
                for slot in ref_slots + [pl.slot]:
                    self._settle(slot)
                job = self.rec.decompress_picture(descs[i])
                jobs[i] = job
                self.trace.append(("submit", i))
                for slot in ref_slots + [pl.slot]:
                    self.users.setdefault(slot, []).append(job)
            if self.need[i] and self.world > 1:
                if mine:
                    self.rec.wait(jobs[i])                              # the slot holds the picture
                    self.trace.append(("wait", i))
                    self.users[pl.slot] = [j for j in self.users.get(pl.slot, []) if j != jobs[i]]
                else:
                    self._settle(pl.slot)
                    self._drain_users(pl.slot)                          # nobody here still reads what the slot held before
                self.pending[pl.slot] = dist.broadcast(self._slot_view(pl.slot), src=self.owners[i], async_op=True)
                self.trace.append(("bcast_send" if mine else "bcast_recv", i))
                self.n_bcast += 1
        for slot in list(self.pending):
            self._settle(slot)
        self.rec.sync()
        self.users.clear()
        return jobs
