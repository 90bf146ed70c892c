"""Multi-GPU sharding of the reconstruction path: one process per GPU.

PICTURE mode (`PictureParallel`; SURVEY.md §8(e), BASELINE north_star: "frames shard one-per-GPU, DPB replicated via RCCL ... only for
inter-GPU references"; what `bench.py --gpus N` reports as `value`): ONE stream over all ranks, pictures round-robin within their
temporal layer, every rank keeps a DPB of its own in one torch tensor (`vvr_config.ext_planes`), and a reconstructed picture that a
picture on ANOTHER rank predicts from goes, slot to slot, from the rank that reconstructed it to the ranks that need it.  The transfer
is point to point by default (`transfer="p2p"`: exactly the ranks that own a dependant take part - one or two of eight for most
pictures of a hierarchical-B GOP; xGMI is a point-to-point fabric, a ring broadcast would move the 25 MB of a 4K picture over every
link) and a rank-wide RCCL broadcast on request (`transfer="broadcast"`: the literal reading of north_star; every rank receives every
replicated picture).  Both are ordered on the device against the back-end's pictures (vvr_stream_wait_job / vvr_stream_wait_slot /
vvr_slot_external_event); the reference's DPB book-keeping this mirrors is CommonLib/PicListManager.cpp:234-283.

SEGMENT mode (`reconstruct_segments`; reported beside it as `value_segment_mode`): the stream shards by independently decodable
segment (an IRAP picture and everything that predicts from it up to the next IRAP - the unit `DecLib` can start decoding at,
source/Lib/DecoderLib/DecLib.cpp:182-312); a segment stays on one GPU with its own DPB, segments never read each other's pictures:
no data-path collective, torch.distributed carries the barrier, the max-over-ranks time and the per-picture MD5s only.
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


def dependants(plans, owners):
    """for every picture: the ranks (other than its owner) that reconstruct a picture predicting from it - the only ranks its slot is sent to"""
    deps = [set() for _ in plans]
    by_poc = {pl.poc: i for i, pl in enumerate(plans)}
    for j, pl in enumerate(plans):
        for poc in list(pl.l0) + list(pl.l1):
            i = by_poc.get(poc)
            if i is not None and owners[i] != owners[j]:
                deps[i].add(owners[j])
    return [sorted(d) for d in deps]


def strong_scaling_ceiling(plans, cost):
    """What sharding ONE stream by picture can reach at best, whatever the number of ranks: total work / critical path of the pictures'
    dependency graph (a picture starts when its reference pictures are done - DecLibRecon's whole-picture gating, DecLibRecon.cpp:460-489).
    cost(plan) = time of one picture alone on a device.  Returns (ceiling, total, critical)."""
    done = {}
    total = critical = 0.0
    for pl in plans:
        c = float(cost(pl))
        start = max([done.get(poc, 0.0) for poc in list(pl.l0 or []) + list(pl.l1 or [])], default=0.0)
        done[pl.poc] = start + c
        total += c
        critical = max(critical, done[pl.poc])
    return (total / critical if critical else 1.0), total, critical


def predicted_speedup(plans, world, cost, transfer_cost=0.0):
    """What the picture split of `plans` (submission order) over `world` ranks should reach against one rank, by a plain list schedule of the reference graph:
    every rank takes its pictures (assign_owners) in order, one after the other (a device's throughput: cost(plan) per picture), a picture starts when its
    rank is free and its reference pictures are done - plus transfer_cost when a reference was reconstructed on another rank.  Returns (speedup, makespan of
    one rank, makespan of `world` ranks).  For the driver's window sizes (20 pictures per GPU) the first IRAP's chain and the fill of the hierarchy are part
    of the window; over an open stream the IRAPs, which depend on nothing, leave the critical path and the speedup approaches `world`."""
    def makespan(n):
        owners = assign_owners(plans, n)
        free = [0.0] * n
        done = {}
        for pl, r in zip(plans, owners):
            start = free[r]
            for poc in list(pl.l0 or []) + list(pl.l1 or []):
                if poc in done:
                    t, o = done[poc]
                    start = max(start, t + (transfer_cost if o != r else 0.0))
            end = start + float(cost(pl))
            done[pl.poc] = (end, r)
            free[r] = end
        return max(free)
    t1, tn = makespan(1), makespan(world)
    return (t1 / tn if tn else 1.0), t1, tn


class TorchDeviceRuntime:
    """streams and events of the collective on a GPU: one torch stream the RCCL operations are ordered on"""

    def __init__(self):
        import torch
        self.torch = torch
        self.stream = torch.cuda.Stream()
        self.events = []

    def stream_ptr(self):
        return self.stream.cuda_stream

    def transfer(self, ops):
        """point-to-point operations (torch.distributed.P2POp) ordered behind everything on the collective's stream so far; the stream then
        waits for them (NCCL: Work.wait() orders the stream, it does not block the host)"""
        import torch.distributed as dist
        with self.torch.cuda.stream(self.stream):
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def broadcast(self, view, src):
        """rank-wide RCCL broadcast of a slot, ordered like transfer()"""
        import torch.distributed as dist
        with self.torch.cuda.stream(self.stream):
            dist.broadcast(view, src)

    def event_ptr(self):
        ev = self.torch.cuda.Event()
        ev.record(self.stream)
        self.events.append(ev)          # kept alive until the run is over (the back-end's pictures wait for it)
        return ev.cuda_event

    def finish(self):
        """the collective's stream has drained: every event handed to the back-end is complete"""
        self.stream.synchronize()

    def release(self):
        """after finish() AND the back-end's sync(): the back-end has dropped the (complete) events, they may go (vvr.h, vvr_slot_external_event)"""
        self.events.clear()


class HostStubRuntime:
    """the same on the stand-in runtime of the CPU tests (tests/hoststub: streams and events are inert objects whose waits / records are traced);
    the transfer itself is a blocking gloo operation"""

    def __init__(self, stub):
        import ctypes as C
        self.C, self.L = C, stub
        s = C.c_void_p()
        stub.hipStreamCreateWithFlags(C.byref(s), 0)
        self.s = s
        self.events = []

    def stream_ptr(self):
        return self.s

    def transfer(self, ops):
        import torch.distributed as dist
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def broadcast(self, view, src):
        import torch.distributed as dist
        dist.broadcast(view, src)

    def event_ptr(self):
        e = self.C.c_void_p()
        self.L.hipEventCreate(self.C.byref(e))
        self.L.hipEventRecord(e, self.s)
        self.events.append(e)
        return e

    def finish(self):
        pass

    def release(self):
        for e in self.events:
            self.L.hipEventDestroy(e)       # (the stand-in runtime counts any later use of a destroyed event: vvt_dead_event_uses)
        self.events = []


class PictureParallel:
    """One stream over `world` ranks, one picture per rank at a time (DecLibRecon's whole-picture reference gating,
    DecLibRecon.cpp:460-489, across GPUs).

    Every rank walks the same plan.  The owner of a picture submits it to its back-end; if pictures on OTHER ranks predict from it, its DPB
    slot (three planes, one contiguous range of the DPB tensor) goes from the owner to exactly those ranks - point-to-point sends over the
    xGMI links of the pairs concerned (RCCL; gloo in the CPU tests), no rank that has no use for the picture takes part.  Everything is ordered
    ON THE DEVICE; the host waits for no picture:
      * sender: the collective's stream waits for the picture's completion event (vvr_stream_wait_job), the sends follow on that stream, an event
        behind them keeps later pictures from overwriting the slot under the transfer (vvr_slot_external_event, reader);
      * receiver: the collective's stream first waits for every local picture that still uses what the slot held before (vvr_stream_wait_slot),
        then receives; pictures submitted afterwards that read the slot wait for the event behind the receive (vvr_slot_external_event, writer);
      * the owner goes on submitting its next pictures while the transfer of the last one runs.
    A picture can only be waited for once its work lists are built and it has been handed to the device (worker threads of the back-end); the
    communication steps therefore sit in a first-in first-out queue per rank that is pumped without blocking before every submit - in plan
    order, which keeps the sends and receives of every pair of ranks matched - and drained at the end of run().
    `rec` is this rank's Reconstructor created with ext_planes = dpb.data_ptr(); `dpb` a uint8 tensor of num_slots * rec.slot_bytes();
    `runtime`: TorchDeviceRuntime() on GPUs (default when dpb is a CUDA tensor), HostStubRuntime(stub library) in the CPU tests."""

    def __init__(self, rec, dpb, plans, rank, world, replicate=True, runtime=None, transfer="p2p"):
        assert transfer in ("p2p", "broadcast")
        self.rec, self.dpb, self.plans, self.rank, self.world = rec, dpb, plans, rank, world
        self.transfer = transfer
        self.owners = assign_owners(plans, world)
        self.deps = dependants(plans, self.owners) if replicate else [[] for _ in plans]
        if transfer == "broadcast":       # every rank receives every replicated picture
            self.deps = [[r for r in range(world) if r != self.owners[i]] if d else [] for i, d in enumerate(self.deps)]
        self.need = [bool(d) for d in self.deps]
        self.slot_bytes = rec.slot_bytes()
        self.rt = runtime if runtime is not None else (TorchDeviceRuntime() if getattr(dpb, "is_cuda", False) else None)
        self.fifo = []              # communication steps not issued yet: (picture index, job or None)
        self.trace = []             # (op, picture index): "submit", "send", "recv", and "host_wait" whenever the host had to wait - what the tests look at
        self.n_bcast = 0
        self.bytes_sent = 0

    def _slot_view(self, slot):
        return self.dpb[slot * self.slot_bytes:(slot + 1) * self.slot_bytes]

    def _issue(self, i, job, blocking):
        """the communication step of picture i; False if it cannot be ordered yet (blocking=False) - nothing was issued then"""
        import torch.distributed as dist
        pl, owner = self.plans[i], self.owners[i]
        view = self._slot_view(pl.slot)
        if owner == self.rank:
            if not self.rec.stream_wait_job(job, self.rt.stream_ptr(), blocking):
                return False
            if self.transfer == "broadcast":
                self.rt.broadcast(view, owner)
            else:
                self.rt.transfer([dist.P2POp(dist.isend, view, r) for r in self.deps[i]])
            self.rec.slot_external_event(pl.slot, self.rt.event_ptr(), writes=False)
            self.trace.append(("send", i))
            self.bytes_sent += self.slot_bytes * len(self.deps[i])
        else:
            if not self.rec.stream_wait_slot(pl.slot, self.rt.stream_ptr(), blocking):
                return False
            if self.transfer == "broadcast":
                self.rt.broadcast(view, owner)
            else:
                self.rt.transfer([dist.P2POp(dist.irecv, view, owner)])
            self.rec.slot_external_event(pl.slot, self.rt.event_ptr(), writes=True)
            self.trace.append(("recv", i))
        self.n_bcast += 1
        return True

    def _pump(self, blocking=False):
        while self.fifo:
            i, job = self.fifo[0]
            if not self._issue(i, job, blocking):
                return
            self.fifo.pop(0)

    def run(self, descs, i0=0, i1=None):
        """pictures [i0, i1) of the plan (default: all).  descs[i]: description of plans[i] for the pictures this rank owns (None elsewhere is
        fine).  Returns {picture index: job} of the pictures reconstructed here; everything of the range is complete on return, the DPB state
        carries over to the next call."""
        jobs = {}
        for i in range(i0, len(self.plans) if i1 is None else i1):
            pl = self.plans[i]
            mine = self.owners[i] == self.rank
            ref_slots = [slot for lst in (pl.ref_slots or ([], [])) for (slot, _) in lst]
            if mine:
                # a reference slot (or the slot this picture overwrites) that a queued receive / send still has to touch: that step must be
                # issued first - its event is what orders this picture behind it.  Only then does the host wait (for a picture to be handed
                # to the device, never for the device)
                touched = set(ref_slots + [pl.slot])
                if any(self.plans[k].slot in touched for k, _ in self.fifo):
                    before = len(self.fifo)
                    self._pump(False)
                    if any(self.plans[k].slot in touched for k, _ in self.fifo):
                        self.trace.append(("host_wait", i))
                        while any(self.plans[k].slot in touched for k, _ in self.fifo):
                            k, job = self.fifo[0]
                            self._issue(k, job, True)
                            self.fifo.pop(0)
                else:
                    self._pump(False)
                job = self.rec.decompress_picture(descs[i])
                jobs[i] = job
                self.trace.append(("submit", i))
            if self.need[i] and self.world > 1 and (mine or self.rank in self.deps[i]):
                self.fifo.append((i, jobs.get(i)))
                self._pump(False)
        self._pump(True)
        # order matters: the collective's stream drains (its events complete), the back-end syncs (and forgets the complete external events),
        # only then are the events destroyed - the slots of the last pictures still name them until the back-end has looked
        if self.rt is not None:
            self.rt.finish()
        self.rec.sync()
        if self.rt is not None:
            self.rt.release()
        return jobs
