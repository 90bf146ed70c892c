"""CPU, world_size 2 over gloo: the N > 1 path of bench.py / vvdec_amd.parallel.

The distributed part of the path is control plane only (segments are independent: no data-path collective), so what must
hold for N > 1 is: every rank reconstructs exactly its share of the segments, the gathered per-picture MD5s are identical
to a single-process run, and the timing reduction is a max over ranks.  The reconstruction itself is done by the CPU oracle
here (the checker standing in for the GPU back-end, which cannot run without a device)."""
import os
import socket
import subprocess
import sys
import json
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import refdrv
from vvdec_amd import abi, synth, stream, parallel

W, H, GOP, NSEG = 128, 64, 4, 3
TOOLS = abi.TOOL_SAO_LUMA | abi.TOOL_ALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST

def reconstruct_segment(seg):
    plans, _ = stream.ra_plan(GOP + 1, gop=GOP, seed_poc0_is_external=False)
    dpb, out = {{}}, []
    for pl in plans:
        d = synth.picture_for_plan(pl, W, H, seed=parallel.segment_seed(1234, seg), tool_flags=TOOLS, log2_ctu=6, p_intra=0.2)
        planes = refdrv.oracle_reconstruct(d, dpb)
        dpb[pl.slot] = planes
        out.append((pl.poc, parallel.picture_md5(planes)))
    return out

rank, world, _ = parallel.init(backend="gloo")
parallel.barrier()
t0 = time.perf_counter()
res = parallel.reconstruct_segments(NSEG, reconstruct_segment, rank, world)
dt = time.perf_counter() - t0 + (0.25 if rank == 1 else 0.0)      # rank 1 pretends to be slower
tmax = parallel.max_over_ranks(dt)
mine = parallel.segments_for_rank(NSEG, rank, world)
print("RESULT " + json.dumps(dict(rank=rank, world=world, mine=mine, res=res, dt=dt, tmax=tmax)))
import torch.distributed as dist
if dist.is_initialized():
    dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            raise
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][0][7:]))
    return outs


def test_two_ranks_equal_one_rank(built, tmp_path):
    one = _run(1, tmp_path)[0]
    two = _run(2, tmp_path)
    assert len(one["res"]) == 3 * 5
    assert two[0]["mine"] == [0, 2] and two[1]["mine"] == [1]
    for r in two:
        assert r["res"] == one["res"], "gathered MD5s differ from the single-process run"
        assert abs(r["tmax"] - max(t["dt"] for t in two)) < 1e-9
    # different segments are different content (different seeds), same POC structure
    md5s = {(s, poc): m for (s, poc, m) in one["res"]}
    assert md5s[(0, 0)] != md5s[(1, 0)]
