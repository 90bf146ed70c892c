"""CPU: host-side logic — decode-order/DPB plan of the synthetic stream, generator invariants, segment sharding."""
import numpy as np
import pytest

from vvdec_amd import abi, synth, stream, parallel


@pytest.mark.parametrize("gop,frames", [(4, 5), (8, 17), (16, 33), (32, 65)])
def test_ra_plan_is_decodable(gop, frames):
    for ext in (True, False):
        plans, nslots = stream.ra_plan(frames, gop=gop, seed_poc0_is_external=ext)
        assert sorted(p.poc for p in plans) == list(range(1 if ext else 0, frames))
        live = {0: 0} if ext else {}                 # slot -> poc currently stored there
        for p in plans:
            for lst, pocs in zip(p.ref_slots, (p.l0, p.l1)):
                assert [poc for (_, poc) in lst] == pocs
                for (slot, poc) in lst:
                    assert live.get(slot) == poc, "POC %d reads POC %d from slot %d which holds %r" % (p.poc, poc, slot, live.get(slot))
                    assert slot != p.slot, "a picture must not be written over one of its own references"
            assert 0 <= p.slot < nslots
            live[p.slot] = p.poc
        assert nslots <= 8                           # hierarchical-B needs log2(gop) + a few pictures, never the whole GOP
        assert plans[0].slice_type == (abi.SLICE_B if ext else abi.SLICE_I)


@pytest.mark.parametrize("pool,lookahead", [(0, 0), (24, 0), (24, 8), (0, 5)])
def test_ra_plan_with_periodic_irap(pool, lookahead):
    """IRAP every 64 pictures, optionally submitted ahead of its decoding-order position: every reference precedes its user in the
    submission order, key pictures do not reference across the IRAP, no slot is overwritten while a later picture still reads it"""
    plans, nslots = stream.ra_plan(161, gop=16, seed_poc0_is_external=False, pool=pool, intra_period=64, irap_lookahead=lookahead)
    assert sorted(p.poc for p in plans) == list(range(161))
    iraps = [p.poc for p in plans if p.slice_type == 2]
    assert iraps == [0, 64, 128]
    pos = {p.poc: i for i, p in enumerate(plans)}
    content = {}
    for i, p in enumerate(plans):
        for lst in p.ref_slots:
            for slot, poc in lst:
                assert pos[poc] < i and content.get(slot) == poc, "POC %d reads POC %d from slot %d" % (p.poc, poc, slot)
        if p.slice_type != 2 and p.layer == 0:
            irap = max(q for q in iraps if q <= p.poc)
            assert all(r >= irap for r in p.l0 + p.l1)
        content[p.slot] = p.poc
        assert p.slot < nslots
    if lookahead:
        base, _ = stream.ra_plan(161, gop=16, seed_poc0_is_external=False, pool=pool, intra_period=64)
        bpos = {p.poc: i for i, p in enumerate(base)}
        assert all(pos[q] == max(1, bpos[q] - lookahead) for q in iraps[1:])


def test_ra_plan_layers():
    plans, _ = stream.ra_plan(17, gop=16)
    by_layer = {}
    for p in plans:
        by_layer.setdefault(p.layer, []).append(p.poc)
    assert [len(by_layer[l]) for l in sorted(by_layer)] == [1, 1, 2, 4, 8]
    assert all(not p.is_ref for p in plans if p.layer == 4) and all(p.is_ref for p in plans if p.layer < 4)


@pytest.mark.parametrize("W,H,l2", [(256, 128, 7), (200, 136, 6), (136, 72, 5)])
def test_generator_invariants(built, W, H, l2):
    tools = abi.TOOL_SAO_LUMA | abi.TOOL_ALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    for pl in plans[:3]:
        d = synth.picture_for_plan(pl, W, H, seed=77, tool_flags=tools, log2_ctu=l2, p_intra=0.3)
        d2 = synth.picture_for_plan(pl, W, H, seed=77, tool_flags=tools, log2_ctu=l2, p_intra=0.3)
        same = lambda a, b: len(a) == len(b) and all(np.array_equal(a[n], b[n]) for n in a.dtype.names)      # field-wise: struct padding is not data
        assert same(d.cu, d2.cu) and same(d.tu, d2.tu) and np.array_equal(d.coef, d2.coef) and same(d.motion, d2.motion), "generator must be deterministic"
        # CUs tile the picture exactly once
        cover = np.zeros((H, W), np.int32)
        for cu in d.cu:
            cover[cu["y"]:cu["y"] + cu["h"], cu["x"]:cu["x"] + cu["w"]] += 1
        assert (cover == 1).all()
        # CTU raster order, ctu_first_cu consistent
        ctu = 1 << l2
        ids = [(int(cu["y"]) // ctu) * d.ctus_x + int(cu["x"]) // ctu for cu in d.cu]
        assert ids == sorted(ids)
        assert d.ctu_first_cu[0] == 0 and d.ctu_first_cu[-1] == len(d.cu)
        # TUs belong to their CU, level offsets stay inside the packed stream
        for i, cu in enumerate(d.cu):
            for t in range(cu["first_tu"], cu["first_tu"] + cu["num_tu"]):
                tu = d.tu[t]
                assert tu["cu"] == i
                assert cu["x"] <= tu["x"] and tu["x"] + tu["w"] <= cu["x"] + cu["w"]
                for c in range(3):
                    if (tu["cbf"] >> c) & 1:
                        n = (int(tu["max_scan_x"][c]) + 1) * (int(tu["max_scan_y"][c]) + 1)
                        assert int(tu["coef_off"][c]) + n <= len(d.coef)
            if pl.slice_type == abi.SLICE_I:
                assert cu["pred_mode"] == abi.PRED_INTRA
        # levels are conformant-stream-like: no int16 overflow after dequantisation is provoked
        assert np.abs(d.coef.astype(np.int32)).max() < 32768


def test_segment_sharding_partitions_the_stream():
    for world in (1, 2, 3, 4, 8):
        for nseg in (1, 2, 7, 8, 64):
            seen = []
            for r in range(world):
                mine = parallel.segments_for_rank(nseg, r, world)
                assert len(mine) in (nseg // world, nseg // world + 1)
                seen += mine
            assert sorted(seen) == list(range(nseg))
    assert parallel.segment_seed(1234, 0) == 1234 and parallel.segment_seed(1234, 3) != parallel.segment_seed(1234, 2)


def test_picture_md5_layout():
    y = np.arange(8, dtype=np.uint16).reshape(2, 4)
    cb = np.array([[1]], np.uint16)
    cr = np.array([[2]], np.uint16)
    import hashlib
    assert parallel.picture_md5([y, cb, cr]) == hashlib.md5(y.tobytes() + cb.tobytes() + cr.tobytes()).hexdigest()


def _edge_reach(l, dr, pos, ctu):
    """(written P, written Q, read P, read Q) sample counts of one luma edge segment; mirrors xEdgeFilterLuma (LoopFilter.cpp:1464)."""
    if not (int(l["bs"]) & 3):
        return None
    v = int(l["side_max_filt_length"])
    lp, lq = (v >> 4) & 7, v & 7
    pl, ql = lp > 3, lq > 3
    if dr == 1 and pos % ctu == 0:
        pl = False
    w = 3 if (lp > 2 and lq > 2) else 2 if (lp > 1 and lq > 1) else 1
    r = 4 if (lp > 2 and lq > 2) else 3
    wp, wq, rp, rq = w, w, r, r
    if pl or ql:
        wp, wq = max(wp, lp if pl else 3), max(wq, lq if ql else 3)
        rp, rq = max(rp, lp + 1 if pl else 4), max(rq, lq + 1 if ql else 4)
    return wp, wq, rp, rq


def test_luma_edges_of_one_direction_are_independent(built):
    """k_deblock filters all edges of one direction in one launch.  That needs the sample sets of neighbouring edges to be disjoint.
    Checked on generated edge tables with affine / SbTMVP / SBT / small-CU content.  (A 7-sample P side right behind a sub-block edge of an
    SbTMVP CU, which the kernels order themselves, was an artefact of tables derived without the affine flag of sub-block merge CUs.)"""
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    tools = abi.TOOL_SAO_LUMA | abi.TOOL_ALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS
    ordered = 0
    for pl in plans:
        d = synth.picture_for_plan(pl, 416, 240, seed=162, tool_flags=tools, log2_ctu=7, p_intra=0.15, p_affine=0.2, p_sbtmvp=0.3, p_sbt=0.2, p_geo=0.1, p_ciip=0.1)
        ctu = 1 << d.hdr.log2_ctu
        for dr in range(2):
            lf = d.lfp[dr].reshape(d.h4, d.w4)
            lines = lf if dr == 0 else lf.T
            for line in lines:
                edges = [(b * 4, _edge_reach(l, dr, b * 4, ctu), l) for b, l in enumerate(line) if int(l["bs"]) & 3]
                for (e0, r0, l0), (e1, r1, l1) in zip(edges, edges[1:]):
                    assert not (e0 + r0[1] - 1 >= e1 - r1[2] or e1 - r1[0] <= e0 + r0[3] - 1), (pl.poc, dr, e0, e1)
                    ordered += 1
    assert ordered > 1000       # (pairs of neighbouring edges looked at)


def test_no_kernel_spills_and_the_resident_tile_counts_hold():
    """what round 6 relies on, read off the compiler (no GPU): no kernel of the library uses scratch memory (a spilled register or a record indexed per lane is a memory
    round trip on the critical path of every wavefront), and the kernels whose launches are meant to be resident at once keep the registers and the LDS for it -
    eight wavefronts per SIMD for the deblocking tiles, the DMVR sub-blocks, the affine tiles, the <= 32 transform classes and the edge-parameter cells; 20.5 KB of LDS
    at most for the horizontal deblocking tile and the affine tile (eight workgroups of four wavefronts per compute unit), 10 KB for a DMVR sub-block (sixteen)"""
    import os, re, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vvdec_amd", "csrc")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-scalarize-global-loads=false", "-w", "-c", "--cuda-device-only", "-x", "hip",
                          "vvr_kernels.hip", "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], cwd=src, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1); kernels[name] = {}
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and name:
                kernels[name][key] = int(m.group(1))
    assert len(kernels) >= 25, sorted(kernels)
    spilled = {k: v["scratch"] for k, v in kernels.items() if v.get("scratch")}
    assert not spilled, spilled

    def one(prefix):
        ks = [k for k in kernels if prefix in k]
        assert ks, prefix
        return [kernels[k] for k in ks]
    for prefix in ("k_deblock_tile", "k_mc_dmvr", "k_mc_affine", "k_itransILi16", "k_lf_init"):
        for r in one(prefix):
            assert r["occ"] == 8 and r["vgpr"] <= 64, (prefix, r)
    assert one("k_itransILi32")[0]["vgpr"] <= 64
    assert one("k_deblock_tileILb0ELi1")[0]["lds"] <= 20480 and one("k_mc_affine")[0]["lds"] <= 20480 and one("k_mc_dmvr")[0]["lds"] <= 10240
