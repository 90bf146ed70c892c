"""CPU: the host half of the intra-stage scheduler (vvdec_amd/csrc/vvr_api.cpp: work lists, units, dependency graph, ticket order),
compiled against a stand-in HIP runtime (tests/hoststub) so that it runs without a GPU.  Nothing is reconstructed here: the
tests check the tables the kernel would be handed.

Properties checked on generated pictures (I / B, every tool, IBC, dual tree, 4xN CUs, LMCS chroma scaling):
  * forward progress: a unit only waits for units with a lower ticket (workgroups take tickets in order, so a waiting workgroup's
    producers have always started), at most VVR_INTRA_MAX_DEPS of them;
  * the units partition the block list; blocks sit in their unit's (component, CTU);
  * soundness: wherever a block reads samples another intra-stage block produces (the row above / column left of it, the
    co-located luma of a CCLM block, the reference block of an IBC block), the producer is either an earlier block of the same
    unit or belongs to a unit the consumer's unit (transitively) waits for;
  * coverage: every cell of an intra / IBC / CIIP CU is produced by exactly the blocks of the list.
"""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

from vvdec_amd import abi, synth, stream

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hoststub", "vvr_host_stub.cpp")
LIB = os.path.join(HERE, "hoststub", "libvvr_hoststub.so")
API = os.path.join(os.path.dirname(HERE), "vvdec_amd", "csrc", "vvr_api.cpp")
HIP_INC = "/opt/rocm/include"

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(HIP_INC, "hip", "hip_runtime_api.h")), reason="HIP headers not installed")

UNIT_DT = np.dtype([("ent", "<u4"), ("i0", "<u4"), ("i1", "<u4"), ("bbox", "<u4"), ("ndeps", "<u4"), ("deps", "<u4", (26,)), ("iA", "<u4")])
ITEM_DT = np.dtype([("x", "<u2"), ("y", "<u2"), ("lw", "u1"), ("lh", "u1"), ("mode", "u1"), ("flags", "u1"), ("nTL", "u1"), ("nA", "u1"), ("nL", "u1"), ("comp", "u1"), ("tu", "<u4")])
MODE_RESI_ADD, MODE_IBC = 255, 254
TOOLS = (abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST |
         abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF)


def build_stub():
    """(re)build the product's host code against the stand-in runtime when a source is newer -> path of the library"""
    deps = [SRC, API] + [os.path.join(os.path.dirname(API), f) for f in ("vvr_device.h", "vvr_host.h", "vvr_prepare.cpp", "vvr_output.inc", "vvr_lf_init.h")] + [os.path.join(os.path.dirname(HERE), "include", "vvr.h")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(p) for p in deps):
        # (several test processes may get here at once - pytest -n: build under a private name, then rename into place)
        tmp = "%s.%d.tmp" % (LIB, os.getpid())
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-Wl,-Bsymbolic", "-I" + HIP_INC, "-D__HIP_PLATFORM_AMD__", "-DVVR_DEV_ENV", "-w", SRC, "-o", tmp])
        os.replace(tmp, LIB)
    return LIB


@pytest.fixture(scope="module")
def stub():
    L = C.CDLL(build_stub())
    L.vvr_last_error.restype = C.c_char_p
    L.vvr_last_error.argtypes = [C.c_void_p]
    L.vvt_sizeof.restype = C.c_size_t
    L.vvt_sync_capacity.restype = C.c_size_t
    L.vvt_sync_capacity.argtypes = [C.c_void_p, C.c_int]
    L.vvr_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.vvr_submit_prepared.argtypes = [C.c_void_p, C.c_void_p]
    L.vvr_free_prepared.argtypes = [C.c_void_p, C.c_void_p]
    L.vvr_destroy.argtypes = [C.c_void_p]
    L.vvr_sync.argtypes = [C.c_void_p]
    L.vvt_intra_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    assert L.vvt_sizeof(0) == UNIT_DT.itemsize and L.vvt_sizeof(1) == ITEM_DT.itemsize
    return L


class Ctx:
    def __init__(self, L, W, H, nslots, log2_ctu=7, bit_depth=10, chroma_format=1, streams=1, leaf=False, by_level=True):
        """leaf: pictures with scattered intra blocks take the one-wavefront-per-block path (the product's default); False: the CTU-tile path with
        units for every picture, what most tests of this file are about (the library reads VVR_INTRA_LEAF when the context is created)"""
        self.L = L
        os.environ["VVR_INTRA_LEAF"] = "1" if leaf else "0"
        os.environ["VVR_LEAF_BY_LEVEL"] = "1" if by_level else "0"
        cfg = abi.Config()
        cfg.abi_version = abi.VVR_ABI_VERSION
        cfg.device, cfg.max_width, cfg.max_height = 0, W, H
        cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = chroma_format, bit_depth, log2_ctu
        cfg.num_slots, cfg.num_streams = nslots, streams
        self.ctx = C.c_void_p()
        assert L.vvr_create(C.byref(cfg), C.byref(self.ctx)) == abi.VVR_OK
        del os.environ["VVR_INTRA_LEAF"], os.environ["VVR_LEAF_BY_LEVEL"]

    def prepare(self, d):
        p = d.c()
        h = C.c_void_p()
        rc = self.L.vvr_prepare(self.ctx, C.byref(p), C.byref(h))
        assert rc == abi.VVR_OK, self.L.vvr_last_error(self.ctx).decode()
        return h

    def tables(self, h):
        up, ip, nu, ni = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
        assert self.L.vvt_intra_tables(h, C.byref(up), C.byref(nu), C.byref(ip), C.byref(ni)) == 0
        units = np.frombuffer((C.c_char * (UNIT_DT.itemsize * nu.value)).from_address(up.value), UNIT_DT).copy() if nu.value else np.zeros(0, UNIT_DT)
        items = np.frombuffer((C.c_char * (ITEM_DT.itemsize * ni.value)).from_address(ip.value), ITEM_DT).copy() if ni.value else np.zeros(0, ITEM_DT)
        return units, items

    def resi_tables(self, h):
        """the residual-add blocks (k_resi_add) of a prepared picture and how the stage is launched: (blocks, luma units that take the first tickets)"""
        p, n = C.c_void_p(), C.c_size_t()
        self.L.vvt_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        assert self.L.vvt_table(h, 9, C.byref(p), C.byref(n)) == 0
        resi = np.frombuffer(C.string_at(p.value, n.value), ITEM_DT).copy() if n.value else np.zeros(0, ITEM_DT)
        nl, w0, w1 = C.c_int(), C.c_int(), C.c_int()
        self.L.vvt_intra_launches.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        self.L.vvt_intra_launches(h, C.byref(nl), C.byref(w0), C.byref(w1))
        return resi, nl.value, w0.value, w1.value

    def close(self):
        self.L.vvr_destroy(self.ctx)


def _check_resi_add(d, units, resi, num_luma, wg0, wg1):
    """the scaled chroma residuals of inter blocks (LMCS chroma residual scaling) are not blocks of the stage's dependency graph: k_resi_add adds
    them between the luma units and the chroma units - every such transform block is in its list exactly once, and with the list present the
    luma units hold the first tickets"""
    h = d.hdr
    l2, ctusX = h.log2_ctu, (h.width + (1 << h.log2_ctu) - 1) >> h.log2_ctu
    want = set()
    if (h.tool_flags & abi.TOOL_LMCS) and (h.tool_flags & abi.TOOL_LMCS_CSCALE) and h.chroma_format:
        for cu in d.cu:
            if cu["pred_mode"] != abi.PRED_INTER or not (int(cu["flags"]) & abi.CU_ROOT_CBF):
                continue
            if d.slices is not None:
                f = int(d.slices["tool_flags"][d.ctu_slice[(int(cu["y"]) >> l2) * ctusX + (int(cu["x"]) >> l2)]])
                if not (f & abi.TOOL_LMCS) or not (f & abi.TOOL_LMCS_CSCALE):
                    continue
            if (int(cu["flags"]) & abi.CU_CIIP) and int(cu["w"]) != 4:
                continue            # blended in the stage itself
            for t in range(int(cu["first_tu"]), int(cu["first_tu"]) + int(cu["num_tu"])):
                tu = d.tu[t]
                for comp in (1, 2):
                    if (int(tu["comp_mask"]) & (1 << comp)) and (((int(tu["cbf"]) >> comp) & 1) or int(tu["joint_cbcr"])) and (int(tu["w"]) >> 1) * (int(tu["h"]) >> 1) > 4:
                        want.add((comp, int(tu["x"]) >> 1, int(tu["y"]) >> 1, int(tu["w"]) >> 1, int(tu["h"]) >> 1))
    got = [(int(r["comp"]) & 3, int(r["x"]), int(r["y"]), 1 << int(r["lw"]), 1 << int(r["lh"])) for r in resi]
    assert len(got) == len(set(got)) and set(got) == want, "residual-add blocks: %d listed, %d expected" % (len(got), len(want))
    assert all(int(r["mode"]) == MODE_RESI_ADD and (int(r["flags"]) & 9) == 9 for r in resi)
    comps = (units["ent"] >> 24) & 3
    if len(resi):
        assert (comps[:num_luma] == 0).all() and (comps[num_luma:] != 0).all(), "with residual-add blocks the luma units take the first tickets"
        assert (num_luma == 0) == (wg0 == 0) and (num_luma == len(units)) == (wg1 == 0) and wg0 <= max(num_luma, 0) and wg1 <= len(units) - num_luma
    else:
        assert num_luma == 0 and wg1 == 0
    return len(got)


def _check_tables(d, units, items):
    h = d.hdr
    W, H, l2 = h.width, h.height, h.log2_ctu
    w4, h4 = (W + 3) >> 2, (H + 3) >> 2
    ctusX = (W + (1 << l2) - 1) >> l2
    ncomp = 3 if h.chroma_format else 1
    nU, nI = len(units), len(items)
    # ---- forward progress
    for t in range(nU):
        nd = int(units["ndeps"][t])
        assert nd <= 26
        assert all(int(x) < t for x in units["deps"][t][:nd]), "unit %d waits for a unit with a higher ticket" % t
    # ---- the units partition the block list; blocks sit in their unit's (component, CTU)
    unit_of = np.full(nI, -1, np.int64)
    for t in range(nU):
        i0, i1, iA = int(units["i0"][t]), int(units["i1"][t]), int(units["iA"][t])
        assert i0 <= i1 <= nI and iA in (i0, i1)
        assert (unit_of[i0:i1] == -1).all(), "blocks in two units"
        unit_of[i0:i1] = t
        comp, ctu = (int(units["ent"][t]) >> 24) & 3, int(units["ent"][t]) & 0xffffff
        S = (1 << l2) >> (1 if comp else 0)
        ox, oy = (ctu % ctusX) * S, (ctu // ctusX) * S
        it = items[i0:i1]
        assert ((it["comp"] & 3) == comp).all()
        assert ((it["x"] >= ox) & (it["x"] < ox + S) & (it["y"] >= oy) & (it["y"] < oy + S)).all(), "block outside its unit's CTU"
        if i1 > i0:
            assert ((it["mode"] == MODE_RESI_ADD).all() and comp > 0) if iA == i1 else not (it["mode"] == MODE_RESI_ADD).any()
    assert (unit_of >= 0).all(), "block without a unit"
    # ---- transitive producers of every unit (bit sets)
    anc = [0] * nU
    for t in range(nU):
        a = 0
        for x in units["deps"][t][:int(units["ndeps"][t])]:
            a |= anc[int(x)] | (1 << int(x))
        anc[t] = a
    # ---- who produces which cell, decode order of the cells
    order = np.full((2, h4, w4), 1 << 30, np.int64)
    for cu in d.cu:
        for t in range(int(cu["first_tu"]), int(cu["first_tu"]) + int(cu["num_tu"])):
            tu = d.tu[t]
            for chn, m in ((0, 1), (1, 6)):
                if not (int(tu["comp_mask"]) & m):
                    continue
                x0, y0, ww, hh = (int(cu[k]) for k in "xywh") if (chn == 1 and cu["isp_mode"]) else (int(tu[k]) for k in "xywh")
                order[chn, y0 >> 2:(y0 + hh + 3) >> 2, x0 >> 2:(x0 + ww + 3) >> 2] = t
    prod = np.full((ncomp, h4, w4), -1, np.int64)
    for i in range(nI):
        it = items[i]
        icomp = int(it["comp"]) & 3
        cs = 1 if icomp else 0
        x0, y0, ww, hh = int(it["x"]) << cs, int(it["y"]) << cs, (1 << int(it["lw"])) << cs, (1 << int(it["lh"])) << cs
        # a block of more than 256 samples (ordinary prediction modes) comes as 2, 4 or 8 items, one band of rows each (one wavefront per band)
        part, lparts = (int(it["nTL"]) >> 1) & 7, (int(it["nTL"]) >> 4) & 3
        samples = (ww * hh) >> (2 * cs)
        assert part < (1 << lparts) and (lparts == 0 or (samples >> lparts) == 256 or (lparts == 3 and (samples >> lparts) > 256))
        ordinary = int(it["mode"]) <= 66 and (icomp or ((int(it["flags"]) & 8) == 0 and (int(it["flags"]) & 6) != 6))
        assert (samples >> lparts) <= 512 or not ordinary, "an ordinary block of more than 512 samples in one item"
        assert lparts or samples <= 256 or not ordinary, "an ordinary block of more than 256 samples that is not split"
        if lparts:
            assert i - part >= 0 and all(int(items[i - part + e]["x"]) == int(it["x"]) and int(items[i - part + e]["y"]) == int(it["y"]) and ((int(items[i - part + e]["nTL"]) >> 1) & 7) == e for e in range(1 << lparts)), "the bands of a block are consecutive items"
            assert unit_of[i - part] == unit_of[i - part + (1 << lparts) - 1]
            # a band reads what the whole block reads and nothing of the bands before it
            assert (int(it["comp"]) >> 2) == min(63, (int(items[i - part]["comp"]) >> 2) + part)
            y0 += part * (hh >> lparts)
            hh >>= lparts
        sub = prod[icomp, y0 >> 2:(y0 + hh + 3) >> 2, x0 >> 2:(x0 + ww + 3) >> 2]
        isp_narrow = (int(it["flags"]) & 6) == 6 and not icomp and min(ww, hh) < 4
        assert isp_narrow or (sub == -1).all(), "two blocks produce one cell"
        sub[...] = i

    def ordered_before(j, i):
        if j == i or j < 0:
            return True
        uj, ui = int(unit_of[j]), int(unit_of[i])
        # inside a unit: block i starts when every block of the unit before i - indep is done (the kernel's progress counters)
        return (j < i - (int(items[i]["comp"]) >> 2)) if uj == ui else bool((anc[ui] >> uj) & 1)

    ctu4 = 1 << (l2 - 2)

    def same_slice_tile(cy, cx, by, bx):
        a, b = (cy // ctu4) * ctusX + cx // ctu4, (by // ctu4) * ctusX + bx // ctu4
        return (d.ctu_slice is None or d.ctu_slice[a] == d.ctu_slice[b]) and (d.ctu_tile is None or d.ctu_tile[a] == d.ctu_tile[b])

    def cell(comp, xc, yc):            # component sample -> cell of the 4x4 luma grid
        cs = 1 if comp else 0
        return (yc << cs) >> 2, (xc << cs) >> 2

    nchk = 0
    for i in range(nI):
        it = items[i]
        comp, mode = int(it["comp"]) & 3, int(it["mode"])
        if mode == MODE_RESI_ADD:
            continue
        cs, chn = (1 if comp else 0), (1 if comp else 0)
        unit = 4 >> cs
        x0, y0, ww, hh = int(it["x"]), int(it["y"]), 1 << int(it["lw"]), 1 << int(it["lh"])
        myo = order[(chn,) + cell(comp, x0, y0)]
        reads = []
        if mode == MODE_IBC:
            dx, dy = ((int(it["tu"]) & 0xffff) ^ 0x8000) - 0x8000, ((int(it["tu"]) >> 16) ^ 0x8000) - 0x8000
            for yy in list(range(0, hh, unit)) + [hh - 1]:
                for xx in list(range(0, ww, unit)) + [ww - 1]:
                    reads.append((comp, x0 + dx + xx, y0 + dy + yy, True))
        elif (int(it["flags"]) & 6) == 6 and not comp:
            pass                        # ISP partitions: their CU's reference line and the partition chain (inside one unit by construction)
        else:
            # what the kernel reads (k_intra's reference fill): the nA units above incl. above-right, the nL units left incl. below-left, the
            # corner - one line (luma: possibly a line 1 or 2 further out, multi-reference-line prediction)
            mrl = 0 if (comp or (int(it["flags"]) & 8)) else (int(it["flags"]) >> 4) & 3
            for k in range(0, int(it["nA"]) * unit, unit):
                reads.append((comp, x0 + k, y0 - 1 - mrl, True))
            for k in range(0, int(it["nL"]) * unit, unit):
                reads.append((comp, x0 - 1 - mrl, y0 + k, True))
            if int(it["nTL"]) & 1:
                reads.append((comp, x0 - 1 - mrl, y0 - 1 - mrl, True))
            if comp and 67 <= mode <= 69:
                for yy in range(0, 2 * hh, 4):
                    for xx in range(0, 2 * ww, 4):
                        reads.append((0, 2 * x0 + xx, 2 * y0 + yy, True))
        for (k, xc, yc, must) in reads:
            sh = 1 if k else 0
            if xc < 0 or yc < 0 or (xc << sh) >= W or (yc << sh) >= H:
                assert not must
                continue
            cy, cx = cell(k, xc, yc)
            if not must and order[1 if k else 0, cy, cx] >= myo:
                continue                # not reconstructed before the block: not available, not read
            if not must and not same_slice_tile(cy, cx, *cell(comp, x0, y0)):
                continue                # in another slice or tile: not available, not read
            if must and k == comp:
                assert order[1 if k else 0, cy, cx] < myo, "a sample the block reads is not reconstructed before the block"
            j = int(prod[k, cy, cx])
            assert ordered_before(j, i), "block %d (unit %d) reads block %d (unit %d) without waiting for it" % (i, unit_of[i], j, unit_of[j])
            nchk += j >= 0 and j != i
    # ---- coverage: the cells of intra / IBC / CIIP CUs are produced by blocks of the list
    for cu in d.cu:
        intra_stage = cu["pred_mode"] in (abi.PRED_INTRA, abi.PRED_IBC) or (int(cu["flags"]) & abi.CU_CIIP)
        if not intra_stage:
            continue
        x0, y0, ww, hh = (int(cu[k]) for k in "xywh")
        comps = [0] if cu["tree"] == abi.TREE_LUMA else [1, 2] if cu["tree"] == abi.TREE_CHROMA else list(range(ncomp))
        for k in comps:
            if k and (int(cu["flags"]) & abi.CU_CIIP) and ww == 4:
                continue                # 2-wide chroma of a 4-wide CIIP CU stays pure inter
            assert (prod[k, y0 >> 2:(y0 + hh) >> 2, x0 >> 2:(x0 + ww) >> 2] >= 0).all(), "CU cells without a block"
    return nchk


STREAMS = [
    ("i_b_all_tools", 416, 240, 5, 4, 501, TOOLS | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_intra=0.3, p_affine=0.1, p_geo=0.1, p_ciip=0.15, p_sbtmvp=0.1, p_cclm=0.3, p_mip=0.2, p_isp=0.2, p_sbt=0.1, p_jccr=0.2, p_coded_chroma=0.5)),
    ("ibc_lmcs_ctu64", 416, 240, 3, 2, 502, TOOLS | abi.TOOL_IBC | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(log2_ctu=6, p_ibc=0.5, p_intra=0.4, p_cclm=0.3, p_ciip=0.1, p_coded_chroma=0.5)),
    ("ibc_ctu32", 256, 192, 3, 2, 503, TOOLS | abi.TOOL_IBC, dict(log2_ctu=5, p_ibc=0.6, p_intra=0.5)),
    ("dual_tree_ibc", 256, 128, 3, 2, 504, TOOLS | abi.TOOL_IBC | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(dual_tree=2.0, p_ibc=0.4, p_split_scale=1.5, p_cclm=0.3, p_isp=0.2)),
    ("small_cus", 416, 240, 3, 2, 505, TOOLS | abi.TOOL_IBC | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(min_cu_log2=2, p_ibc=0.3, p_intra=0.4, p_split_scale=1.8, p_cclm=0.3, p_ciip=0.3)),
    ("1080p", 1920, 1080, 3, 2, 506, TOOLS | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_intra=0.15, p_cclm=0.1, p_ciip=0.03, p_isp=0.05, p_mip=0.05)),
]


@pytest.mark.parametrize("name,W,H,frames,gop,seed,tools,kw", STREAMS, ids=[s[0] for s in STREAMS])
def test_intra_stage_tables(stub, name, W, H, frames, gop, seed, tools, kw):
    kw = dict(kw)
    l2 = kw.pop("log2_ctu", 7)
    plans, nslots = stream.ra_plan(frames, gop=gop, seed_poc0_is_external=False)
    ctx = Ctx(stub, W, H, nslots, log2_ctu=l2)
    checked = resi_blocks = 0
    for pl in plans:
        d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, log2_ctu=l2, **kw)
        hnd = ctx.prepare(d)
        units, items = ctx.tables(hnd)
        checked += _check_tables(d, units, items)
        resi_blocks += _check_resi_add(d, units, *ctx.resi_tables(hnd))
        stub.vvr_free_prepared(ctx.ctx, hnd)
    ctx.close()
    assert checked > 0 and (resi_blocks > 0) == bool((tools & abi.TOOL_LMCS_CSCALE) and frames > 1)


MODE_CSFAC = 253


def _check_leaf_items(d, items, resi=None, by_level=True):
    """the item list of k_intra_leaf (vvr_intra_leaf.inc): luma blocks, the chroma-scaling factors of the VPDUs, Cb blocks, Cr blocks - sorted by level (1 + the
    highest level among the items whose cells the item polls; decoding order within a level), or, by_level False, each list in decoding order - either way
    everything an item reads of another item's output comes from an item BEFORE it (the kernel hands items out by ticket, in list order: whoever waits,
    waits for a wavefront that is running or done).  Checked: the order of the parts (decoding order), bands consecutive, ISP partitions consecutive, every
    cell of an intra / CIIP CU produced exactly once, the residual-add blocks complete, and for every sample an item reads (reference lines, the
    previous ISP partition, co-located luma and templates of CCLM, the luma a scaling factor is averaged over, the factor of a block's VPDU) that its
    producer - if the stage has one - precedes the item."""
    h = d.hdr
    W, H, l2 = h.width, h.height, h.log2_ctu
    w4, h4 = (W + 3) >> 2, (H + 3) >> 2
    ncomp = 3 if h.chroma_format else 1
    nI = len(items)
    comp = items["comp"] & 3
    mode = items["mode"]
    assert ((items["comp"] >> 2) == 0).all()
    # ---- parts in order: luma, factors, Cb, Cr
    kind = np.where(mode == MODE_CSFAC, 1, np.where(comp == 0, 0, comp + 1))
    assert by_level or (np.diff(kind.astype(np.int64)) >= 0).all(), "items are not in the order luma, factors, Cb, Cr"
    vl = min(6, l2)
    vpdusX = (W + (1 << vl) - 1) >> vl
    fac_at = {int(items["tu"][i]): i for i in range(nI) if mode[i] == MODE_CSFAC}
    assert len(fac_at) == int((mode == MODE_CSFAC).sum()), "a VPDU's factor twice"
    prod = np.full((ncomp, h4, w4), -1, np.int64)
    isp_first = {}
    # the residual-add blocks are grouped by VPDU; the VPDU's factor item adds them (x | y << 16 = first, lw | lh << 8 = count): every block once, in its VPDU
    assert not (mode == MODE_RESI_ADD).any(), "residual-add blocks are no items of their own"
    if resi is not None:
        seen = np.zeros(len(resi), np.int64)
        for vp, i in fac_at.items():
            f, n = int(items["x"][i]) | (int(items["y"][i]) << 16), int(items["lw"][i]) | (int(items["lh"][i]) << 8)
            assert f + n <= len(resi)
            seen[f:f + n] += 1
            for r in resi[f:f + n]:
                k = int(r["comp"]) & 3
                x0, y0, ww, hh = int(r["x"]) << 1, int(r["y"]) << 1, (1 << int(r["lw"])) << 1, (1 << int(r["lh"])) << 1
                assert (y0 >> vl) * vpdusX + (x0 >> vl) == vp, "a residual-add block in another VPDU's group"
                sub = prod[k, y0 >> 2:(y0 + hh + 3) >> 2, x0 >> 2:(x0 + ww + 3) >> 2]
                assert (sub == -1).all()
                sub[...] = i
        assert (seen == 1).all(), "a residual-add block that no factor item (or more than one) adds"
    for i in range(nI):
        it = items[i]
        if mode[i] == MODE_CSFAC:
            continue
        k = int(comp[i]); cs = 1 if k else 0
        x0, y0, ww, hh = int(it["x"]) << cs, int(it["y"]) << cs, (1 << int(it["lw"])) << cs, (1 << int(it["lh"])) << cs
        part, lparts = (int(it["nTL"]) >> 1) & 7, (int(it["nTL"]) >> 4) & 3
        samples = (ww * hh) >> (2 * cs)
        is_isp = k == 0 and mode[i] <= 66 and (int(it["flags"]) & 6) == 6 and not (int(it["flags"]) & 8)
        ordinary = int(mode[i]) <= 66 and (k or ((int(it["flags"]) & 8) == 0 and not is_isp))
        assert part < (1 << lparts) and (lparts == 0 or ordinary)
        assert lparts or samples <= 256 or not ordinary, "an ordinary block of more than 256 samples that is not split"
        assert (samples >> lparts) <= 1024 or (k == 0 and (int(it["flags"]) & 8) and int(mode[i]) < MODE_CSFAC), "an item of more than 1024 samples that is no MIP block (the wavefront's LDS block holds 1024)"
        if lparts:
            assert i - part >= 0 and all(int(items[i - part + e]["x"]) == int(it["x"]) and int(items[i - part + e]["y"]) == int(it["y"]) and ((int(items[i - part + e]["nTL"]) >> 1) & 7) == e for e in range(1 << lparts)), "the bands of a block are consecutive items"
            y0 += part * (hh >> lparts); hh >>= lparts
        if is_isp:
            tu = int(it["tu"])
            if tu & 0xfff:
                # a later partition: predicted by the wavefront of the first one, which reaches it by walking the list
                assert i > 0 and (i - 1) in isp_first, "an ISP partition that does not follow a partition of its coding unit"
                isp_first[i] = isp_first[i - 1]
            else:
                isp_first[i] = i
        sub = prod[k, y0 >> 2:(y0 + hh + 3) >> 2, x0 >> 2:(x0 + ww + 3) >> 2]
        assert (is_isp and min(ww, hh) < 4) or (sub == -1).all(), "two items produce one cell"
        sub[...] = isp_first.get(i, i)      # (the cells of an ISP coding unit are cleared by its first partition's wavefront, after the last partition)
    nchk = 0
    cu_at = np.full((h4, w4), -1, np.int64)
    for ci, cu in enumerate(d.cu):
        if cu["tree"] != abi.TREE_CHROMA:
            cu_at[int(cu["y"]) >> 2:(int(cu["y"]) + int(cu["h"]) + 3) >> 2, int(cu["x"]) >> 2:(int(cu["x"]) + int(cu["w"]) + 3) >> 2] = ci

    def luma_cu_origin(x, y):
        cu = d.cu[int(cu_at[y >> 2, x >> 2])]
        return int(cu["x"]), int(cu["y"])
    for i in range(nI):
        it = items[i]
        m, k = int(mode[i]), int(comp[i])
        cs = 1 if k else 0
        unit = 4 >> cs
        x0, y0, ww, hh = int(it["x"]), int(it["y"]), 1 << int(it["lw"]), 1 << int(it["lh"])
        reads = []          # (component, x, y) in component samples
        flags = int(it["flags"])
        if m == MODE_CSFAC:
            vp = int(it["tu"])
            # (Reshape::calculateChromaAdjVpduNei: the column left of / the row above the CU at the VPDU's origin; here a superset: around the VPDU AND around every
            # CU origin up to a CTU further up / left is not known to the test - it checks the VPDU's own border, which the CU's border contains or precedes)
            vx, vy = (vp % vpdusX) << vl, (vp // vpdusX) << vl
            cux, cuy = luma_cu_origin(vx, vy)
            reads += [(0, cux - 1, yy) for yy in range(cuy, min(cuy + (1 << vl), H), 4)] if cux > 0 else []
            reads += [(0, xx, cuy - 1) for xx in range(cux, min(cux + (1 << vl), W), 4)] if cuy > 0 else []
            for (kc, xc, yc) in reads:
                j = int(prod[kc, yc >> 2, xc >> 2])
                assert j < i, "the factor item %d reads luma that item %d produces" % (i, j)
                nchk += j >= 0
            continue
        if k and (flags & 8):
            vp = ((y0 << 1) >> vl) * vpdusX + ((x0 << 1) >> vl)
            assert vp in fac_at and fac_at[vp] < i, "a block scales its residual with a factor no item before it computes"
        is_isp = k == 0 and m <= 66 and (flags & 6) == 6 and not (flags & 8)
        mrl = 0 if (k or (flags & 8) or is_isp) else (flags >> 4) & 3
        rx0, ry0 = x0, y0
        if is_isp:
            tu = int(it["tu"])
            rx0, ry0 = x0 - (tu & 63), y0 - ((tu >> 6) & 63)
        for kk in range(0, int(it["nA"]) * unit, unit):
            reads.append((k, rx0 + kk, ry0 - 1 - mrl))
        for kk in range(0, int(it["nL"]) * unit, unit):
            reads.append((k, rx0 - 1 - mrl, ry0 + kk))
        if int(it["nTL"]) & 1:
            reads.append((k, rx0 - 1 - mrl, ry0 - 1 - mrl))
        if k and 67 <= m <= 69:
            lm = int(it["tu"])
            top, left = lm & 0xff, (lm >> 8) & 0xff
            for yy in range(-4, 2 * max(hh, left), 4):
                for xx in range(-4, 2 * max(ww, top), 4):
                    if yy < 2 * hh or xx < 0:
                        if xx < 2 * ww or yy < 0:
                            reads.append((0, 2 * x0 + xx, 2 * y0 + yy))
        me = isp_first.get(i, i)
        for (kc, xc, yc) in reads:
            sh = 1 if kc else 0
            if xc < 0 or yc < 0 or (xc << sh) >= W or (yc << sh) >= H:
                continue
            j = int(prod[kc, (yc << sh) >> 2, (xc << sh) >> 2])
            if kc == k and j == me:
                continue            # (CCLM templates and reference lines never lie in the block itself; a later ISP partition reads its own coding unit)
            assert j < me, "item %d reads a cell item %d produces" % (i, j)
            nchk += j >= 0
    # ---- coverage
    for cu in d.cu:
        if not (cu["pred_mode"] == abi.PRED_INTRA or (int(cu["flags"]) & abi.CU_CIIP)):
            continue
        x0, y0, ww, hh = (int(cu[kx]) for kx in "xywh")
        comps = [0] if cu["tree"] == abi.TREE_LUMA else [1, 2] if cu["tree"] == abi.TREE_CHROMA else list(range(ncomp))
        for kc in comps:
            if kc and (int(cu["flags"]) & abi.CU_CIIP) and ww == 4:
                continue
            assert (prod[kc, y0 >> 2:(y0 + hh) >> 2, x0 >> 2:(x0 + ww) >> 2] >= 0).all(), "CU cells without an item"
    return nchk


@pytest.mark.parametrize("by_level", [True, False], ids=["by_level", "decoding_order"])
@pytest.mark.parametrize("name,W,H,frames,gop,seed,tools,kw", STREAMS, ids=[s[0] for s in STREAMS])
def test_intra_leaf_items(stub, name, W, H, frames, gop, seed, tools, kw, by_level):
    """pictures with scattered intra blocks (every picture that is not all intra CUs and holds no IBC CU): one wavefront per block, no units"""
    kw = dict(kw)
    l2 = kw.pop("log2_ctu", 7)
    plans, nslots = stream.ra_plan(frames, gop=gop, seed_poc0_is_external=False)
    ctx = Ctx(stub, W, H, nslots, log2_ctu=l2, leaf=True, by_level=by_level)
    stub.vvt_is_leaf.argtypes = [C.c_void_p]
    checked = leaf_pictures = reordered = 0
    for pl in plans:
        d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, log2_ctu=l2, **kw)
        hnd = ctx.prepare(d)
        units, items = ctx.tables(hnd)
        all_intra = all(cu["pred_mode"] == abi.PRED_INTRA for cu in d.cu) and d.hdr.slice_type == abi.SLICE_I
        any_ibc = any(cu["pred_mode"] == abi.PRED_IBC for cu in d.cu)
        want_leaf = not all_intra and not any_ibc and len(items) > 0
        assert bool(stub.vvt_is_leaf(hnd)) == want_leaf or len(items) == 0
        if stub.vvt_is_leaf(hnd):
            assert len(units) == 0
            resi = ctx.resi_tables(hnd)[0]
            _check_resi_add(d, np.zeros(0, UNIT_DT), resi, 0, 0, 0)
            checked += _check_leaf_items(d, items, resi, by_level)
            kind = np.where(items["mode"] == MODE_CSFAC, 1, np.where((items["comp"] & 3) == 0, 0, (items["comp"] & 3) + 1)).astype(np.int64)
            reordered += bool((np.diff(kind) < 0).any())
            leaf_pictures += 1
            assert stub.vvr_submit_prepared(ctx.ctx, hnd) >= 0 and stub.vvr_sync(ctx.ctx) == 0
            assert stub.vvt_last_leaf_items() == len(items)
        else:
            _check_tables(d, units, items)
            stub.vvr_free_prepared(ctx.ctx, hnd)
    ctx.close()
    assert (leaf_pictures > 0 and checked > 0) or name.startswith("ibc") or name in ("dual_tree_ibc", "small_cus")
    assert reordered == (leaf_pictures if by_level else 0), "the list by level interleaves luma and chroma items (every picture here has blocks that wait)"


def test_sync_buffer_grows_with_the_number_of_units(stub):
    """the per-lane ticket / flag buffer is sized for ordinary pictures and grows when a picture has more units"""
    W, H = 416, 240
    plans, nslots = stream.ra_plan(3, gop=2, seed_poc0_is_external=False)
    ctx = Ctx(stub, W, H, nslots)
    stub.vvt_shrink_sync.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    stub.vvt_shrink_sync(ctx.ctx, 0, 9)            # as if the context had been sized for one unit per CTU
    cap0 = stub.vvt_sync_capacity(ctx.ctx, 0)
    d = synth.picture_for_plan(plans[1], W, H, seed=507, tool_flags=TOOLS, p_intra=0.4, p_split_scale=1.6)
    hnd = ctx.prepare(d)
    units, _ = ctx.tables(hnd)
    assert len(units) + 1 > cap0          # the case the test is about
    assert stub.vvr_submit_prepared(ctx.ctx, hnd) >= 0
    assert stub.vvt_last_intra_launch() == len(units)
    assert stub.vvt_sync_capacity(ctx.ctx, 0) >= len(units) + 1
    stub.vvr_sync(ctx.ctx)
    stub.vvr_free_prepared(ctx.ctx, hnd)
    ctx.close()


def test_checker_detects_a_missing_dependency(stub):
    """the soundness check is not vacuous: removing the producers of the units of a picture is noticed"""
    W, H = 416, 240
    plans, nslots = stream.ra_plan(3, gop=2, seed_poc0_is_external=False)
    ctx = Ctx(stub, W, H, nslots)
    d = synth.picture_for_plan(plans[1], W, H, seed=508, tool_flags=TOOLS | abi.TOOL_IBC, p_intra=0.4, p_ibc=0.4, p_cclm=0.3)
    hnd = ctx.prepare(d)
    units, items = ctx.tables(hnd)
    assert _check_tables(d, units, items) > 0
    broken = units.copy()
    broken["ndeps"][:] = 0
    with pytest.raises(AssertionError, match="without waiting"):
        _check_tables(d, broken, items)
    stub.vvr_free_prepared(ctx.ctx, hnd)
    ctx.close()


def test_blocks_of_a_unit_that_do_not_read_from_each_other(stub):
    """the kernel predicts a unit's blocks with several wavefronts; a block starts when every block up to index - indep - 1 of its unit is done
    (IntraItem::comp bits 2..7).  The host finds independent blocks (B pictures: the clusters that share a unit), and the checker notices when a
    block claims more independence than it has"""
    W, H = 416, 240
    plans, nslots = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    ctx = Ctx(stub, W, H, nslots)
    d = synth.picture_for_plan(plans[2], W, H, seed=509, tool_flags=TOOLS | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, p_intra=0.3, p_cclm=0.3, p_ciip=0.15, p_isp=0.2)
    hnd = ctx.prepare(d)
    units, items = ctx.tables(hnd)
    assert _check_tables(d, units, items) > 0
    indep = items["comp"] >> 2
    assert (indep > 0).sum() > 10, "no independent blocks found in a B picture with isolated intra CUs"
    caught = 0
    serial = [i for i in range(1, len(items)) if indep[i] == 0 and items["mode"][i] < 254]
    for i in serial[:60]:
        b = items.copy()
        b["comp"][i] = (int(b["comp"][i]) & 3) | (1 << 2)
        try:
            _check_tables(d, units, b)
        except AssertionError:
            caught += 1
    assert caught > 0
    stub.vvr_free_prepared(ctx.ctx, hnd)
    ctx.close()


def test_external_slot_users_are_ordered_on_the_device(stub):
    """vvr_stream_wait_job / vvr_stream_wait_slot / vvr_slot_external_event (the collective that replicates reference pictures between GPUs): the
    caller's stream waits for the picture's completion EVENT, a later picture that reads a slot an external producer wrote waits for the caller's
    event on its lane - stream and event operations only, recorded by the stand-in runtime; the non-blocking forms say VVR_NOT_READY while a
    picture is still with the worker threads"""
    W, H = 256, 128
    plans, nslots = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    cfg = abi.Config()
    cfg.abi_version = abi.VVR_ABI_VERSION
    cfg.device, cfg.max_width, cfg.max_height = 0, W, H
    cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = 1, 10, 7
    cfg.num_slots, cfg.num_streams, cfg.host_threads = nslots, 3, 2
    ctx = C.c_void_p()
    assert stub.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_OK
    stub.vvr_stream_wait_job.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    stub.vvr_stream_wait_slot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    stub.vvr_slot_external_event.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    stub.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
    ext = C.c_void_p()
    stub.hipStreamCreateWithFlags(C.byref(ext), 0)
    descs = [synth.picture_for_plan(pl, W, H, seed=611, tool_flags=TOOLS) for pl in plans[:3]]
    pics = [d.c() for d in descs]
    stub.vvt_set_delay(20000)                               # the workers take a while: the picture is not with the device yet
    j0 = stub.vvr_submit(ctx, C.byref(pics[0]))
    assert j0 >= 0
    assert stub.vvr_stream_wait_job(ctx, j0, ext, 0) in (abi.VVR_NOT_READY, abi.VVR_OK)
    stub.vvt_set_delay(0)
    scratch = (C.c_int * 30000)()
    stub.vvt_take_trace(scratch, len(scratch))              # (start the record of stream / event operations here)
    assert stub.vvr_stream_wait_job(ctx, j0, ext, 1) == abi.VVR_OK       # waits (host) for the hand-over to the device only
    # an external producer writes the slot of the next key picture (as a receive from another GPU would): nobody uses it yet
    slot = plans[1].slot
    assert stub.vvr_stream_wait_slot(ctx, slot, ext, 1) == abi.VVR_OK
    ev = C.c_void_p()
    stub.hipEventCreate(C.byref(ev))
    stub.hipEventRecord(ev, ext)
    stub.vvt_events_pending(1)                              # (the receive is still running: an external event that is complete would be dropped instead of waited for)
    assert stub.vvr_slot_external_event(ctx, slot, ev, 1) == abi.VVR_OK
    # picture 2 predicts from that slot: its lane waits for the external event before anything of it runs
    assert plans[2].ref_slots and slot in [s for lst in plans[2].ref_slots for (s, _) in lst]
    j2 = stub.vvr_submit(ctx, C.byref(pics[2]))
    assert j2 >= 0 and stub.vvr_stream_wait_job(ctx, j2, ext, 1) == abi.VVR_OK      # (handed to the device)
    stub.vvt_events_pending(0)
    assert stub.vvr_sync(ctx) == abi.VVR_OK
    # the event is complete and the back-end has been through vvr_sync: it holds the handle no longer, the caller may destroy it
    dead0 = stub.vvt_dead_event_uses()
    stub.hipEventDestroy(ev)
    j1 = stub.vvr_submit(ctx, C.byref(pics[1]))             # writes the slot, then a picture that reads it again
    assert j1 >= 0 and stub.vvr_sync(ctx) == abi.VVR_OK
    assert stub.vvr_stream_wait_slot(ctx, slot, ext, 1) == abi.VVR_OK
    assert stub.vvt_dead_event_uses() == dead0, "the back-end used an external event after it was complete and vvr_sync had returned"
    n = stub.vvt_take_trace(scratch, len(scratch))
    ops = [(scratch[3 * k], scratch[3 * k + 1], scratch[3 * k + 2]) for k in range(n // 3)]
    waits = [(s, e) for (op, s, e) in ops if op == 0]
    records = [(s, e) for (op, s, e) in ops if op == 1]
    ext_stream = max([s for (s, _) in records] + [s for (s, _) in waits])   # (the stream created last: ours)
    ext_events = [e for (s, e) in records if s == ext_stream]
    assert ext_events, "the external event was not recorded on the external stream"
    assert any(s == ext_stream for (s, e) in waits), "the external stream never waited for a picture's event"
    assert any(e == ext_events[-1] and s != ext_stream for (s, e) in waits), "no lane waited for the external producer's event"
    stub.vvr_destroy(ctx)


def test_an_external_event_is_waited_for_whatever_a_query_says(stub):
    """hipEventQuery cannot tell an event that is complete from one that has not been recorded yet: between two vvr_sync calls the back-end therefore
    waits for every registered event of a slot it uses and forgets none (round-4 advice: an event registered a moment before its record was dropped
    at the next hand-over and the picture read the slot under the collective).  The stand-in says "complete" for every event here"""
    W, H = 256, 128
    plans, nslots = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    ctx = Ctx(stub, W, H, nslots, streams=3)
    stub.vvr_stream_wait_slot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    stub.vvr_slot_external_event.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    stub.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
    ext = C.c_void_p()
    stub.hipStreamCreateWithFlags(C.byref(ext), 0)
    descs = [synth.picture_for_plan(pl, W, H, seed=612, tool_flags=TOOLS) for pl in plans[:3]]
    pics = [d.c() for d in descs]
    assert stub.vvr_submit(ctx.ctx, C.byref(pics[0])) >= 0 and stub.vvr_sync(ctx.ctx) == abi.VVR_OK
    slot = plans[1].slot
    assert slot in [s for lst in plans[2].ref_slots for (s, _) in lst]
    scratch = (C.c_int * 30000)()
    stub.vvt_take_trace(scratch, len(scratch))
    stub.vvt_events_pending(0)                              # hipEventQuery: hipSuccess, as for an event nobody has recorded yet
    assert stub.vvr_stream_wait_slot(ctx.ctx, slot, ext, 1) == abi.VVR_OK
    ev = C.c_void_p()
    stub.hipEventCreate(C.byref(ev))
    assert stub.vvr_slot_external_event(ctx.ctx, slot, ev, 1) == abi.VVR_OK
    stub.hipEventRecord(ev, ext)                            # (recorded AFTER the registration)
    assert stub.vvr_submit(ctx.ctx, C.byref(pics[2])) >= 0
    n_before_sync = None
    assert stub.vvr_sync(ctx.ctx) == abi.VVR_OK
    n = stub.vvt_take_trace(scratch, len(scratch))
    ops = [(scratch[3 * k], scratch[3 * k + 1], scratch[3 * k + 2]) for k in range(n // 3)]
    records = [(s_, e) for (op, s_, e) in ops if op == 1]
    waits = [(s_, e) for (op, s_, e) in ops if op == 0]
    ext_stream = max(s_ for (s_, _) in records + waits)
    mine = [e for (s_, e) in records if s_ == ext_stream]
    assert mine and any(e == mine[-1] and s_ != ext_stream for (s_, e) in waits), "the picture that reads the slot did not wait for the external event"
    ctx.close()


def test_job_status_without_waiting(stub):
    """vvr_test: the ready check of a completion task (no thread sleeps in vvr_wait) - not ready while the device works, VVR_OK afterwards, and the job
    can still be waited for"""
    W, H = 256, 128
    plans, nslots = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    cfg = abi.Config()
    cfg.abi_version = abi.VVR_ABI_VERSION
    cfg.device, cfg.max_width, cfg.max_height = 0, W, H
    cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = 1, 10, 7
    cfg.num_slots, cfg.num_streams, cfg.host_threads = nslots, 2, 2
    ctx = C.c_void_p()
    assert stub.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_OK
    stub.vvr_test.argtypes = [C.c_void_p, C.c_int]
    stub.vvr_wait.argtypes = [C.c_void_p, C.c_int]
    stub.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
    stub.vvr_stream_wait_job.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    d = synth.picture_for_plan(plans[0], W, H, seed=612, tool_flags=TOOLS)
    p = d.c()
    stub.vvt_events_pending(1)                              # the device has not finished anything
    j = stub.vvr_submit(ctx, C.byref(p))
    assert j >= 0
    assert stub.vvr_test(ctx, j) == abi.VVR_NOT_READY       # with the workers, or on the device
    ext = C.c_void_p()
    stub.hipStreamCreateWithFlags(C.byref(ext), 0)
    assert stub.vvr_stream_wait_job(ctx, j, ext, 1) == abi.VVR_OK      # handed to the device
    assert stub.vvr_test(ctx, j) == abi.VVR_NOT_READY
    stub.vvt_events_pending(0)
    assert stub.vvr_test(ctx, j) == abi.VVR_OK and stub.vvr_test(ctx, j) == abi.VVR_OK
    assert stub.vvr_wait(ctx, j) == abi.VVR_OK
    # a picture that fails while its work lists are built: the status comes back from vvr_test as it does from vvr_wait
    bad = synth.picture_for_plan(plans[0], W, H, seed=613, tool_flags=TOOLS)
    bad.cu["w"][3] = 3
    pb = bad.c()
    jb = stub.vvr_submit(ctx, C.byref(pb))
    assert jb >= 0
    import time as _t
    rc = abi.VVR_NOT_READY
    for _ in range(2000):
        rc = stub.vvr_test(ctx, jb)
        if rc != abi.VVR_NOT_READY:
            break
        _t.sleep(0.005)
    assert rc < 0 and stub.vvr_wait(ctx, jb) == rc
    stub.vvr_destroy(ctx)


def _expect_error(ctx, d, code, text):
    p = d.c()
    h = C.c_void_p()
    rc = ctx.L.vvr_prepare(ctx.ctx, C.byref(p), C.byref(h))
    assert rc == code, "expected %d, got %d (%s)" % (code, rc, ctx.L.vvr_last_error(ctx.ctx).decode())
    assert text in ctx.L.vvr_last_error(ctx.ctx).decode()


def test_malformed_descriptions_are_rejected(stub):
    """validate() / the work-list builder refuse descriptions the kernels could not run safely (error code + message, nothing queued)"""
    W, H = 256, 128
    plans, nslots = stream.ra_plan(3, gop=2, seed_poc0_is_external=False)
    ctx = Ctx(stub, W, H, nslots)
    mk = lambda pl, tools=TOOLS, **kw: synth.picture_for_plan(pl, W, H, seed=509, tool_flags=tools, **kw)
    # geometry / header
    d = mk(plans[0]); d.hdr.abi_version += 1
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "abi_version")
    d = mk(plans[0]); d.hdr.out_slot = nslots
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "out_slot")
    d = mk(plans[1]); d.hdr.ref_slot[0][0] = d.hdr.out_slot
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "reference slot")
    d = mk(plans[0], tools=TOOLS | abi.TOOL_LMCS_CSCALE)
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "LMCS")
    # coding units
    d = mk(plans[0]); d.cu["x"][0] = W - 4
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "CU outside")
    d = mk(plans[1], p_intra=0.0); d.cu["ref_idx"][0] = (5, 5)
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "ref_idx")
    d = mk(plans[1], p_intra=0.0); k = int(np.nonzero(d.cu["ref_idx"][:, 0] >= 0)[0][0]); d.cu["mv"][k][0][0] = (1 << 17, 0)
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "18-bit range")
    d = mk(plans[1], p_intra=0.0); d.cu["mc_mode"][0] = 77
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "mc_mode")
    d = mk(plans[1], p_intra=0.0); d.cu["w"][0] = 4; d.cu["h"][0] = 4
    t0 = int(d.cu["first_tu"][0]); d.tu["w"][t0:t0 + int(d.cu["num_tu"][0])] = 4; d.tu["h"][t0:t0 + int(d.cu["num_tu"][0])] = 4
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "4x4 inter CU")
    d = mk(plans[1], p_intra=0.0); d.cu["w"][0] //= 2             # a hole in the picture (and a TU that sticks out of its CU)
    t0 = int(d.cu["first_tu"][0]); d.tu["w"][t0:t0 + int(d.cu["num_tu"][0])] //= 2
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "do not cover the picture")
    # transform units: owned by their CU, inside it, coded corner inside the block and the level stream
    d = mk(plans[0]); d.tu["cu"][0] = 1
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "does not name its CU")
    d = mk(plans[0]); d.tu["x"][0] += 4
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "TU outside its CU")
    d = mk(plans[0], p_coded=1.0, p_ts=0.0, p_bdpcm=0.0)
    k = int(np.nonzero((d.tu["cbf"] & 1) != 0)[0][0])
    d.tu["max_scan_x"][k][0] = 200
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "last significant position")
    d = mk(plans[0], p_coded=1.0, p_ts=0.0, p_bdpcm=0.0); d.tu["coef_off"][k][0] = len(d.coef)
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "outside the level stream")
    d = mk(plans[0]); d.cu["intra_dir"][0] = (90, 0)
    _expect_error(ctx, d, abi.VVR_ERR_UNSUPPORTED, "intra mode")
    # intra block copy
    d = mk(plans[0], tools=TOOLS | abi.TOOL_IBC, p_ibc=0.6)
    k = int(np.nonzero(d.cu["pred_mode"] == abi.PRED_IBC)[0][0])
    d.hdr.tool_flags &= ~abi.TOOL_IBC
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "VVR_TOOL_IBC")
    d = mk(plans[0], tools=TOOLS | abi.TOOL_IBC, p_ibc=0.6); d.cu["mv"][k][0][0][0] += 3
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "fractional block vector")
    d = mk(plans[0], tools=TOOLS | abi.TOOL_IBC, p_ibc=0.6); d.cu["mv"][k][0][0] = (-16 * 4096, 0)
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "reference block outside")
    d = mk(plans[0], tools=TOOLS | abi.TOOL_IBC, p_ibc=0.6); d.cu["mv"][k][0][0] = (0, 0)       # "copies" itself: not reconstructed yet
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "not reconstructed before")
    # a well-formed description still goes through afterwards
    hnd = ctx.prepare(mk(plans[0], tools=TOOLS | abi.TOOL_IBC, p_ibc=0.6))
    stub.vvr_free_prepared(ctx.ctx, hnd)
    ctx.close()


def test_slice_headers_in_the_host_glue(stub):
    """slices with headers of their own: the chroma blocks' residual-scaling flag follows the slice of the CTU, the intra tables stay valid, and
    headers that do not fit the picture are refused"""
    W, H, l2 = 512, 384, 6
    plans, nslots = stream.ra_plan(3, gop=2, seed_poc0_is_external=False)
    ctx = Ctx(stub, W, H, nslots, log2_ctu=l2)
    T = TOOLS | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_SCALING_LIST | abi.TOOL_WP

    def mk(pl):
        d = synth.picture_for_plan(pl, W, H, seed=520, tool_flags=T, log2_ctu=l2, num_slices=3, p_intra=0.4, p_coded=0.9, p_coded_chroma=0.8, p_cclm=0.3)
        return synth.vary_slices(d, 521 + pl.poc)
    for pl in plans:
        d = mk(pl)
        hnd = ctx.prepare(d)
        units, items = ctx.tables(hnd)
        assert _check_tables(d, units, items) > 0
        assert _check_resi_add(d, units, *ctx.resi_tables(hnd)) > 0 or pl.slice_type == abi.SLICE_I
        ch = items[((items["comp"] & 3) != 0) & (items["mode"] != 254)]                         # chroma blocks (not IBC): bit 8 = IT_F_CSCALE
        ctus_x = (W + (1 << l2) - 1) >> l2
        sl = d.ctu_slice[(ch["y"].astype(int) >> (l2 - 1)) * ctus_x + (ch["x"].astype(int) >> (l2 - 1))]
        on = (d.slices["tool_flags"][sl] & abi.TOOL_LMCS_CSCALE) != 0
        assert on.any() and (~on).any()
        assert not (ch["flags"][~on] & 8).any(), "chroma residual scaling in a slice without it"
        assert (ch["flags"][on] & 8).any()
        stub.vvr_free_prepared(ctx.ctx, hnd)
    d = mk(plans[0]); d.slices = d.slices[:2]
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "without a header")
    d = mk(plans[0]); d.slices["alf_set"][1] = 2
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "table that is not there")
    d = mk(plans[1]); d.slices["wp_set"][0] = 2
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "table that is not there")
    d = mk(plans[0]); d.slices["tool_flags"][1] |= abi.TOOL_LMCS_CSCALE
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "LMCS chroma residual scaling without LMCS")
    d = mk(plans[0]); d.ctu_slice = None
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "slice map")
    d = mk(plans[0]); d.slices["slice_type"][0] = 3
    _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "slice type")
    ctx.close()


def test_context_configuration_limits(stub):
    """vvr_create: Main 10 sample formats and CTU sizes only, ABI version checked"""
    def create(**kw):
        cfg = abi.Config()
        cfg.abi_version = abi.VVR_ABI_VERSION
        cfg.device, cfg.max_width, cfg.max_height, cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu, cfg.num_slots, cfg.num_streams = 0, 256, 128, 1, 10, 7, 4, 1
        for k, v in kw.items():
            setattr(cfg, k, v)
        ctx = C.c_void_p()
        rc = stub.vvr_create(C.byref(cfg), C.byref(ctx))
        if rc == abi.VVR_OK:
            stub.vvr_destroy(ctx)
        return rc
    assert create() == abi.VVR_OK
    assert create(bit_depth=8) == abi.VVR_OK and create(chroma_format=0) == abi.VVR_OK and create(log2_ctu=5) == abi.VVR_OK
    assert create(bit_depth=12) == abi.VVR_ERR_UNSUPPORTED and create(bit_depth=7) == abi.VVR_ERR_UNSUPPORTED
    assert create(chroma_format=2) == abi.VVR_ERR_UNSUPPORTED and create(log2_ctu=4) == abi.VVR_ERR_UNSUPPORTED
    assert create(abi_version=abi.VVR_ABI_VERSION + 1) == abi.VVR_ERR_PARAMETER
    assert create(device=3) == abi.VVR_ERR_NO_DEVICE          # the stand-in runtime reports one device


@pytest.mark.parametrize("lanes,frames,gop,pool", [(3, 17, 8, 0), (4, 33, 16, 12), (2, 9, 4, 0)])
def test_pictures_in_flight_are_ordered_by_their_slots(stub, lanes, frames, gop, pool):
    """several pictures in flight on different streams: a picture must be ordered (stream order or event waits, transitively) after the
    writer of every reference slot it reads and after every earlier user of the slot it writes (DPB slots are reused)"""
    W, H = 64, 64
    plans, nslots = stream.ra_plan(frames, gop=gop, seed_poc0_is_external=False, pool=pool)
    ctx = Ctx(stub, W, H, nslots, log2_ctu=5, streams=lanes)
    buf = (C.c_int * 30000)()
    stub.vvt_take_trace(buf, len(buf))
    jobs = []                                   # per picture: (lane, set of event numbers waited for, event number of the picture)
    for pl in plans:
        d = synth.picture_for_plan(pl, W, H, seed=510, tool_flags=TOOLS, log2_ctu=5, p_intra=0.1)
        hnd = ctx.prepare(d)
        assert stub.vvr_submit_prepared(ctx.ctx, hnd) >= 0
        n = stub.vvt_take_trace(buf, len(buf))
        ops = [tuple(buf[i:i + 3]) for i in range(0, n, 3)]
        rec = [o for o in ops if o[0] == 1]
        assert len(rec) == 2                    # the picture's completion events (one for later pictures' streams, one for host threads), on its lane
        jobs.append((rec[0][1], {o[2] for o in ops if o[0] == 0}, rec[0][2], hnd))
        assert all(o[1] == rec[0][1] for o in ops)
    # all lanes are used; with more than one, the I picture at the head of the stream goes to the lane with the high-priority stream
    assert len({j[0] for j in jobs}) == lanes + (1 if lanes >= 2 else 0)
    if lanes >= 2:
        assert all(jobs[0][0] != j[0] for j in jobs[1:])
    ev_to_job = {j[2]: k for k, j in enumerate(jobs)}

    def violations(use_waits):
        before, last_on_lane, users, bad = [], {}, {}, []
        for k, (lane, waits, _, _) in enumerate(jobs):
            hb = set()
            if lane in last_on_lane:
                hb |= before[last_on_lane[lane]] | {last_on_lane[lane]}
            for e in (waits if use_waits else ()):
                hb |= before[ev_to_job[e]] | {ev_to_job[e]}
            before.append(hb)
            last_on_lane[lane] = k
        for k, pl in enumerate(plans):           # users[slot] = [writer, readers...] since the last write
            reads = [slot for lst in pl.ref_slots for (slot, poc) in lst]
            for slot in reads:
                assert slot in users, "reference slot never written"
                if users[slot][0] not in before[k]:
                    bad.append("picture %d reads slot %d without being ordered after its writer" % (k, slot))
            for u in users.get(pl.slot, []):
                if u not in before[k]:
                    bad.append("picture %d overwrites slot %d while picture %d may still use it" % (k, pl.slot, u))
            for slot in reads:
                users[slot].append(k)
            users[pl.slot] = [k]
        return bad

    assert violations(True) == []
    assert violations(False) != []              # (the check is not vacuous: stream order alone does not cover the hazards)
    stub.vvr_sync(ctx.ctx)
    for j in jobs:
        stub.vvr_free_prepared(ctx.ctx, j[3])
    ctx.close()


def test_output_window_and_8bit_frames(stub):
    """vvr_read_output: conformance-window crop and the narrowing to 8-bit frames (VVDecImpl::copyComp), on planes written through the ABI"""
    W, H = 64, 48
    stub.vvr_read_output.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_size_t]
    stub.vvr_write_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(3)
    for bd in (8, 10):
        ctx = Ctx(stub, W, H, 2, log2_ctu=5, bit_depth=bd)
        planes = [rng.integers(0, 1 << bd, (H >> s, W >> s)).astype(np.uint16) for s in (0, 1, 1)]
        for c, pl in enumerate(planes):
            assert stub.vvr_write_plane(ctx.ctx, 1, c, pl.ctypes.data, pl.shape[1]) == abi.VVR_OK
        x, y, w, h = 8, 4, 40, 36
        for c, pl in enumerate(planes):
            s = 1 if c else 0
            out = np.zeros((h >> s, (w >> s) + 5), np.uint16)                       # row pitch larger than the window
            assert stub.vvr_read_output(ctx.ctx, 1, c, x >> s, y >> s, w >> s, h >> s, 2, out.ctypes.data, out.strides[0]) == abi.VVR_OK
            assert np.array_equal(out[:, :w >> s], pl[y >> s:(y + h) >> s, x >> s:(x + w) >> s]) and not out[:, w >> s:].any()
            out8 = np.zeros((h >> s, w >> s), np.uint8)
            rc = stub.vvr_read_output(ctx.ctx, 1, c, x >> s, y >> s, w >> s, h >> s, 1, out8.ctypes.data, out8.strides[0])
            if bd == 8:
                assert rc == abi.VVR_OK and np.array_equal(out8, pl[y >> s:(y + h) >> s, x >> s:(x + w) >> s].astype(np.uint8))
            else:
                assert rc == abi.VVR_ERR_PARAMETER                                   # only 8-bit content is narrowed
        assert stub.vvr_read_output(ctx.ctx, 1, 0, 32, 0, 40, 8, 2, out.ctypes.data, 256) == abi.VVR_ERR_PARAMETER      # window leaves the plane
        ctx.close()


@pytest.mark.parametrize("bd,cf", [(10, 1), (8, 1), (10, 0)])
def test_decoded_picture_hash(stub, bd, cf):
    """vvr_picture_hash (MD5 / CRC / checksum of the decoded-picture-hash SEI) against the reference's own functions (PicYuvMD5.cpp) and,
    for MD5, Python's hashlib, on planes written through the ABI"""
    import hashlib
    import refdrv
    W, H = 136, 72                                          # not a multiple of the 32-sample blocks the reference feeds MD5 with
    stub.vvr_write_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    stub.vvr_picture_hash.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(bd * 10 + cf)
    ctx = Ctx(stub, W, H, 2, log2_ctu=5, bit_depth=bd, chroma_format=cf)
    planes = [rng.integers(0, 1 << bd, (H >> s, W >> s)).astype(np.uint16) for s in ((0, 1, 1) if cf else (0,))]
    for c, pl in enumerate(planes):
        assert stub.vvr_write_plane(ctx.ctx, 1, c, pl.ctypes.data, pl.shape[1]) == abi.VVR_OK
    for method, length in ((0, 16), (1, 2), (2, 4)):
        buf = (C.c_uint8 * 48)()
        n = C.c_int()
        assert stub.vvr_picture_hash(ctx.ctx, 1, method, buf, C.byref(n)) == abi.VVR_OK and n.value == length
        got = bytes(buf[:length * len(planes)])
        if method == 0:
            want = b"".join(hashlib.md5(pl.astype(np.uint8 if bd == 8 else "<u2").tobytes()).digest() for pl in planes)
            assert got == want
        if refdrv.available():
            L = refdrv.lib()
            ptrs = (C.POINTER(C.c_uint16) * 3)()
            for c, pl in enumerate(planes):
                ptrs[c] = pl.ctypes.data_as(C.POINTER(C.c_uint16))
            ref = (C.c_uint8 * 48)()
            assert L.vvref_picture_hash(ptrs, W, H, cf, bd, method, ref) == length
            assert got == bytes(ref[:length * len(planes)]), "method %d differs from the reference" % method
    ctx.close()


def test_golden_fixtures_pass_the_host_glue(stub):
    """every committed fixture (descriptions written by earlier versions of the generator) is accepted by validate() and gives sound tables"""
    import glob
    import golden_io
    n = 0
    for path in sorted(glob.glob(os.path.join(HERE, "golden", "*.npz"))):
        d, refs, _ = golden_io.load(path)
        h = d.hdr
        MW = max([h.width] + [r[0].shape[1] for r in refs.values()]); MH = max([h.height] + [r[0].shape[0] for r in refs.values()])      # (scaled reference pictures have their own sizes)
        ctx = Ctx(stub, MW, MH, max([h.out_slot] + list(refs.keys())) + 1, log2_ctu=h.log2_ctu, bit_depth=h.bit_depth, chroma_format=h.chroma_format)
        hnd = ctx.prepare(d)
        units, items = ctx.tables(hnd)
        _check_tables(d, units, items)
        stub.vvr_free_prepared(ctx.ctx, hnd)
        ctx.close()
        n += 1
    assert n >= 39


@pytest.mark.parametrize("ranks", [1, 2])
def test_bench_control_flow(stub, ranks):
    """bench.py against the stand-in runtime, one process and two ranks over gloo (the N > 1 launch of the driver): every rank gets through
    the barriers, rank 0 prints exactly one JSON line with the contract's keys"""
    import json
    import sys
    tool = os.path.join(os.path.dirname(HERE), "tools", "bench_host_side.py")
    env = dict(os.environ, VVR_BENCH_BACKEND="gloo")
    cmd = [sys.executable, tool] if ranks == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                                                    "--master-addr", "127.0.0.1", "--master-port", "29533", tool]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out
    assert out["n_gpus"] == ranks and out["steps"] == 6 and out["warmup"] == 4 and "workload" in out["config"]
    if ranks == 1:
        assert out["scaling"] is None and out["config"]["sharding"] is None          # (one GPU: nothing is sharded)
    else:
        # N > 1 (round 6): `value` IS the picture mode - ONE stream sharded by picture over the ranks, a step = one picture per GPU (steps x ranks pictures in the
        # window, weak scaling) - and the closed-GOP segment mode stands beside it
        assert out["scaling"] == "weak" and "ONE stream sharded by picture" in out["config"]["sharding"] and "picture mode" in out["config"]["value_is"]
        pm, sm = out["value_picture_mode"], out["value_segment_mode"]
        assert out["value"] == pm["fps"] > 0 and pm["pictures"] == out["steps"] * ranks and pm["ranks"] == ranks and sm["fps"] > 0 and sm["pictures"] == out["steps"] * ranks
        assert abs(out["ms_per_step"] - 1e3 * out["steps"] * ranks / out["value"] / out["steps"]) < 0.05 * out["ms_per_step"]
        c = pm["ceiling"]
        assert 1.0 <= c["speedup_over_one_gpu_at_least"]["open_stream"] <= c["speedup_over_one_gpu_at_most"]["open_stream"] <= ranks
        assert "rccl_ranks" in out
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"]
    if ranks > 1:
        ps = out["config"]["picture_sharding"]           # the pass itself: ONE stream, pictures sharded over the ranks, reference slots to their dependants
        assert "error" not in ps and ps["fps"] > 0 and ps["scaling"] == "weak" and ps["point_to_point_sends_in_window"] > 0, ps


def test_bench_line_survives_the_picture_sharding_pass(stub):
    """the picture-sharding pass of bench.py runs last and under a watchdog: if it does not come back (here: a deadline of zero seconds), rank 0 still
    prints the line with the segment-mode result, and every rank exits"""
    import json
    import sys
    tool = os.path.join(os.path.dirname(HERE), "tools", "bench_host_side.py")
    env = dict(os.environ, VVR_BENCH_BACKEND="gloo", VVR_BENCH_EXTRA="--picture-sharding-timeout 0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29535", tool]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.returncode, r.stderr[-1500:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "error" in out["config"]["picture_sharding"] or "fps" in out["config"]["picture_sharding"]      # (it may have finished before the deadline fired)
    # a pass that was given up is not a success: the line says so and the launch does not end with 0
    if out["config"]["picture_sharding"].get("timeout"):
        assert out.get("timeout") is True and r.returncode != 0


def _submit_stream(stub, threads, W=416, H=240, frames=17, gop=8, lanes=3):
    """a whole stream through vvr_submit on a context with `threads` worker threads -> the (op, stream) sequence of the stream / event operations"""
    plans, nslots = stream.ra_plan(frames, gop=gop, seed_poc0_is_external=False, pool=12)
    cfg = abi.Config()
    cfg.abi_version = abi.VVR_ABI_VERSION
    cfg.device, cfg.max_width, cfg.max_height, cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = 0, W, H, 1, 10, 5
    cfg.num_slots, cfg.num_streams, cfg.host_threads = max(nslots, 12), lanes, threads
    ctx = C.c_void_p()
    assert stub.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_OK
    stub.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
    stub.vvr_wait.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_int * 60000)()
    stub.vvt_take_trace(buf, len(buf))
    descs = [synth.picture_for_plan(pl, W, H, seed=511, tool_flags=TOOLS | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, log2_ctu=5, p_intra=0.15, p_affine=0.1) for pl in plans]
    pics = [d.c() for d in descs]
    jobs = [stub.vvr_submit(ctx, C.byref(p)) for p in pics]
    assert all(j >= 0 for j in jobs) and jobs == sorted(jobs)
    for j in jobs:
        assert stub.vvr_wait(ctx, j) == abi.VVR_OK
    n = stub.vvt_take_trace(buf, len(buf))
    ops = [(buf[i], buf[i + 1]) for i in range(0, n, 3)]
    base = min(o[1] for o in ops)                       # (the stand-in runtime numbers streams across contexts)
    stub.vvr_destroy(ctx)
    # the event records: per picture one on the copy stream (its upload) and two on its lane (its completion: for streams, for host threads).  Which waits are issued also
    # depends on which earlier pictures the host already knows to be finished (their events are not waited for again), i.e. on the ring size
    return [(op, st - base) for op, st in ops if op == 1]


def test_worker_threads_commit_in_submission_order(stub):
    """pictures prepared concurrently by worker threads are enqueued on the device exactly like pictures prepared by the submitting thread:
    same lanes, same copies, uploads and completions in the same order"""
    inline = _submit_stream(stub, 0)
    assert len(inline) == 17 * 3
    for threads in (1, 3):
        assert _submit_stream(stub, threads) == inline


def test_i_pictures_are_prepared_by_the_workers_together(stub):
    """the work lists of a picture are built in parts (bands of CTU rows) by the worker threads together and appended in band order - an I picture always, a
    picture with inter CUs while the device is short of work: everything that is uploaded for a stream - I pictures of several sizes among B pictures - is
    byte for byte what one thread uploads"""
    stub.vvt_take_h2d_hash.restype = C.c_ulonglong
    stub.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
    stub.vvr_wait.argtypes = [C.c_void_p, C.c_int]

    def run(threads, W, H, l2, frames, gop, intra_period, tools, **kw):
        plans, nslots = stream.ra_plan(frames, gop=gop, seed_poc0_is_external=False, pool=12, intra_period=intra_period)
        cfg = abi.Config()
        cfg.abi_version = abi.VVR_ABI_VERSION
        cfg.device, cfg.max_width, cfg.max_height, cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = 0, W, H, 1, 10, l2
        cfg.num_slots, cfg.num_streams, cfg.host_threads = max(nslots, 12), 2, threads
        ctx = C.c_void_p()
        assert stub.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_OK
        stub.vvt_take_h2d_hash()
        descs = [synth.picture_for_plan(pl, W, H, seed=530, tool_flags=tools, log2_ctu=l2, **kw) for pl in plans]
        assert sum(1 for pl in plans if pl.slice_type == abi.SLICE_I) >= 2
        pics = [d.c() for d in descs]
        hashes = []
        for p in pics:                           # one picture at a time: the copies of a picture are issued in a fixed order
            j = stub.vvr_submit(ctx, C.byref(p))
            assert j >= 0 and stub.vvr_wait(ctx, j) == abi.VVR_OK, stub.vvr_last_error(ctx).decode()
            hashes.append(stub.vvt_take_h2d_hash())
        stub.vvr_destroy(ctx)
        return hashes
    # (the last case: edge parameters left to the back-end - the list of sub-block motion the bands build for k_lf_maps is appended in band order too)
    for (W, H, l2, kw) in ((1920, 1080, 7, dict(p_cclm=0.3, p_isp=0.2, p_mip=0.2)), (832, 480, 6, dict(dual_tree=1.0, p_cclm=0.3)), (416, 240, 5, dict()),
                           (1280, 720, 7, dict(extra=abi.TOOL_LFP_ON_DEVICE, p_intra=0.1, p_sbtmvp=0.2, p_geo=0.15, p_affine=0.2))):
        kw = dict(kw)
        T = TOOLS | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | kw.pop("extra", 0)
        one = run(0, W, H, l2, 9, 4, 4, T, **kw)
        assert len(set(one)) == len(one)
        for threads in (2, 3, 8):
            before = stub.vvt_band_pictures()
            assert run(threads, W, H, l2, 9, 4, 4, T, **kw) == one, (W, H, threads)
            # (submitted one at a time the device is always short of work: the B pictures go the same way, what a band reads of the band above looked up when
            # the bands are joined)
            assert stub.vvt_band_pictures() - before >= (5 if W >= 832 else 0), (W, H, threads)        # (the smallest size has too few CUs to be split)
    # records that are wrong in a part other than the first are reported like by one thread
    plans, nslots = stream.ra_plan(1, gop=1, seed_poc0_is_external=False)
    d = synth.picture_for_plan(plans[0], 1920, 1080, seed=531, tool_flags=TOOLS)
    d.cu["w"][len(d.cu) - 3] = 0
    cfg = abi.Config()
    cfg.abi_version = abi.VVR_ABI_VERSION
    cfg.device, cfg.max_width, cfg.max_height, cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = 0, 1920, 1080, 1, 10, 7
    cfg.num_slots, cfg.num_streams, cfg.host_threads = 4, 2, 4
    ctx = C.c_void_p()
    assert stub.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_OK
    p = d.c()
    j = stub.vvr_submit(ctx, C.byref(p))
    assert j >= 0 and stub.vvr_wait(ctx, j) == abi.VVR_ERR_PARAMETER and "CU outside" in stub.vvr_last_error(ctx).decode()
    stub.vvr_destroy(ctx)


def test_errors_of_queued_pictures_come_back_from_wait(stub):
    """with worker threads, what only shows while the work lists are built is parked on the job (the reference parks exceptions on reconDone):
    vvr_submit has returned a job id, vvr_wait returns the error, later pictures are not held up"""
    W, H = 256, 128
    plans, nslots = stream.ra_plan(3, gop=2, seed_poc0_is_external=False)
    cfg = abi.Config()
    cfg.abi_version = abi.VVR_ABI_VERSION
    cfg.device, cfg.max_width, cfg.max_height, cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = 0, W, H, 1, 10, 7
    cfg.num_slots, cfg.num_streams, cfg.host_threads = nslots, 2, 2
    ctx = C.c_void_p()
    assert stub.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_OK
    stub.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
    stub.vvr_wait.argtypes = [C.c_void_p, C.c_int]
    mk = lambda pl: synth.picture_for_plan(pl, W, H, seed=509, tool_flags=TOOLS | abi.TOOL_IBC, p_ibc=0.6)
    bad = mk(plans[0])
    k = int(np.nonzero(bad.cu["pred_mode"] == abi.PRED_IBC)[0][0])
    bad.cu["mv"][k][0][0] = (0, 0)                       # "copies" itself: only the work-list builder sees that
    good = [mk(plans[0]), mk(plans[1])]
    pb, pg = bad.c(), [g.c() for g in good]
    jb = stub.vvr_submit(ctx, C.byref(pb))
    jg = [stub.vvr_submit(ctx, C.byref(p)) for p in pg]
    assert jb >= 0 and all(j >= 0 for j in jg)
    assert stub.vvr_wait(ctx, jg[1]) == abi.VVR_OK and stub.vvr_wait(ctx, jg[0]) == abi.VVR_OK
    assert stub.vvr_wait(ctx, jb) == abi.VVR_ERR_PARAMETER
    assert "not reconstructed before" in stub.vvr_last_error(ctx).decode()
    # what is wrong with the header is still refused by vvr_submit at once
    bad2 = mk(plans[0]); bad2.hdr.out_slot = nslots
    assert stub.vvr_submit(ctx, C.byref(bad2.c())) == abi.VVR_ERR_PARAMETER
    # the records are checked by the picture's worker (the submitting thread does not spend 0.3 ms per 4K picture on them): the job fails
    bad3 = mk(plans[0]); bad3.cu["w"][3] = 0
    pb3 = bad3.c()
    jb3 = stub.vvr_submit(ctx, C.byref(pb3))
    assert jb3 >= 0 and stub.vvr_wait(ctx, jb3) == abi.VVR_ERR_PARAMETER
    assert "CU outside the picture" in stub.vvr_last_error(ctx).decode()
    jg = stub.vvr_submit(ctx, C.byref(pg[0]))
    assert jg >= 0 and stub.vvr_wait(ctx, jg) == abi.VVR_OK
    stub.vvr_destroy(ctx)


def test_a_wait_of_the_intra_stage_that_gave_up_fails_its_own_picture(stub):
    """k_intra_leaf bounds every wait for a neighbouring block; a wavefront that gives up writes the error word of the picture's JOB (pinned host memory) and
    reconstructs from whatever is there.  That picture - and no other - fails with VVR_ERR_DEVICE wherever the decoder asks about it: vvr_wait, vvr_test, vvr_sync;
    the pictures around it are fine (round-5 advisor: only vvr_sync looked at a per-lane word, and the drop-in only calls vvr_wait)"""
    W, H = 256, 128
    plans, nslots = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    cfg = abi.Config()
    cfg.abi_version = abi.VVR_ABI_VERSION
    cfg.device, cfg.max_width, cfg.max_height, cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = 0, W, H, 1, 10, 7
    cfg.num_slots, cfg.num_streams, cfg.host_threads = nslots, 2, 2
    ctx = C.c_void_p()
    assert stub.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_OK
    stub.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
    stub.vvr_wait.argtypes = [C.c_void_p, C.c_int]
    stub.vvr_test.argtypes = [C.c_void_p, C.c_int]
    stub.vvr_sync.argtypes = [C.c_void_p]
    descs = [synth.picture_for_plan(pl, W, H, seed=611, tool_flags=TOOLS, p_intra=0.3) for pl in plans]        # (kept alive: the C structs point into their arrays)
    pics = [d.c() for d in descs]
    j0 = stub.vvr_submit(ctx, C.byref(pics[0]))                       # the I picture: the CTU-tile kernel, no such wait
    assert j0 >= 0 and stub.vvr_wait(ctx, j0) == abi.VVR_OK
    stub.vvt_fail_leaf_waits(1)
    j1 = stub.vvr_submit(ctx, C.byref(pics[1]))                       # its intra stage gives up a wait
    assert j1 >= 0 and stub.vvr_wait(ctx, j1) == abi.VVR_ERR_DEVICE
    assert "waited for its neighbours beyond the bound" in stub.vvr_last_error(ctx).decode()
    assert stub.vvr_test(ctx, j1) == abi.VVR_ERR_DEVICE               # (asked again: the same answer)
    j2 = stub.vvr_submit(ctx, C.byref(pics[2]))
    assert j2 >= 0 and stub.vvr_wait(ctx, j2) == abi.VVR_OK           # the next picture is not tainted
    stub.vvt_fail_leaf_waits(1)
    j3 = stub.vvr_submit(ctx, C.byref(pics[3]))
    j4 = stub.vvr_submit(ctx, C.byref(pics[4]))
    assert j3 >= 0 and j4 >= 0
    assert stub.vvr_sync(ctx) == abi.VVR_ERR_DEVICE                   # vvr_sync reports the failed job ...
    assert stub.vvr_wait(ctx, j4) == abi.VVR_OK                       # ... and the other one of the pair is fine
    stub.vvt_fail_leaf_waits(0)
    stub.vvr_destroy(ctx)


def test_records_in_pinned_memory_are_not_staged(stub):
    """arrays of a description that lie in memory of vvr_host_alloc go to the device from where they are: per picture one copy of the staged
    rest plus one per pinned array, the same bytes in total (up to alignment padding)"""
    W, H = 832, 480
    plans, nslots = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    cfg = abi.Config()
    cfg.abi_version = abi.VVR_ABI_VERSION
    cfg.device, cfg.max_width, cfg.max_height, cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = 0, W, H, 1, 10, 6
    cfg.num_slots, cfg.num_streams, cfg.host_threads = nslots, 2, 2
    ctx = C.c_void_p()
    assert stub.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_OK
    stub.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
    stub.vvr_wait.argtypes = [C.c_void_p, C.c_int]
    stub.vvr_inputs_done.argtypes = [C.c_void_p, C.c_int]
    stub.vvr_host_alloc.restype = C.c_void_p
    stub.vvr_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
    stub.vvt_take_h2d.argtypes = [C.c_void_p, C.c_void_p]

    def pinned(n, dt):
        dt = np.dtype(dt)
        nb = max(1, n) * dt.itemsize
        return np.frombuffer((C.c_char * nb).from_address(stub.vvr_host_alloc(ctx, nb)), dt, count=max(1, n))[:n]

    def run(alloc):
        cnt, nbytes = C.c_size_t(), C.c_size_t()
        stub.vvt_take_h2d(C.byref(cnt), C.byref(nbytes))
        out = []
        for pl in plans:
            d = synth.picture_for_plan(pl, W, H, seed=512, tool_flags=TOOLS, log2_ctu=6, p_intra=0.2, p_split_scale=1.5, p_coded=0.9, p_small_corner=0.1, alloc=alloc)
            p = d.c()
            j = stub.vvr_submit(ctx, C.byref(p))
            assert j >= 0 and stub.vvr_inputs_done(ctx, j) == abi.VVR_OK and stub.vvr_wait(ctx, j) == abi.VVR_OK
            stub.vvt_take_h2d(C.byref(cnt), C.byref(nbytes))
            out.append((cnt.value, nbytes.value, [a.nbytes >= 65536 for a in (d.lfp[0], d.lfp[1], d.cu, d.coef, d.tu)]))
        return out

    staged, direct = run(None), run(pinned)
    for (c0, b0, _), (c1, b1, big) in zip(staged, direct):
        ndirect = 0
        for flag in big:                       # (only a leading run of large arrays is copied directly)
            if not flag:
                break
            ndirect += 1
        assert c0 == 1 and c1 == 1 + ndirect and ndirect >= 2 and 0 <= b0 - b1 < 256 * ndirect      # (the staged image pads every array to 256 bytes)
    stub.vvr_destroy(ctx)


@pytest.mark.timeout(240)
def test_pipeline_under_load_does_not_stall(stub):
    """many pictures through 8 worker threads, the launcher and the upload ring, with device calls that take time (the stand-in sleeps where a real
    device would block): every picture completes.  (Found by a hang on the GPU: a job further back took the ring entry of the job whose turn it
    was, which then waited for an entry that could only be freed behind its own commit - entries are taken strictly by turn now.)"""
    W, H = 416, 240
    plans, nslots = stream.ra_plan(65, gop=16, seed_poc0_is_external=False, pool=24, intra_period=32, irap_lookahead=8)
    cfg = abi.Config()
    cfg.abi_version = abi.VVR_ABI_VERSION
    cfg.device, cfg.max_width, cfg.max_height, cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = 0, W, H, 1, 10, 6
    cfg.num_slots, cfg.num_streams, cfg.host_threads = max(nslots, 24), 8, 8
    ctx = C.c_void_p()
    assert stub.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_OK
    stub.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
    stub.vvr_wait.argtypes = [C.c_void_p, C.c_int]
    descs = [synth.picture_for_plan(pl, W, H, seed=513, tool_flags=TOOLS | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, log2_ctu=6, p_intra=0.15) for pl in plans]
    pics = [d.c() for d in descs]
    stub.vvt_set_delay(150)
    try:
        for rep in range(4):
            jobs = [stub.vvr_submit(ctx, C.byref(p)) for p in pics]
            assert all(j >= 0 for j in jobs)
            hs = []
            for p in pics[:20]:                                   # resident pictures queue up behind the streaming ones
                h = C.c_void_p()
                assert stub.vvr_prepare(ctx, C.byref(p), C.byref(h)) == abi.VVR_OK
                assert stub.vvr_submit_prepared(ctx, h) >= 0
                hs.append(h)
            assert stub.vvr_sync(ctx) == abi.VVR_OK
            for h in hs:
                stub.vvr_free_prepared(ctx, h)
    finally:
        stub.vvt_set_delay(0)
    stub.vvr_destroy(ctx)


def test_collocated_motion_host_stage(stub):
    """VVR_TOOL_COL_MOTION against the stand-in runtime (no DMVR kernel runs): what vvr_read_col_motion returns is the motion field at every second
    4x4 unit in both directions, for odd numbers of 4x4 columns / rows too, with worker threads and through prepared handles; asking without a
    motion field is refused"""
    stub.vvr_read_col_motion.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    stub.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
    stub.vvr_wait.argtypes = [C.c_void_p, C.c_int]
    for (W, H, threads) in ((200, 136, 0), (416, 240, 2)):
        plans, nslots = stream.ra_plan(3, gop=2, seed_poc0_is_external=False)
        cfg = abi.Config()
        cfg.abi_version = abi.VVR_ABI_VERSION
        cfg.device, cfg.max_width, cfg.max_height, cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = 0, W, H, 1, 10, 7
        cfg.num_slots, cfg.num_streams, cfg.host_threads = nslots, 2, threads
        ctx = C.c_void_p()
        assert stub.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_OK
        for k, pl in enumerate(plans):
            d = synth.picture_for_plan(pl, W, H, seed=521, tool_flags=TOOLS | abi.TOOL_COL_MOTION, p_intra=0.1, p_bi=0.8)
            p = d.c()
            if k == 1:
                h = C.c_void_p()
                assert stub.vvr_prepare(ctx, C.byref(p), C.byref(h)) == abi.VVR_OK
                job = stub.vvr_submit_prepared(ctx, h)
            else:
                job = stub.vvr_submit(ctx, C.byref(p))
            assert job >= 0
            n = stub.vvr_read_col_motion(ctx, job, None, 0)
            assert n == ((d.w4 + 1) // 2) * ((d.h4 + 1) // 2)
            got = np.zeros(n, np.dtype(abi.Motion))
            assert stub.vvr_read_col_motion(ctx, job, got.ctypes.data_as(C.c_void_p), n) == n
            want = d.motion.reshape(d.h4, d.w4)[::2, ::2].reshape(-1)
            assert got.tobytes() == want.tobytes()
            if k == 1:
                stub.vvr_free_prepared(ctx, h)
        d.motion = None
        p = d.c()
        job = stub.vvr_submit(ctx, C.byref(p))
        assert job == abi.VVR_ERR_PARAMETER or stub.vvr_wait(ctx, job) == abi.VVR_ERR_PARAMETER
        stub.vvr_destroy(ctx)


def test_pictures_pass_pictures_that_are_still_being_prepared(stub):
    """commit order: a picture may be enqueued ahead of a picture that was submitted before it and whose host stage is still running, if it has nothing
    to do with that picture (no slot in common, transitively) - and only then.  A stream with IRAPs handed over early (as a parsing-ahead host does);
    with worker threads, (a) slow I pictures: the pictures behind an IRAP in submission order overtake it; (b) slow B pictures: an IRAP, which depends
    on nothing, overtakes the B pictures submitted before it.  Either way every picture still finds in its reference slots exactly what it finds when
    everything is prepared and enqueued inline in submission order (the stand-in stamps a picture with a hash of its reference slots' stamps)"""
    W, H = 416, 240
    plans, nslots = stream.ra_plan(33, gop=8, seed_poc0_is_external=False, pool=40, intra_period=16, irap_lookahead=6)
    assert nslots >= len(plans)                         # no slot is reused: every stamp can be read at the end
    descs = [synth.picture_for_plan(pl, W, H, seed=531, tool_flags=TOOLS, p_intra=0.1) for pl in plans]
    pics = [d.c() for d in descs]
    stub.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
    stub.vvr_wait.argtypes = [C.c_void_p, C.c_int]
    stub.vvr_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    buf = (C.c_int * 60000)()
    results = {}
    stub.vvt_overtakes.restype = C.c_ulonglong
    stub.vvt_overtakes.argtypes = [C.c_void_p]
    for threads, slow, attempt in [(0, "I", 0)] + [(4, sl, a) for sl in ("I", "B") for a in range(3)]:
        if (threads, slow) in results:
            continue
        cfg = abi.Config()
        cfg.abi_version = abi.VVR_ABI_VERSION
        cfg.device, cfg.max_width, cfg.max_height, cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu = 0, W, H, 1, 10, 7
        cfg.num_slots, cfg.num_streams, cfg.host_threads = nslots, 3, threads
        ctx = C.c_void_p()
        assert stub.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_OK
        stub.vvt_take_trace(buf, len(buf))
        if slow == "I":
            stub.vvt_slow_i_pictures(20000)             # the host stage of the IRAPs after the first takes 20 ms longer here: the pictures behind them are ready first
        else:
            stub.vvt_slow_b_pictures(4000)              # ... of every B picture 4 ms longer: an IRAP is ready long before the B pictures in front of it
        jobs = [stub.vvr_submit(ctx, C.byref(p)) for p in pics]
        assert all(j >= 0 for j in jobs)
        for j in jobs:
            assert stub.vvr_wait(ctx, j) == abi.VVR_OK
        stub.vvt_slow_i_pictures(0)
        stub.vvt_slow_b_pictures(0)
        overtakes = stub.vvt_overtakes(ctx)
        if threads and not overtakes and attempt < 2:
            stub.vvr_destroy(ctx)           # (a busy machine: the "slow" pictures were not the slowest this time - once more)
            continue
        assert (overtakes > 0) == (threads > 0), overtakes          # pictures did pass others - and never without worker threads
        n = stub.vvt_take_trace(buf, len(buf))
        ops = [tuple(buf[i:i + 3]) for i in range(0, n, 3)]
        stamps = []
        row = np.zeros(W * H, np.uint16)
        for pl in plans:
            assert stub.vvr_read_plane(ctx, pl.slot, 0, row.ctypes.data_as(C.c_void_p), W) == abi.VVR_OK
            stamps.append(tuple(int(v) for v in row[:4]))
        results[(threads, slow)] = (stamps, ops)
        stub.vvr_destroy(ctx)
    assert results[(0, "I")][0] == results[(4, "I")][0] == results[(4, "B")][0], "a picture was enqueued before a picture it depends on"
    assert len(set(results[(0, "I")][0])) == len(plans)          # (the stamps tell the pictures apart)


def test_i_pictures_take_the_priority_lane_when_it_is_free(stub):
    """lanes: an I picture goes to the lane with the high-priority stream - unless the I picture before it is still running there, in which case it
    goes round the ordinary lanes like every other picture (a stream of I pictures only must not queue up on one lane).  Stream ids tell the lanes
    apart in the stand-in's trace: the picture's completion events are recorded on its lane's stream."""
    W, H = 64, 64
    plans, nslots = stream.ra_plan(7, gop=1, seed_poc0_is_external=False, pool=8, intra_period=1)       # I pictures only
    assert all(pl.slice_type == abi.SLICE_I for pl in plans)
    buf = (C.c_int * 30000)()
    lanes_used = {}
    for pending in (0, 1):
        ctx = Ctx(stub, W, H, nslots, log2_ctu=5, streams=3)
        stub.vvt_take_trace(buf, len(buf))
        stub.vvt_events_pending(pending)
        lanes = []
        for pl in plans:
            d = synth.picture_for_plan(pl, W, H, seed=540, tool_flags=TOOLS, log2_ctu=5)
            hnd = ctx.prepare(d)
            assert stub.vvr_submit_prepared(ctx.ctx, hnd) >= 0
            n = stub.vvt_take_trace(buf, len(buf))
            ops = [tuple(buf[i:i + 3]) for i in range(0, n, 3)]
            lanes.append([o for o in ops if o[0] == 1][0][1])
        stub.vvt_events_pending(0)
        stub.vvr_sync(ctx.ctx)
        lanes_used[pending] = lanes
    free, busy = lanes_used[0], lanes_used[1]
    assert len(set(free)) == 1                              # the device keeps up: every I picture finds the priority lane free
    assert busy[0] == min(busy[1:]) + 3                     # nothing finishes: only the first one gets it (the stream created after the three ordinary ones) ...
    assert len(set(busy[1:])) == 3                          # ... the others go round the three ordinary lanes


MC_ITEM_DT = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "u1"), ("h", "u1"), ("flags", "<u2"), ("cu", "<u4"), ("mv", "<i4", (2, 2)), ("ref", "i1", (2,)), ("bcw", "u1"), ("clipW4", "u1"), ("clipX", "<u2"), ("clipY", "<u2")])


def _mc_table(ctx, h, which, dt=MC_ITEM_DT):
    p, n = C.c_void_p(), C.c_size_t()
    ctx.L.vvt_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    assert ctx.L.vvt_table(h, which, C.byref(p), C.byref(n)) == 0
    return np.frombuffer(C.string_at(p.value, n.value), dt).copy() if n.value else np.zeros(0, dt)


def _sbtmvp_pieces(cu, motion, w4):
    """the blocks xSubPuMC predicts (InterPrediction.cpp:466-543), written down independently of the host glue: {(x, y) of a sub-block: (x, y, w) of its piece}"""
    ver = cu["h"] > cu["w"]
    n_f, n_s = ((cu["w"], cu["h"]) if ver else (cu["h"], cu["w"]))
    n_f, n_s = int(n_f) // 8, int(n_s) // 8
    pos = lambda f, s: (int(cu["x"]) + 8 * (f if ver else s), int(cu["y"]) + 8 * (s if ver else f))
    def mi(f, s):
        x, y = pos(f, s)
        m = motion[(y >> 2) * w4 + (x >> 2)]
        return tuple((int(m["ref_idx"][l]), tuple(int(v) for v in m["mv"][l]) if m["ref_idx"][l] >= 0 else None) for l in range(2))
    out = {}
    for f in range(n_f):
        s0 = 0
        while s0 < n_s:
            s1 = s0 + 1
            while s1 < n_s and mi(f, s1) == mi(f, s0):
                s1 += 1
            length = 8 * (s1 - s0)
            parts = [(0, length & ~15), (length & ~15, length & 15)] if (length > 16 and length & 15) else [(0, length)]
            for (o, l) in parts:
                for s in range(s0 + o // 8, s0 + (o + l) // 8):
                    px, py = pos(f, s0)
                    out[pos(f, s)] = (px, py + o, 8) if ver else (px + o, py, l)
            s0 = s1
    return out


def test_sbtmvp_tiles_carry_their_piece_under_wrap_around(stub):
    """reference wrap-around: wrapClipMv depends on position and WIDTH of the predicted block, which for SbTMVP is the run of equal sub-blocks xSubPuMC
    joins (cut at the largest multiple of 16) - every 8x8 tile of an SbTMVP CU names that piece (clipX, clipY, clipW4), tiles of other CUs and of pictures
    without wrap-around leave the width to the CU"""
    from test_oracle_vs_ref import ALL
    seen = {"joined": 0, "cut": 0, "vertical": 0, "single": 0}
    for (W, H, l2, seed, off) in [(384, 256, 6, 801, 368), (512, 256, 7, 802, 512), (384, 256, 5, 803, 320), (384, 256, 6, 804, 0)]:
        plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
        d = synth.picture_for_plan(plans[2], W, H, seed=seed, tool_flags=ALL, log2_ctu=l2, wrap_offset=off, p_intra=0.05, p_sbtmvp=0.6, mv_sigma=6.0)
        ctx = Ctx(stub, W, H, 8, log2_ctu=l2)
        hnd = ctx.prepare(d)
        tiles = _mc_table(ctx, hnd, 0)
        n_sbt = 0
        for k, cu in enumerate(d.cu):
            if cu["pred_mode"] != abi.PRED_INTER or cu["mc_mode"] != abi.MC_SBTMVP:
                continue
            mine = tiles[tiles["cu"] == k]
            assert len(mine) == (cu["w"] // 8) * (cu["h"] // 8)
            pieces = _sbtmvp_pieces(cu, d.motion, d.w4)
            for t in mine:
                n_sbt += 1
                if not off:
                    assert (t["clipX"], t["clipY"], t["clipW4"]) == (t["x"], t["y"], 0)
                    continue
                px, py, pw = pieces[(int(t["x"]), int(t["y"]))]
                assert (int(t["clipX"]), int(t["clipY"]), 4 * int(t["clipW4"])) == (px, py, pw), (k, t, (px, py, pw))
                seen["vertical" if cu["h"] > cu["w"] else "joined" if pw > 8 else "single"] += 1
                seen["cut"] += pw not in (8, 16, 32, 64, 128) or (px - int(cu["x"])) % 16 != 0
        assert n_sbt > 50
        assert np.all(tiles["clipW4"][~np.isin(tiles["cu"], np.nonzero(d.cu["mc_mode"] == abi.MC_SBTMVP)[0])] == 0)
        ctx.close()
    assert all(v > 0 for v in seen.values()), seen


def test_scaled_reference_pictures_in_the_host_glue(stub):
    """reference picture resampling: the tiles of every CU (SbTMVP: sub-block) that reads a scaled reference picture are on the list of k_mc_rpr and on
    no other, they cover those CUs exactly; a picture smaller than the context's is accepted; inconsistent tables are refused"""
    from test_oracle_vs_ref import RPR_CASES, rpr_case, ALL
    assert MC_ITEM_DT.itemsize == 36
    for (W, H, l2, idx, seed, specs, win, colloc, kw) in RPR_CASES:
        kw = dict(kw)
        tools = ALL | kw.pop("tool_flags_extra", 0)
        d, refs = rpr_case(W, H, l2, idx, seed, specs, win=win, colloc=colloc, tools=tools, **kw)
        MW = max([W] + [r[0].shape[1] for r in refs.values()]); MH = max([H] + [r[0].shape[0] for r in refs.values()])
        ctx = Ctx(stub, MW, MH, 8, log2_ctu=l2, bit_depth=d.hdr.bit_depth, chroma_format=d.hdr.chroma_format)
        hnd = ctx.prepare(d)
        rpr = _mc_table(ctx, hnd, 11)
        others = np.concatenate([_mc_table(ctx, hnd, 0), _mc_table(ctx, hnd, 3)])
        dev_cus = _mc_table(ctx, hnd, 10, np.dtype([("cu", "<u4"), ("first", "<u4")]))       # CUs whose tiles the device writes
        scaled = lambda l, i: i >= 0 and bool(d.rpr.ref[l][i].scaled)
        area = 0
        for k, cu in enumerate(d.cu):
            if cu["pred_mode"] != abi.PRED_INTER:
                continue
            mine = rpr[rpr["cu"] == k]
            if cu["mc_mode"] == abi.MC_SBTMVP:
                for t in mine:
                    assert scaled(0, t["ref"][0]) or scaled(1, t["ref"][1])
                for t in others[others["cu"] == k]:
                    assert not (scaled(0, t["ref"][0]) or scaled(1, t["ref"][1]))
                assert int((mine["w"].astype(int) * mine["h"]).sum()) + int((others[others["cu"] == k]["w"].astype(int) * others[others["cu"] == k]["h"]).sum()) == int(cu["w"]) * int(cu["h"])
            else:
                if cu["mc_mode"] == abi.MC_GEO:
                    s = any(scaled((int(g) >> 4) - 1, int(g) & 15) for g in cu["geo_dir_ref"])
                else:
                    s = scaled(0, cu["ref_idx"][0]) or scaled(1, cu["ref_idx"][1])
                assert (len(mine) > 0) == s, "CU %d" % k
                if s:
                    assert int((mine["w"].astype(int) * mine["h"]).sum()) == int(cu["w"]) * int(cu["h"]) and k not in dev_cus["cu"] and not (others["cu"] == k).any()
                    assert bool(mine["flags"][0] & 16) == (cu["mc_mode"] == abi.MC_AFFINE)
            area += int((mine["w"].astype(int) * mine["h"]).sum())
        assert area > 0
        stub.vvr_free_prepared(ctx.ctx, hnd)
        # refusals
        keep = abi.RprParams.from_buffer_copy(d.rpr)
        l, i = [(l, i) for l in range(2) for i in range(d.hdr.num_ref[l]) if d.rpr.ref[l][i].scaled][0]
        d.rpr.ref[l][i].ratio[0] = (1 << 15) + 1
        _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "scaling ratio")
        d.rpr = abi.RprParams.from_buffer_copy(keep); d.rpr.ref[l][i].scaled = 0
        _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "is a scaled one")
        d.rpr = abi.RprParams.from_buffer_copy(keep); d.rpr.ref[l][i].width = MW + 8
        _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "reference picture size")
        d.rpr = abi.RprParams.from_buffer_copy(keep); d.hdr.wrap_offset = W
        _expect_error(ctx, d, abi.VVR_ERR_UNSUPPORTED, "wrap-around")
        d.hdr.wrap_offset = 0
        bi = np.nonzero((d.cu["pred_mode"] == abi.PRED_INTER) & (d.cu["mc_mode"] == abi.MC_BI) & (d.cu["w"] >= 8) & (d.cu["h"] >= 8) & (d.cu["w"].astype(int) * d.cu["h"] >= 128) & (d.cu["bcw_idx"] == 2) & ((d.cu["flags"] & abi.CU_CIIP) == 0))[0]
        bi = [k for k in bi if scaled(0, d.cu["ref_idx"][k][0]) or scaled(1, d.cu["ref_idx"][k][1])]
        if bi and d.wp is None:
            d.cu["mc_mode"][bi[0]] = abi.MC_BDOF
            _expect_error(ctx, d, abi.VVR_ERR_PARAMETER, "scaled reference picture")
        ctx.close()


@pytest.mark.parametrize("threads", [1, 5])
def test_read_picture_lays_the_rows_out_at_the_callers_strides(stub, threads):
    """vvr_read_picture: every plane of the picture in a slot at the picture's own size (vvr_slot_picture_size), rows at the caller's strides and
    nothing written past them; staging buffers allocated with the context (vvr_config.read_buffers) or with the first call"""
    W, H = 96, 64
    stub.vvr_write_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    stub.vvr_read_picture.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(5)
    for pre in (0, 2):
        cfg = abi.Config()
        cfg.abi_version, cfg.max_width, cfg.max_height, cfg.chroma_format, cfg.bit_depth, cfg.log2_ctu, cfg.num_slots, cfg.num_streams, cfg.read_buffers = abi.VVR_ABI_VERSION, W, H, 1, 10, 5, 2, 1, pre
        ctx = C.c_void_p()
        assert stub.vvr_create(C.byref(cfg), C.byref(ctx)) == abi.VVR_OK
        for (w, h) in ((W, H), (64, 40)):
            assert stub.vvr_slot_picture_size(ctx, 1, w, h) == abi.VVR_OK
            planes = [rng.integers(0, 1024, (h >> s, w >> s)).astype(np.uint16) for s in (0, 1, 1)]
            for c, pl in enumerate(planes):
                assert stub.vvr_write_plane(ctx, 1, c, pl.ctypes.data, pl.shape[1]) == abi.VVR_OK
            outs = [np.full((h >> s, (w >> s) + 7), 0xffff, np.uint16) for s in (0, 1, 1)]
            dst = (C.c_void_p * 3)(*[o.ctypes.data for o in outs]); strides = (C.c_size_t * 3)(*[o.shape[1] for o in outs])
            assert stub.vvr_read_picture(ctx, 1, dst, strides, threads) == abi.VVR_OK
            for o, pl in zip(outs, planes):
                assert np.array_equal(o[:, :pl.shape[1]], pl) and (o[:, pl.shape[1]:] == 0xffff).all()
        short = (C.c_size_t * 3)(8, 8, 8)
        assert stub.vvr_read_picture(ctx, 1, dst, short, threads) == abi.VVR_ERR_PARAMETER
        assert stub.vvr_slot_picture_size(ctx, 1, W + 8, H) == abi.VVR_ERR_PARAMETER
        stub.vvr_destroy(ctx)


def test_transform_block_items_say_what_the_kernel_must_know_of_the_cu(stub):
    """k_itrans asks for the CU record, the coded levels, the basis rows and the prediction samples in ONE memory round trip; what those loads depend on of the CU
    it reads from the item (TbItem::pad, vvr_device.h): bit 0 - a chroma block of an ISP coding unit takes position and size from the CU; bit 1 - BDPCM (every level of
    the block is coded); bit 2 - the CU applies LFNST to this component (lfnst_idx > 0, and not the chroma of a single-tree CU).  Every item of an I picture with
    these tools on, against the records"""
    TB_DT = np.dtype([("tu", "<u4"), ("comp", "u1"), ("mode", "u1"), ("ict", "u1"), ("pad", "u1")])
    W, H = 416, 240
    plans, nslots = stream.ra_plan(1, gop=1, seed_poc0_is_external=False)
    seen = {1: 0, 2: 0, 4: 0}
    for seed, kw in ((91, dict(p_isp=0.5, p_lfnst=0.6, p_coded=0.9, p_coded_chroma=0.8)), (92, dict(p_bdpcm=0.5, p_lfnst=0.5, p_coded=0.9, p_coded_chroma=0.8, dual_tree=1.0)), (93, dict(p_isp=0.3, p_bdpcm=0.3, p_lfnst=0.5, p_coded=0.9, p_coded_chroma=0.9))):
        d = synth.picture_for_plan(plans[0], W, H, seed=seed, tool_flags=TOOLS, log2_ctu=6, **kw)
        ctx = Ctx(stub, W, H, max(nslots, 2), log2_ctu=6)
        h = ctx.prepare(d)
        stub.vvt_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n_items = 0
        for which in (4, 5, 6):
            p, n = C.c_void_p(), C.c_size_t()
            assert stub.vvt_table(h, which, C.byref(p), C.byref(n)) == 0
            items = np.frombuffer(C.string_at(p.value, n.value), TB_DT) if n.value else np.zeros(0, TB_DT)
            for it in items:
                tu = d.tu[int(it["tu"])]
                cu = d.cu[int(tu["cu"])]
                comp = int(it["comp"])
                want = 0
                if comp and int(cu["isp_mode"]):
                    want |= 1
                if int(cu["bdpcm"][1 if comp else 0]):
                    want |= 2
                if int(cu["lfnst_idx"]) and (int(cu["tree"]) != abi.TREE_JOINT or comp == 0):
                    want |= 4
                assert int(it["pad"]) == want, (seed, which, int(it["tu"]), comp, int(it["pad"]), want)
                for b in (1, 2, 4):
                    seen[b] += 1 if want & b else 0
                n_items += 1
        assert n_items > 100
        stub.vvr_free_prepared(ctx.ctx, h)
        ctx.close()
    assert seen[1] and seen[2] and seen[4], seen        # (chroma blocks of ISP coding units, BDPCM blocks and LFNST blocks were among them)
