"""CPU: round trip of the reference-side glue (integration/vvr_extract.h).

A generated description is turned into the reference decoder's own objects by the test harness (oracle/ref_harness.cpp: SPS / PPS /
picture header / slice / APSs / CodingStructure with CUs, TUs, motion, edge parameters, SAO / ALF CTU data, levels in the
reconstruction buffer -- the state the reference is in after parsing, MIDER and LF_INIT).  The extractor walks those objects with the
reference's own helpers and writes a description again; every field must come back: the "derived" ones (final intra modes, transform
types from TrQuant::getTrTypes, the motion-compensation branch, CIIP neighbour flags, DMVR offsets, LMCS tables rebuilt by the
reference's Reshape class, final ALF filters after reconstructCoeffAPSs, SAO offsets) are recomputed on the way, not copied.

Needs oracle/_ref (built where /root/reference exists); skipped elsewhere."""
import ctypes as C
import os
import numpy as np
import pytest

import refdrv
from vvdec_amd import abi, synth, stream

pytestmark = pytest.mark.skipif(not refdrv.available(), reason="oracle/_ref not built")
ALL = (abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST | abi.TOOL_BDOF |
       abi.TOOL_DMVR | abi.TOOL_PROF)
LM = abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE


def _fields_differ(a, b, what):
    out = []
    assert len(a) == len(b), "%s: %d vs %d entries" % (what, len(a), len(b))
    for name in a.dtype.names:
        if name.startswith("pad"):
            continue
        if not np.array_equal(a[name], b[name]):
            idx = np.nonzero((a[name] != b[name]).reshape(len(a), -1).any(axis=1))[0]
            out.append("%s.%s differs in %d entries, first #%d: %s vs %s" % (what, name, len(idx), idx[0], a[name][idx[0]], b[name][idx[0]]))
    return out


def _struct_differ(a, b, what, skip=()):
    out = []
    for (name, _) in a._fields_:
        if name.startswith("pad") or name in skip:
            continue
        v0, v1 = getattr(a, name), getattr(b, name)
        if hasattr(v0, "__len__"):
            v0, v1 = np.ctypeslib.as_array(v0), np.ctypeslib.as_array(v1)
            if not np.array_equal(v0, v1):
                out.append("%s.%s differs" % (what, name))
        elif v0 != v1:
            out.append("%s.%s: %s vs %s" % (what, name, v0, v1))
    return out


def _alf_differ(a0, a1, h0):
    bad = []
    na = a0.num_luma_aps
    if a1.num_luma_aps != na:
        bad.append("alf_params.num_luma_aps")
    for name in ("luma_coeff", "luma_clip"):
        if not np.array_equal(np.ctypeslib.as_array(getattr(a0, name))[:na], np.ctypeslib.as_array(getattr(a1, name))[:na]):
            bad.append("alf_params.%s differs" % name)
    if h0.chroma_format:
        names = ("chroma_coeff", "chroma_clip") + (("ccalf_coeff",) if h0.tool_flags & abi.TOOL_CCALF else ())
        for name in names:
            if not np.array_equal(np.ctypeslib.as_array(getattr(a0, name)), np.ctypeslib.as_array(getattr(a1, name))):
                bad.append("alf_params.%s differs" % name)
    return bad


def _wp_differ(w0, w1, h0):
    bad = []
    if list(w0.log2_denom) != list(w1.log2_denom):
        bad.append("wp.log2_denom")
    for l in range(2):
        for i in range(h0.num_ref[l]):
            for c in range(3):
                a, b = w0.e[l][i][c], w1.e[l][i][c]
                if (a.weight, a.offset, a.present) != (b.weight, b.offset, b.present):
                    bad.append("wp.e[%d][%d][%d]" % (l, i, c))
    return bad


def _compare(d, e, sets=True):
    bad = []
    h0, h1 = d.hdr, e["hdr"]
    # (the harness always switches the SPS MTS flag on and resolves the transform types per TU, so that bit is not a property of the stream)
    f0, f1 = h0.tool_flags | abi.TOOL_MTS, h1.tool_flags | abi.TOOL_MTS
    if f0 != f1:
        bad.append("hdr.tool_flags %x vs %x" % (f0, f1))
    # (with slice headers the picture header's own deblocking offsets are not used: the extractor puts the first slice's there)
    bad += _struct_differ(h0, h1, "hdr", skip=("tool_flags",) + (() if sets else ("deblock_beta_offset_div2", "deblock_tc_offset_div2")))
    bad += _fields_differ(d.cu, e["cu"], "cu")
    bad += _fields_differ(d.tu, e["tu"], "tu")
    if not np.array_equal(d.ctu_first_cu, e["ctu_first_cu"]):
        bad.append("ctu_first_cu differs")
    n = max(len(d.coef), len(e["coef"]))
    c0, c1 = np.zeros(n, np.int16), np.zeros(n, np.int16)
    c0[:len(d.coef)], c1[:len(e["coef"])] = d.coef, e["coef"]
    if not np.array_equal(c0, c1):
        bad.append("coef differs")
    bad += _fields_differ(d.motion, e["motion"], "motion")
    for k in range(2):
        bad += _fields_differ(d.lfp[k], e["lfp"][k], "lfp%d" % k)
    if h0.tool_flags & (abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA):
        bad += _fields_differ(d.sao, e["sao"], "sao")
    if h0.tool_flags & abi.TOOL_ALF:
        bad += _fields_differ(d.alf, e["alf"], "alf")
        if sets:
            bad += _alf_differ(d.alf_params, e["alf_params"], h0)
    if h0.tool_flags & abi.TOOL_LMCS:
        l0, l1 = d.lmcs, e["lmcs"]
        nv = 1 << h0.bit_depth
        for name in ("fwd_lut", "inv_lut"):
            if not np.array_equal(np.ctypeslib.as_array(getattr(l0, name))[:nv], np.ctypeslib.as_array(getattr(l1, name))[:nv]):
                bad.append("lmcs.%s differs" % name)
        bad += _struct_differ(l0, l1, "lmcs", skip=("fwd_lut", "inv_lut"))
    if sets and (h0.tool_flags & abi.TOOL_WP) and h0.slice_type != abi.SLICE_I:
        bad += _wp_differ(d.wp, e["wp"], h0)
    if h0.tool_flags & abi.TOOL_SCALING_LIST:
        bad += _struct_differ(d.scaling, e["scaling"], "scaling")
    # slices, tiles (the index of a slice / tile is what the reference numbers them with: compared as partitions), sub-pictures
    for name in ("ctu_slice", "ctu_tile"):
        a, b = getattr(d, name), e[name]
        if (a is None) != (b is None):
            bad.append("%s present %s vs %s" % (name, a is not None, b is not None))
        elif a is not None and not np.array_equal(a[:, None] == a[None, :], b[:, None] == b[None, :]):
            bad.append("%s: another partition" % name)
    if (d.subpics is None) != (e["subpics"] is None):
        bad.append("subpics present %s vs %s" % (d.subpics is not None, e["subpics"] is not None))
    elif d.subpics is not None:
        bad += _fields_differ(np.ascontiguousarray(d.subpics), e["subpics"], "subpics")
    return bad


CASES = [
    ("intra_tools", 256, 128, 7, 0, 601, ALL, dict(p_cclm=0.3, p_mip=0.2, p_isp=0.2, p_lfnst=0.3, p_bdpcm=0.1)),
    ("inter_tools_lmcs", 384, 256, 6, 2, 602, ALL | LM, dict(p_intra=0.25, p_affine=0.2, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.1, p_sbt=0.1, p_bcw=0.2, p_jccr=0.3, p_coded_chroma=0.5)),
    ("key_picture", 256, 192, 7, 1, 603, ALL | LM | abi.TOOL_JCCR_SIGN | abi.TOOL_STILL_REF, dict(p_intra=0.2, p_affine=0.1, p_jccr=0.4, p_coded_chroma=0.6)),
    ("weighted_prediction", 256, 128, 6, 3, 604, ALL | abi.TOOL_WP, dict(p_intra=0.1, p_affine=0.2, p_sbtmvp=0.15, p_ciip=0.1)),
    ("scaling_lists", 256, 128, 7, 2, 605, ALL | abi.TOOL_SCALING_LIST | abi.TOOL_SCALING_LIST_NO_LFNST, dict(p_intra=0.3, p_coded=0.8, p_coded_chroma=0.6, p_lfnst=0.5, p_sbt=0.3)),
    ("dual_tree_implicit_mts", 256, 128, 6, 0, 606, ALL | LM | abi.TOOL_IMPLICIT_MTS, dict(dual_tree=2.0, p_cclm=0.3, p_lfnst=0.4, p_isp=0.3, p_mip=0.3, p_coded=0.7, p_coded_chroma=0.5, p_split_scale=1.5)),
    ("small_cus_local_dual_tree", 264, 200, 7, 3, 607, ALL | LM, dict(min_cu_log2=2, p_intra=0.3, p_split_scale=1.8, p_cclm=0.3, p_isp=0.2, p_sbt=0.2, p_ciip=0.3, p_coded_chroma=0.5)),
    ("intra_block_copy", 384, 256, 6, 0, 608, ALL | LM | abi.TOOL_IBC | abi.TOOL_CCLM_COLLOC, dict(p_ibc=0.5, p_split_scale=1.4, p_cclm=0.3, p_jccr=0.3, p_coded_chroma=0.5)),
    ("ibc_b_picture", 264, 200, 7, 2, 609, ALL | abi.TOOL_IBC, dict(p_ibc=0.6, p_intra=0.4, min_cu_log2=2, p_split_scale=1.6)),
    ("8bit", 256, 128, 6, 2, 610, ALL | LM, dict(bit_depth=8, p_intra=0.25, p_affine=0.15, p_geo=0.1, p_ciip=0.1, p_mip=0.2, p_isp=0.2, p_cclm=0.3)),
    ("monochrome", 256, 128, 7, 3, 611, ALL | abi.TOOL_LMCS, dict(chroma_format=0, p_intra=0.3, p_affine=0.2, p_mip=0.2, p_isp=0.2)),
    ("no_filters", 200, 136, 5, 2, 612, abi.TOOL_DEP_QUANT | abi.TOOL_DEBLOCK_OFF, dict(p_intra=0.2)),
    ("ladf_virtual_boundaries_wrap_around", 384, 256, 6, 2, 613, ALL | abi.TOOL_LADF, dict(p_intra=0.2, p_affine=0.2, virtual_boundaries=2 | (2 << 2) | 16, wrap_offset=384)),
    ("slices_tiles_subpictures", 512, 384, 6, 3, 614, ALL | abi.TOOL_NO_LF_ACROSS_SLICES, dict(p_intra=0.2, tile_cols=2, tile_rows=2, subpics=1 | (2 << 1) | (2 << 3))),
    ("raster_slices_over_tiles", 512, 384, 6, 1, 615, ALL | abi.TOOL_NO_LF_ACROSS_TILES, dict(p_intra=0.2, num_slices=3, tile_cols=2, tile_rows=2)),
]


@pytest.mark.parametrize("name,W,H,l2,idx,seed,tools,kw", CASES, ids=[c[0] for c in CASES])
def test_description_survives_the_reference_objects(built, name, W, H, l2, idx, seed, tools, kw):
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, log2_ctu=l2, **kw)
    refs = {}
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc, bit_depth=kw.get("bit_depth", 10)))
    e = refdrv.extract(d, refs)
    bad = _compare(d, e)
    assert not bad, "\n".join(bad)
    assert e["num_dmvr"] == d.num_dmvr


def test_p_slice(built):
    W, H = 256, 128
    p = synth.default_params(width=W, height=H, seed=613, tool_flags=ALL | abi.TOOL_WP, slice_type=abi.SLICE_P, p_affine=0.2, p_sbtmvp=0.15, p_ciip=0.1, p_intra=0.1)
    p.poc, p.out_slot = 4, 0
    synth.set_refs(p, [(1, 0), (2, 8)])
    d = synth.generate(p)
    refs = {1: synth.natural_picture(W, H, 614), 2: synth.natural_picture(W, H, 615)}
    bad = _compare(d, refdrv.extract(d, refs))
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("name,W,H,l2,idx,seed,tools,kw", CASES, ids=[c[0] for c in CASES])
def test_edge_tables_of_the_reference_drive_the_oracle(built, name, W, H, l2, idx, seed, tools, kw):
    """the way a real integration runs: the reference derives the edge parameters itself (LF_INIT, calcFilterStrengthsCTU), the extractor
    copies its raw tables (including bits the filters never look at), and the back-end's arithmetic (here: the oracle restatement) must
    reproduce the reference's output from them"""
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, log2_ctu=l2, **kw)
    refs = {}
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc, bit_depth=kw.get("bit_depth", 10)))
    e = refdrv.extract(d, refs, flags=refdrv.DERIVE_LFP)
    want = refdrv.reconstruct(d, refs, flags=refdrv.DERIVE_LFP)["planes"]
    d.lfp = [e["lfp"][0], e["lfp"][1]]
    got = refdrv.oracle_reconstruct(d, refs)
    for c in range(len(got)):
        assert np.array_equal(got[c], want[c]), "comp %d: %d differ" % (c, int((got[c] != want[c]).sum()))


def test_reference_edge_tables_are_safe_for_one_launch_per_direction(built):
    """k_deblock filters all edges of a direction in one launch: on the tables the reference derives itself, neighbouring luma edges touch
    disjoint samples.  (Until round 4 the harness built SbTMVP CUs without the affine flag the parser leaves on every sub-block merge CU; the
    reference then allowed a 7-sample P side 8 samples behind a sub-block edge of such a CU, LoopFilter.cpp:920 - a pair the kernels still order
    themselves, and one that no parsed stream produces.)"""
    from test_host_logic import _edge_reach
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    ordered = 0
    for (name, W, H, l2, idx, seed, tools, kw) in CASES + [("sbtmvp", 384, 256, 7, 1, 620, ALL, dict(p_intra=0.05, p_affine=0.3, p_sbtmvp=0.4, p_split_scale=0.5))]:
        if tools & abi.TOOL_DEBLOCK_OFF:
            continue
        pl = plans[idx]
        d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, log2_ctu=l2, **kw)
        refs = {}
        for lst in pl.ref_slots:
            for (slot, poc) in lst:
                refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc, bit_depth=kw.get("bit_depth", 10)))
        e = refdrv.extract(d, refs, flags=refdrv.DERIVE_LFP)
        ctu = 1 << l2
        for dr in range(2):
            lf = e["lfp"][dr].reshape(d.h4, d.w4)
            for line in (lf if dr == 0 else lf.T):
                edges = [(b * 4, _edge_reach(l, dr, b * 4, ctu), l) for b, l in enumerate(line) if int(l["bs"]) & 3]
                for (e0, r0, l0), (e1, r1, l1) in zip(edges, edges[1:]):
                    assert not (e0 + r0[1] - 1 >= e1 - r1[2] or e1 - r1[0] <= e0 + r0[3] - 1), (name, dr, e0, e1)
                    ordered += 1
    assert ordered > 1000          # (pairs of neighbouring edges looked at)


def _sets_differ(want, got, what, used, cmp):
    """tables selected by the slices: compared through the slices' choice (the extractor numbers distinct tables in the order it meets them)"""
    return ["%s of slice %d differs" % (what, k) for k, (a, b) in enumerate(used) if cmp(want[a], got[b])]


@pytest.mark.parametrize("idx,seed,tools,kw,rotate", [
    (0, 641, ALL | LM | abi.TOOL_SCALING_LIST, dict(num_slices=3, p_cclm=0.3, p_coded=0.8), False),
    (2, 642, ALL | LM | abi.TOOL_WP | abi.TOOL_NO_LF_ACROSS_SLICES, dict(num_slices=4, p_intra=0.3, p_ciip=0.2, p_affine=0.2, p_geo=0.2, p_sbtmvp=0.1), False),
    (2, 643, ALL | LM | abi.TOOL_WP, dict(num_slices=4, p_intra=0.2, p_affine=0.2, p_geo=0.2, p_sbtmvp=0.2, p_bcw=0.2), True),
    (3, 644, ALL | abi.TOOL_SCALING_LIST | abi.TOOL_WP | abi.TOOL_NO_LF_ACROSS_TILES, dict(num_slices=5, tile_cols=3, tile_rows=2, p_intra=0.2, p_geo=0.3), True),
    (2, 645, ALL | LM | abi.TOOL_WP, dict(num_slices=4, intra_slices=0b0101, p_intra=0.2, p_affine=0.2, p_geo=0.2, p_cclm=0.3), True),      # I slices (the first one too) among B slices
])
def test_slices_with_headers_of_their_own_survive_the_reference_objects(built, idx, seed, tools, kw, rotate):
    """every slice of the reference's picture carries its own header (quantisation / LMCS / scaling-list switches, deblocking offsets, APS ids,
    weights) and - with `rotate` - its own reference picture lists (the description's lists rotated by the slice index, CUs / motion / GPM
    partitions / weights numbered to match): the extractor fills vvr_picture::slices, merges the lists into their union and renumbers every
    reference index, so the original description comes back"""
    W, H, l2 = 512, 384, 6
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, log2_ctu=l2, **kw)
    synth.vary_slices(d, seed, intra_slices=kw.get("intra_slices", 0))
    refs = {slot: synth.natural_picture(W, H, seed + 100 + poc) for lst in pl.ref_slots for (slot, poc) in lst}
    if rotate:
        assert max(d.hdr.num_ref[0], d.hdr.num_ref[1]) > 1
    fl = refdrv.ROTATE_REF_LISTS if rotate else 0
    e = refdrv.extract(d, refs, flags=fl)
    # the union comes out in the order the slices list the pictures: the description's own order unless the first slices are rotated or I slices
    same_order = all(list(e["hdr"].ref_poc[l])[:d.hdr.num_ref[l]] == list(d.hdr.ref_poc[l])[:d.hdr.num_ref[l]] for l in range(2))
    if same_order:
        bad = _compare(d, e, sets=False)
    else:
        bad = []
        for l in range(2):      # a permutation of the description's lists, and every reference index follows it
            n = d.hdr.num_ref[l]
            assert e["hdr"].num_ref[l] == n
            perm = [[(e["hdr"].ref_poc[l][j], e["hdr"].ref_slot[l][j]) for j in range(n)].index((d.hdr.ref_poc[l][i], d.hdr.ref_slot[l][i])) for i in range(n)]
            lut = np.array(perm + [-1], np.int8)          # (index -1 stays -1)
            inter = d.cu["pred_mode"] == abi.PRED_INTER
            if not np.array_equal(lut[d.cu["ref_idx"][inter][:, l]], e["cu"]["ref_idx"][inter][:, l]):
                bad.append("cu.ref_idx of list %d does not follow the union" % l)
            if not np.array_equal(lut[d.motion["ref_idx"][:, l]], e["motion"]["ref_idx"][:, l]):
                bad.append("motion.ref_idx of list %d does not follow the union" % l)
    # what the extractor wrote reconstructs to the reference's picture (the oracle stands in for the back-end)
    want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
    got = refdrv.oracle_reconstruct(refdrv.desc_from_extract(e), refs)
    for c in range(3):
        assert np.array_equal(got[c], want[c]), "comp %d: %d samples of the extracted description differ from the reference's picture" % (c, int((got[c] != want[c]).sum()))
    assert same_order or rotate or kw.get("intra_slices", 0) & 1
    # the slice headers (the slice numbering is the reference's: compared through the CTU map)
    assert e["slices"] is not None and len(e["slices"]) == len(d.slices)
    pairs = sorted(set(zip(d.ctu_slice.tolist(), e["ctu_slice"].tolist())))
    assert len(pairs) == len(d.slices)
    for a, b in pairs:
        s0, s1 = d.slices[a], e["slices"][b]
        for name in ("tool_flags", "deblock_beta_offset_div2", "deblock_tc_offset_div2", "slice_type"):
            v0, v1 = s0[name], s1[name]
            if name == "tool_flags":
                if d.hdr.slice_type == abi.SLICE_I:
                    v0, v1 = int(v0) & ~abi.TOOL_WP, int(v1) & ~abi.TOOL_WP
            if not np.array_equal(v0, v1):
                bad.append("slice %d: %s %s vs %s" % (a, name, v0, v1))
    if d.hdr.tool_flags & abi.TOOL_ALF:
        used = [(int(d.slices["alf_set"][a]), int(e["slices"]["alf_set"][b])) for a, b in pairs]
        bad += _sets_differ(d.alf_sets, e["alf_sets"], "ALF table", used, lambda x, y: bool(_alf_differ(x, y, d.hdr)))
    if same_order and (d.hdr.tool_flags & abi.TOOL_WP) and d.hdr.slice_type != abi.SLICE_I:      # (tables are indexed like the lists; another order: the reconstruction above covers them)
        used = [(int(d.slices["wp_set"][a]), int(e["slices"]["wp_set"][b])) for a, b in pairs if d.slices["slice_type"][a] != abi.SLICE_I]      # (an I slice has no weights)
        bad += _sets_differ(d.wp_sets, e["wp_sets"], "weight table", used, lambda x, y: bool(_wp_differ(x, y, d.hdr)))
    assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("feature,text", [(1, "LADF with more than 5"), (2, "wrap-around motion compensation with a period off"), (3, "virtual boundary off the 8-sample grid"), (4, "missing reference picture"), (5, "sub-pictures together with reference wrap-around"),
                                          (6, "colour transform"), (7, "bit depth"), (8, "more slices or tiles"), (9, "scaled reference picture together with reference wrap-around")])
def test_extractor_refuses_what_the_description_cannot_express(built, feature, text):
    """the reference-side glue never flattens a picture into something it is not: LADF, wrap-around, virtual boundaries, several slices / tiles /
    sub-pictures, ACT, more than 10 bits, scaled references in a picture with wrap-around are refused with VVR_ERR_UNSUPPORTED (the binding raises the reference's own
    'not supported' error); the same picture without the feature is accepted"""
    L = refdrv.lib()
    W, H = 256, 128
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[1]
    d = synth.picture_for_plan(pl, W, H, seed=631, tool_flags=ALL, p_intra=0.2)
    refs = {slot: synth.natural_picture(W, H, 632 + poc) for lst in pl.ref_slots for (slot, poc) in lst}
    p = d.c()
    nslots = max(refs) + 1
    ref_ptrs = (C.POINTER(C.c_uint16) * (nslots * 3))()
    keep = []
    for slot, planes in refs.items():
        for c, plane in enumerate(planes):
            a = np.ascontiguousarray(plane, dtype=np.uint16); keep.append(a)
            ref_ptrs[slot * 3 + c] = a.ctypes.data_as(C.POINTER(C.c_uint16))
    why = C.create_string_buffer(256)
    assert L.vvref_check_expressible(C.byref(p), ref_ptrs, 0, why, 256) == abi.VVR_OK, why.value
    rc = L.vvref_check_expressible(C.byref(p), ref_ptrs, feature, why, 256)
    assert rc == abi.VVR_ERR_UNSUPPORTED and text in why.value.decode(), (rc, why.value)


def test_binding_executes_on_the_stand_in_runtime(built):
    """integration/DecLibReconAmd.h instantiated and run: reference-built objects of a picture -> LF_INIT by the reference ->
    extractor -> vvr_submit -> vvr_wait -> DMVR delta MVs back through DecCu::TaskFinishMotionInfo.  Here the back-end is the product's host code
    on the stand-in runtime (no sample is computed, delta MVs are zero): the description the extractor writes from the reference's own state
    passes the product's validation and the work-list builder, and the motion field that comes back is the one the reference derives when no
    refinement moves anything.  The same test with samples runs on the GPU (tests/test_gpu_parity.py)."""
    import test_host_glue as T
    if not refdrv.binding_available() or not os.path.exists(os.path.join(T.HIP_INC, "hip", "hip_runtime_api.h")):
        pytest.skip("binding harness not built or HIP headers not installed")
    T.build_stub()
    W, H = 256, 128
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    for idx, tools, kw, fl in ((0, ALL, dict(p_cclm=0.3, p_mip=0.2), 0), (2, ALL | abi.TOOL_STILL_REF | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_intra=0.2, p_affine=0.2, p_sbtmvp=0.1, p_ciip=0.1), 0),
                               # slices with headers and reference picture lists of their own (the harness rotates the lists per slice, the extractor merges them)
                               (2, ALL | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_WP | abi.TOOL_SCALING_LIST, dict(num_slices=3, log2_ctu=5, p_intra=0.2, p_affine=0.2, p_geo=0.2, p_sbtmvp=0.1), refdrv.ROTATE_REF_LISTS)):
        pl = plans[idx]
        d = synth.picture_for_plan(pl, W, H, seed=641 + idx, tool_flags=tools, **kw)
        if fl:
            synth.vary_slices(d, 660)
        refs = {slot: synth.natural_picture(W, H, 650 + poc) for lst in pl.ref_slots for (slot, poc) in lst}
        planes, motion = refdrv.run_binding(d, refs, T.LIB, flags=fl)
        assert [p.shape for p in planes] == [d.plane_shape(c) for c in range(3)]
        if pl.slice_type != abi.SLICE_I:
            inter = d.motion["ref_idx"].max(axis=1) >= 0
            assert inter.any()
            # no refinement on the stand-in: DMVR CUs keep their MVs, everything else is the field the description was built from
            assert np.array_equal(motion["ref_idx"][inter], d.motion["ref_idx"][inter])
            l0 = inter & (d.motion["ref_idx"][:, 0] >= 0)
            assert np.array_equal(motion["mv"][l0][:, 0], d.motion["mv"][l0][:, 0])


def _rpr_differ(a, b, h):
    bad = []
    if (a is None) != (b is None):
        return ["rpr present %s vs %s" % (a is not None, b is not None)]
    if a is None:
        return bad
    if (a.win_left, a.win_top) != (b.win_left, b.win_top):
        bad.append("rpr window %s vs %s" % ((a.win_left, a.win_top), (b.win_left, b.win_top)))
    for l in range(2):
        for i in range(h.num_ref[l]):
            bad += _struct_differ(a.ref[l][i], b.ref[l][i], "rpr.ref[%d][%d]" % (l, i))
    return bad


def test_scaled_reference_pictures_survive_the_reference_objects(built):
    """reference picture resampling: the extractor writes the table (ratios of the slices, windows and sizes of the PPSs, chroma sample location of
    the reference pictures' SPS) from the reference's objects, decides BDOF / DMVR like the reference does for CUs with a scaled reference
    picture, and what it wrote reconstructs to the reference's picture"""
    from test_oracle_vs_ref import RPR_CASES, rpr_case
    for (W, H, l2, idx, seed, specs, win, colloc, kw) in RPR_CASES:
        kw = dict(kw)
        tools = ALL | kw.pop("tool_flags_extra", 0)
        d, refs = rpr_case(W, H, l2, idx, seed, specs, win=win, colloc=colloc, tools=tools, **kw)
        e = refdrv.extract(d, refs)
        bad = _compare(d, e) + _rpr_differ(d.rpr, e["rpr"], d.hdr)
        assert not bad, "seed %d\n" % seed + "\n".join(bad)
        want = refdrv.reconstruct(d, refs)["planes"]
        got = refdrv.oracle_reconstruct(refdrv.desc_from_extract(e), refs)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))


def test_scaled_reference_pictures_with_slices_that_list_them_differently(built):
    """reference picture resampling together with slices whose headers and reference picture lists differ: the harness rotates the lists per slice
    (a picture is entry 0 of one slice and entry 1 of the next), the extractor merges them into the union and writes the table over the union;
    what it wrote reconstructs (oracle) to the picture the reference's own stages produce from those objects"""
    from test_oracle_vs_ref import rpr_case
    R1 = 1 << 14
    specs = [dict(ratio=(int(R1 * 1.3), int(R1 * 0.8)), size=(520, 168), win=(16, 6)), None]
    for seed, idx in ((671, 2),):
        d, refs = rpr_case(400, 208, 6, idx, seed, specs, win=(8, 4), colloc=(0, 1), tools=ALL | abi.TOOL_WP | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE,
                           num_slices=3, p_intra=0.15, p_affine=0.3, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.2)
        synth.vary_slices(d, seed)
        assert sum(d.rpr.ref[l][i].scaled for l in range(2) for i in range(d.hdr.num_ref[l])) in (1, 2, 3)
        fl = refdrv.ROTATE_REF_LISTS
        want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
        assert all(np.array_equal(a, b) for a, b in zip(refdrv.oracle_reconstruct(d, refs), want))      # (the rotation is immaterial to the picture)
        e = refdrv.extract(d, refs, flags=fl)
        x = refdrv.desc_from_extract(e)
        assert x.rpr is not None
        got = refdrv.oracle_reconstruct(x, refs)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
