"""CPU: the plain-C oracle against the real reference classes (oracle/_ref) on fresh seeds.  Skipped where oracle/_ref is
not built (it needs /root/reference at build time); the committed golden fixtures cover that case."""
import numpy as np
import pytest

import refdrv
from vvdec_amd import abi, synth, stream

pytestmark = pytest.mark.skipif(not refdrv.available(), reason="oracle/_ref not built")
ALL = abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF
STAGES = [refdrv.STOP_AFTER_RECO, refdrv.STOP_AFTER_DBK, refdrv.STOP_AFTER_SAO, 0]


def _case(W, H, l2, idx, seed, tools=ALL, **kw):
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, log2_ctu=l2, **kw)
    refs = {}
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc))
    return d, refs


@pytest.mark.parametrize("W,H,l2,idx,seed,kw", [
    (256, 128, 7, 0, 101, {}),
    (384, 256, 7, 2, 102, dict(p_intra=0.25)),
    (200, 136, 6, 3, 103, dict(p_intra=0.1)),
    (320, 192, 5, 1, 104, dict(p_intra=0.0)),
    (264, 200, 7, 4, 105, dict(p_intra=0.5)),
    (384, 256, 7, 2, 106, dict(p_intra=0.1, p_affine=0.5)),
    (200, 136, 6, 1, 107, dict(p_intra=0.0, p_affine=0.4, mv_sigma=2.0)),
    (384, 256, 7, 3, 108, dict(p_intra=0.1, p_geo=0.5)),
    (384, 256, 7, 2, 109, dict(p_intra=0.25, p_ciip=0.5, p_coded=0.6)),
    (384, 256, 7, 3, 110, dict(p_intra=0.1, p_affine=0.3, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.3)),
    (256, 128, 6, 2, 111, dict(p_intra=0.3, p_jccr=0.7, p_coded_chroma=0.7)),
    (256, 128, 6, 0, 112, dict(p_jccr=0.7, p_coded_chroma=0.7)),
    (256, 128, 7, 0, 113, dict(p_cclm=0.5)),
    (200, 136, 5, 2, 114, dict(p_cclm=0.6, p_intra=0.5)),
    (256, 128, 7, 0, 115, dict(p_mip=0.5, p_cclm=0.2, p_lfnst=0.4)),
    (200, 136, 6, 2, 116, dict(p_mip=0.6, p_intra=0.5, p_split_scale=1.5)),
    (256, 128, 7, 2, 117, dict(p_sbt=0.5, p_coded_chroma=0.5, p_jccr=0.2)),
    (200, 136, 5, 3, 118, dict(p_sbt=0.7, p_intra=0.1, p_affine=0.2, p_geo=0.1)),
    (256, 128, 7, 0, 119, dict(p_isp=0.6, p_lfnst=0.4, p_coded=0.7)),
    (200, 136, 5, 2, 120, dict(p_isp=0.7, p_intra=0.5, p_cclm=0.3)),
    (384, 256, 6, 0, 121, dict(p_isp=0.5, p_split_scale=0.5, p_cclm=0.3, p_lfnst=0.4, p_jccr=0.3)),
    (256, 128, 7, 0, 124, dict(dual_tree=1.0, p_cclm=0.4, p_lfnst=0.5, p_isp=0.3, p_mip=0.2, p_coded_chroma=0.6)),
    (200, 136, 5, 0, 125, dict(dual_tree=1.0, p_cclm=0.4, p_jccr=0.3, p_coded_chroma=0.6)),
    (384, 256, 6, 0, 126, dict(dual_tree=1.0, p_split_scale=0.6, p_lfnst=0.4, p_bdpcm=0.2)),
    (256, 128, 6, 0, 129, dict(dual_tree=2.0, p_cclm=0.4, p_lfnst=0.5, p_isp=0.3, p_mip=0.3, p_coded_chroma=0.6, p_split_scale=1.5)),
    (256, 128, 6, 0, 131, dict(dual_tree=3.0, p_isp=0.7, p_split_scale=1.8, p_coded=0.8, p_lfnst=0.3)),
    (200, 136, 5, 0, 132, dict(dual_tree=3.0, p_isp=0.6, p_split_scale=2.0, p_coded=0.8)),
])
def test_oracle_equals_reference_every_stage(built, W, H, l2, idx, seed, kw):
    d, refs = _case(W, H, l2, idx, seed, **kw)
    nd = getattr(d, "num_dmvr", 0)
    for fl in STAGES:
        r = refdrv.reconstruct(d, refs, flags=fl, want_dmvr=nd)
        want = r["planes"]
        got = refdrv.oracle_reconstruct(d, refs, flags=fl)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "flags %d comp %d: %d differ" % (fl, c, int((got[c] != want[c]).sum()))
        if nd:
            assert np.array_equal(refdrv.oracle_dmvr(nd), r["dmvr"][:nd]), "DMVR delta MVs differ"


@pytest.mark.parametrize("idx,seed,cs", [(0, 201, 0), (2, 202, 0), (3, 203, 1), (0, 204, 1), (2, 205, 1)])
def test_oracle_equals_reference_lmcs(built, idx, seed, cs):
    d, refs = _case(256, 192, 7, idx, seed, tools=ALL | abi.TOOL_LMCS | (abi.TOOL_LMCS_CSCALE if cs else 0), p_intra=0.3, p_cclm=0.2, p_mip=0.2, p_ciip=0.1, p_affine=0.1, p_geo=0.1, p_coded_chroma=0.5)
    for fl in STAGES:
        want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
        got = refdrv.oracle_reconstruct(d, refs, flags=fl)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "flags %d comp %d: %d differ" % (fl, c, int((got[c] != want[c]).sum()))


@pytest.mark.parametrize("slice_type,seed", [(0, 211), (1, 212), (0, 213)])
def test_oracle_equals_reference_weighted_prediction(built, slice_type, seed):
    """explicit weighted prediction of B (pps_weighted_bipred_flag) and P (pps_weighted_pred_flag) pictures: plain, affine,
    SbTMVP and CIIP predictions weighted, BCW / GPM CUs untouched, BDOF / DMVR only between references with default weights"""
    W, H = 256, 128
    p = synth.default_params(width=W, height=H, seed=seed, tool_flags=ALL | abi.TOOL_WP, slice_type=slice_type, log2_ctu=6,
                             p_affine=0.2, p_sbtmvp=0.15, p_ciip=0.1, p_geo=0.1, p_bcw=0.2, p_intra=0.1)
    p.poc, p.out_slot = 4, 0
    synth.set_refs(p, [(1, 0), (2, 8)], [] if slice_type == 1 else [(2, 8), (1, 0)])
    d = synth.generate(p)
    refs = {1: synth.natural_picture(W, H, seed + 1), 2: synth.natural_picture(W, H, seed + 2)}
    for fl in (refdrv.STOP_AFTER_RECO, 0):
        want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
        got = refdrv.oracle_reconstruct(d, refs, flags=fl)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "flags %d comp %d: %d differ" % (fl, c, int((got[c] != want[c]).sum()))
    assert d.wp is not None and any(d.wp.e[0][i][0].present for i in range(2))


@pytest.mark.parametrize("l2,idx,seed,extra,kw", [
    (7, 0, 221, 0, dict(p_coded=0.8, p_coded_chroma=0.6)),
    (7, 2, 222, abi.TOOL_SCALING_LIST_NO_LFNST, dict(p_intra=0.3, p_coded=0.8, p_coded_chroma=0.6, p_lfnst=0.5, p_sbt=0.3)),
    (6, 3, 223, 0, dict(p_intra=0.2, p_coded=0.8, p_coded_chroma=0.6, p_jccr=0.3, p_small_corner=0.2, p_split_scale=0.5)),
    (5, 1, 224, 0, dict(p_intra=0.1, p_coded=0.9, p_mts=0.5, p_ts=0.2, p_lfnst=0.5)),
])
def test_oracle_equals_reference_scaling_lists(built, l2, idx, seed, extra, kw):
    """explicit scaling lists: 2x2 .. 64x64 matrices incl. rectangular blocks, DC entries, transform-skip and (optionally) LFNST
    blocks left flat"""
    d, refs = _case(256, 192, l2, idx, seed, tools=ALL | abi.TOOL_SCALING_LIST | extra, **kw)
    want = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_RECO)["planes"]
    got = refdrv.oracle_reconstruct(d, refs, flags=refdrv.STOP_AFTER_RECO)
    for c in range(3):
        assert np.array_equal(got[c], want[c]), "comp %d: %d differ" % (c, int((got[c] != want[c]).sum()))
    d.hdr.tool_flags &= ~abi.TOOL_SCALING_LIST
    flat = refdrv.oracle_reconstruct(d, refs, flags=refdrv.STOP_AFTER_RECO)
    assert any(not np.array_equal(flat[c], got[c]) for c in range(3)), "the lists changed nothing"


@pytest.mark.parametrize("l2,idx,seed", [(7, 0, 231), (6, 2, 232), (5, 0, 233)])
def test_oracle_equals_reference_cclm_collocated(built, l2, idx, seed):
    """CCLM with sps_chroma_vertical_collocated_flag = 1 (5-tap cross down-sampling)"""
    d, refs = _case(256, 192, l2, idx, seed, tools=ALL | abi.TOOL_CCLM_COLLOC, p_cclm=0.6, p_intra=0.5, p_isp=0.2)
    want = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_RECO)["planes"]
    got = refdrv.oracle_reconstruct(d, refs, flags=refdrv.STOP_AFTER_RECO)
    for c in range(3):
        assert np.array_equal(got[c], want[c]), "comp %d: %d differ" % (c, int((got[c] != want[c]).sum()))


@pytest.mark.parametrize("l2,idx,seed,kw", [(5, 0, 241, dict(dual_tree=2.0)), (6, 2, 242, dict(p_intra=0.5)), (7, 0, 243, dict())])
def test_oracle_equals_reference_implicit_mts(built, l2, idx, seed, kw):
    """MTS without explicit intra MTS: intra luma blocks take DST-7 along every dimension of 4..16 samples (host-resolved tr_type)"""
    d, refs = _case(256, 128, l2, idx, seed, tools=ALL | abi.TOOL_IMPLICIT_MTS, p_mip=0.2, p_lfnst=0.3, p_isp=0.2, p_coded=0.8, **kw)
    want = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_RECO)["planes"]
    got = refdrv.oracle_reconstruct(d, refs, flags=refdrv.STOP_AFTER_RECO)
    for c in range(3):
        assert np.array_equal(got[c], want[c]), "comp %d: %d differ" % (c, int((got[c] != want[c]).sum()))


@pytest.mark.parametrize("W,H,l2,idx,seed,tools,kw", [
    (256, 128, 6, 2, 251, ALL, dict(p_intra=0.3, p_split_scale=1.8)),
    (256, 128, 7, 0, 252, ALL, dict(p_split_scale=1.8, p_cclm=0.3, p_isp=0.3, p_mip=0.2, p_lfnst=0.3)),
    (200, 136, 5, 3, 253, ALL | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_intra=0.25, p_split_scale=2.0, p_sbt=0.2, p_cclm=0.3, p_jccr=0.2, p_coded_chroma=0.5)),
    (384, 256, 6, 1, 254, ALL | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_WP, dict(p_intra=0.2, p_split_scale=1.7, p_isp=0.2, p_coded_chroma=0.5)),
    (200, 136, 5, 3, 255, ALL | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_intra=0.2, p_split_scale=2.0, p_ciip=0.6, p_coded=0.6, p_coded_chroma=0.5)),     # 4xN / Nx4 CIIP
])
def test_oracle_equals_reference_small_cus(built, W, H, l2, idx, seed, tools, kw):
    """minimum CU size 4: 4xN inter CUs (2xN chroma blocks), Nx4 intra CUs (Nx2 chroma blocks), and the local dual tree of intra-only
    sub-trees (luma-tree CUs down to 4x4 + one chroma-tree CU per node) in I and B pictures; edge parameters included"""
    d, refs = _case(W, H, l2, idx, seed, tools=tools, min_cu_log2=2, **kw)
    assert int((d.cu["tree"] == abi.TREE_CHROMA).sum()) > 0 and int(((d.cu["w"] == 4) | (d.cu["h"] == 4)).sum()) > 0
    for fl in (refdrv.STOP_AFTER_RECO, 0, refdrv.DERIVE_LFP):
        want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
        got = refdrv.oracle_reconstruct(d, refs, flags=fl & ~refdrv.DERIVE_LFP)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "flags %d comp %d: %d differ" % (fl, c, int((got[c] != want[c]).sum()))


@pytest.mark.parametrize("W,H,l2,idx,seed,lmcs,kw", [
    (256, 128, 7, 0, 321, 0, dict(p_ibc=0.4, p_coded=0.5)),
    (384, 256, 6, 0, 302, 1, dict(p_ibc=0.5, p_split_scale=1.4, p_cclm=0.3, p_jccr=0.3, p_coded_chroma=0.5)),
    (200, 136, 5, 0, 303, 0, dict(p_ibc=0.7, p_coded=0.3)),
    (384, 256, 7, 2, 304, 1, dict(p_ibc=0.6, p_intra=0.5, p_ciip=0.1, p_affine=0.1)),
    (256, 128, 6, 0, 305, 1, dict(p_ibc=0.5, dual_tree=2.0, p_split_scale=1.5, p_cclm=0.3)),
    (264, 200, 7, 3, 306, 0, dict(p_ibc=0.6, p_intra=0.4, min_cu_log2=2, p_split_scale=1.6)),
])
def test_oracle_equals_reference_intra_block_copy(built, W, H, l2, idx, seed, lmcs, kw):
    """IBC CUs (InterPrediction::xIntraBlockCopy): block vectors into the current and the left CTUs of the row, luma + chroma at the
    halved vector, luma-only CUs of dual / local dual trees, with and without LMCS chroma residual scaling; every stage, and the
    generator's edge parameters against the reference's own derivation (IBC boundary-strength rules)"""
    tools = ALL | abi.TOOL_IBC | ((abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE) if lmcs else 0)
    d, refs = _case(W, H, l2, idx, seed, tools=tools, **kw)
    assert int((d.cu["pred_mode"] == abi.PRED_IBC).sum()) > 10
    for fl in STAGES + [refdrv.STOP_AFTER_DBK | refdrv.DERIVE_LFP]:
        want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
        got = refdrv.oracle_reconstruct(d, refs, flags=fl & ~refdrv.DERIVE_LFP)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "flags %d comp %d: %d differ" % (fl, c, int((got[c] != want[c]).sum()))


@pytest.mark.parametrize("bit_depth,chroma_format", [(8, 1), (10, 0), (8, 0), (9, 1)], ids=["8bit_420", "10bit_400", "8bit_400", "9bit_420"])
def test_oracle_equals_reference_other_sample_formats(built, bit_depth, chroma_format):
    """8-bit samples and 4:0:0 pictures (the other formats of the Main 10 profile), intra and inter tools, with and without LMCS"""
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    for (W, H, l2, idx, seed, kw) in ((256, 128, 7, 0, 401, dict(p_cclm=0.3, p_mip=0.2, p_isp=0.2)),
                                      (384, 256, 6, 2, 402, dict(p_intra=0.25, p_affine=0.2, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.1, p_sbt=0.1, p_bcw=0.2))):
        for lm in (0, abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE):
            pl = plans[idx]
            d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=ALL | lm, log2_ctu=l2, bit_depth=bit_depth, chroma_format=chroma_format, **kw)
            refs = {}
            for lst in pl.ref_slots:
                for (slot, poc) in lst:
                    refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc, bit_depth=bit_depth))
            for fl in STAGES:
                want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
                got = refdrv.oracle_reconstruct(d, refs, flags=fl)
                for c in range(3 if chroma_format else 1):
                    assert np.array_equal(got[c], want[c]), "flags %d comp %d: %d differ" % (fl, c, int((got[c] != want[c]).sum()))


def test_reference_simd_equals_scalar(built):
    """the reference's own differential check (its unit test compares scalar vs SIMD kernels): same bytes at frame level"""
    d, refs = _case(256, 192, 7, 2, 106, p_intra=0.2)
    a = refdrv.reconstruct(d, refs, flags=0)["planes"]
    b = refdrv.reconstruct(d, refs, flags=refdrv.SIMD)["planes"]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_edge_parameters_match_reference_derivation(built):
    """deblocking with the edge parameters the reference derives itself (LoopFilter::calcFilterStrengthsCTU) == with the
    job's table (the host glue's restatement of that derivation)"""
    d, refs = _case(256, 128, 7, 2, 107, p_intra=0.2)
    a = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_DBK)["planes"]
    b = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_DBK | refdrv.DERIVE_LFP)["planes"]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    for args, kw in (((256, 128, 7, 0, 122), dict(p_isp=0.7, p_split_scale=1.6)), ((384, 256, 6, 2, 123), dict(p_isp=0.5, p_intra=0.4, p_cclm=0.3)),
                     ((384, 256, 7, 0, 127), dict(dual_tree=1.0, p_isp=0.2, p_bdpcm=0.3, p_coded_chroma=0.6)),
                     ((384, 256, 7, 1, 134), dict(p_intra=0.0, p_affine=0.5, p_sbtmvp=0.2, p_split_scale=0.5)),      # sub-block edges of affine / SbTMVP CUs, also across TU edges of 128x128 CUs
                     ((200, 136, 6, 3, 135), dict(p_intra=0.1, p_affine=0.3, p_sbtmvp=0.4, mv_sigma=3.0, p_geo=0.1, p_ciip=0.1)),
                     ((256, 128, 6, 0, 130), dict(dual_tree=2.0, p_split_scale=1.5, p_isp=0.2)), ((256, 128, 6, 0, 133), dict(dual_tree=3.0, p_split_scale=1.8, p_isp=0.7)), ((256, 128, 6, 2, 128), dict(p_bdpcm=0.5, p_intra=0.6))):
        d, refs = _case(*args, **kw)                                                       # ISP: partition edges, unsplit chroma
        a = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_DBK)["planes"]
        b = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_DBK | refdrv.DERIVE_LFP)["planes"]
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    d, refs = _case(256, 128, 6, 3, 119, p_intra=0.1, p_sbt=0.6, p_coded_chroma=0.4)     # sub-block transform: TU edges inside inter CUs
    a = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_DBK)["planes"]
    b = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_DBK | refdrv.DERIVE_LFP)["planes"]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_oracle_equals_reference_random_sweep(built):
    """a fixed-seed slice of tools/fuzz_oracle_vs_ref.py: random tool combinations, sizes, CTU sizes, sample formats and stages"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_oracle_vs_ref", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_oracle_vs_ref.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    n, bad = fz.sweep(2026, cases=250)
    assert n == 250 and bad == 0


@pytest.mark.parametrize("W,H,l2,idx,seed,bd,kw", [(256, 128, 7, 0, 251, 10, {}), (384, 256, 6, 2, 252, 10, dict(p_intra=0.25, p_affine=0.2)), (200, 136, 5, 3, 253, 8, dict(p_intra=0.1))])
def test_oracle_equals_reference_ladf(built, W, H, l2, idx, seed, bd, kw):
    """luma-adaptive deblocking (sps_ladf_enabled_flag): the QP of a luma edge segment is shifted by an offset chosen from the local luma level
    (LoopFilter::deriveLADFShift, LoopFilter.cpp:1363)"""
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=ALL | abi.TOOL_LADF, log2_ctu=l2, bit_depth=bd, **kw)
    assert 2 <= d.hdr.ladf_num_intervals <= 5
    refs = {}
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc, bit_depth=bd))
    for fl in (refdrv.STOP_AFTER_DBK, 0):
        want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
        got = refdrv.oracle_reconstruct(d, refs, flags=fl)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "flags %d comp %d: %d differ" % (fl, c, int((got[c] != want[c]).sum()))
    # the offsets matter: without them the deblocked picture is another one
    final = want
    d.hdr.ladf_num_intervals = 0
    assert not np.array_equal(refdrv.oracle_reconstruct(d, refs, flags=0)[0], final[0])


SLICE_TILE_CASES = [
    # W, H, l2, idx, seed, extra tool flags, generator parameters
    (512, 384, 6, 0, 261, abi.TOOL_NO_LF_ACROSS_SLICES, dict(num_slices=3, p_cclm=0.3, p_mip=0.2)),
    (512, 384, 6, 0, 262, 0, dict(num_slices=3, p_cclm=0.3)),                                                    # slices, loop filters cross them
    (512, 384, 6, 2, 263, abi.TOOL_NO_LF_ACROSS_TILES, dict(tile_cols=2, tile_rows=2, p_intra=0.3, p_ciip=0.2, p_cclm=0.3)),
    (512, 384, 6, 2, 264, abi.TOOL_NO_LF_ACROSS_SLICES | abi.TOOL_NO_LF_ACROSS_TILES, dict(num_slices=3, tile_cols=2, tile_rows=2, p_intra=0.3, p_affine=0.2)),
    (512, 384, 6, 3, 265, abi.TOOL_NO_LF_ACROSS_SLICES, dict(num_slices=3, tile_cols=2, tile_rows=2, p_intra=0.2)),   # raster slices over tiles: the ALF corner padding
    (640, 256, 5, 0, 266, abi.TOOL_NO_LF_ACROSS_SLICES | abi.TOOL_NO_LF_ACROSS_TILES | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(num_slices=4, tile_cols=3, tile_rows=2, p_cclm=0.4, p_coded_chroma=0.6)),
    (384, 256, 7, 2, 267, abi.TOOL_NO_LF_ACROSS_TILES | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(tile_cols=3, tile_rows=1, p_intra=0.3, p_coded_chroma=0.6, p_cclm=0.3)),
    (512, 384, 6, 0, 268, abi.TOOL_NO_LF_ACROSS_SLICES | abi.TOOL_NO_LF_ACROSS_TILES, dict(num_slices=2, tile_cols=2, tile_rows=2, dual_tree=1.0, p_cclm=0.4)),
]


@pytest.mark.parametrize("W,H,l2,idx,seed,extra,kw", SLICE_TILE_CASES)
def test_oracle_equals_reference_slices_and_tiles(built, W, H, l2, idx, seed, extra, kw):
    """pictures of several slices and tiles: no intra / CCLM / chroma-scaling neighbourhood across their boundaries (getCURestricted); SAO and ALF stop
    there when the loop filters may not cross (SAO availability flags, ALF border padding incl. the raster-slice corners); deblocking edges switched
    off by the host-derived table.  The reference runs with real Slice objects and a real tile grid (oracle/ref_harness.cpp)."""
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=ALL | extra, log2_ctu=l2, **kw)
    assert (d.ctu_slice is not None) == (kw.get("num_slices", 1) > 1) and (d.ctu_tile is not None) == (kw.get("tile_cols", 1) * kw.get("tile_rows", 1) > 1)
    refs = {}
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc))
    for fl in STAGES:
        want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
        got = refdrv.oracle_reconstruct(d, refs, flags=fl)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "flags %d comp %d: %d differ" % (fl, c, int((got[c] != want[c]).sum()))
    # the boundaries matter: the same records without the maps give another picture
    final = want
    d.ctu_slice = d.ctu_tile = None
    d.hdr.tool_flags &= ~(abi.TOOL_NO_LF_ACROSS_SLICES | abi.TOOL_NO_LF_ACROSS_TILES)
    other = refdrv.oracle_reconstruct(d, refs, flags=0)
    assert any(not np.array_equal(a, b) for a, b in zip(other, final))


SLICE_HEADER_CASES = [
    # W, H, l2, idx, seed, extra tool flags, generator parameters
    (512, 384, 6, 0, 281, abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_SCALING_LIST, dict(num_slices=3, p_cclm=0.3, p_coded=0.8, p_coded_chroma=0.6)),
    (512, 384, 6, 2, 282, abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_WP | abi.TOOL_NO_LF_ACROSS_SLICES, dict(num_slices=4, p_intra=0.3, p_ciip=0.2, p_coded=0.8, p_coded_chroma=0.6, p_affine=0.2)),
    (640, 256, 5, 3, 283, abi.TOOL_SCALING_LIST | abi.TOOL_WP | abi.TOOL_NO_LF_ACROSS_TILES, dict(num_slices=4, tile_cols=3, tile_rows=2, p_intra=0.2, p_coded=0.8, p_sbtmvp=0.2)),
    (512, 384, 6, 1, 284, abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_SCALING_LIST, dict(num_slices=5, p_intra=0.2, p_coded=0.7, p_coded_chroma=0.6, p_geo=0.2)),
    # I slices in a B picture (slices 1 and 2 hold intra CUs only and carry slice type I: no reference lists, no weights)
    (512, 384, 6, 2, 285, abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_WP | abi.TOOL_NO_LF_ACROSS_SLICES, dict(num_slices=4, intra_slices=0b0110, p_intra=0.2, p_cclm=0.3, p_coded=0.8, p_coded_chroma=0.6)),
]


@pytest.mark.parametrize("W,H,l2,idx,seed,extra,kw", SLICE_HEADER_CASES)
def test_oracle_equals_reference_slice_headers(built, W, H, l2, idx, seed, extra, kw):
    """slices whose headers differ (vvr_slice_header): dependent quantisation, LMCS (+ chroma residual scaling), explicit scaling lists on or
    off per slice, deblocking offsets, the APSs the ALF takes its filters from and the prediction weights - every stage takes the values of
    the slice the CU / CTU lies in (ctuData.slice).  The reference runs with Slice objects that carry exactly these headers."""
    d, refs = _case(W, H, l2, idx, seed, tools=ALL | extra, **kw)
    synth.vary_slices(d, seed, intra_slices=kw.get("intra_slices", 0))
    n = len(d.slices)
    if kw.get("intra_slices"):
        ctus_x = (W + (1 << l2) - 1) >> l2
        cu_slice = d.ctu_slice[(d.cu["y"].astype(int) >> l2) * ctus_x + (d.cu["x"].astype(int) >> l2)]
        is_i = d.slices["slice_type"][cu_slice] == abi.SLICE_I
        assert is_i.any() and (~is_i).any() and (d.cu["pred_mode"][is_i] != abi.PRED_INTER).all() and (d.cu["pred_mode"][~is_i] == abi.PRED_INTER).any()
    assert n == kw["num_slices"] and len(set(int(f) for f in d.slices["tool_flags"])) > 1
    for fl in STAGES:
        want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
        got = refdrv.oracle_reconstruct(d, refs, flags=fl)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "flags %d comp %d: %d differ" % (fl, c, int((got[c] != want[c]).sum()))
    # every field matters: the picture changes when the slices lose it
    final = want

    def changed(edit):
        keep = d.slices.copy(), d.alf_sets, d.wp_sets
        edit()
        out = refdrv.oracle_reconstruct(d, refs, flags=0)
        d.slices, d.alf_sets, d.wp_sets = keep
        return any(not np.array_equal(a, b) for a, b in zip(out, final))

    def all_like_first(name):
        def f():
            d.slices[name] = d.slices[name][0]
        return f
    for name in ("tool_flags", "deblock_beta_offset_div2", "deblock_tc_offset_div2", "alf_set") + (("wp_set",) if d.wp is not None else ()):
        assert changed(all_like_first(name)), name


VB_CASES = [
    # W, H, l2, idx, seed, virtual_boundaries (bits 0-1 vertical, 2-3 horizontal, 16: first on a CTU boundary), extra tool flags, generator parameters
    (512, 384, 6, 0, 271, 1 | (1 << 2), 0, dict(p_cclm=0.2)),
    (512, 384, 6, 2, 272, 3 | (3 << 2), 0, dict(p_intra=0.2, p_affine=0.3, p_sbtmvp=0.2)),                        # three each; sub-block edges on a boundary
    (512, 384, 7, 3, 273, 2 | (2 << 2) | 16, 0, dict(p_intra=0.2)),                                               # one of each on a CTU boundary
    (384, 256, 5, 0, 274, 3 | (2 << 2) | 16, abi.TOOL_NO_LF_ACROSS_SLICES, dict(num_slices=3, dual_tree=1.0)),    # together with slices the filters do not cross
    (640, 256, 6, 2, 275, 2 | (1 << 2), abi.TOOL_NO_LF_ACROSS_SLICES | abi.TOOL_NO_LF_ACROSS_TILES | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(num_slices=3, tile_cols=2, tile_rows=2, p_intra=0.3)),
]


@pytest.mark.parametrize("W,H,l2,idx,seed,vb,extra,kw", VB_CASES)
def test_oracle_equals_reference_virtual_boundaries(built, W, H, l2, idx, seed, vb, extra, kw):
    """virtual boundaries of the picture header: edges on them are not deblocked (host table; the reference derives the same from its own objects), SAO
    skips the sample columns / rows next to them for the classes that look across (isProcessDisabled), ALF filters every part of a CTU they cut out
    with a border of its own (filterCTU)"""
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=ALL | extra, log2_ctu=l2, virtual_boundaries=vb, **kw)
    assert d.hdr.num_ver_vb == (vb & 3) and d.hdr.num_hor_vb == ((vb >> 2) & 3)
    refs = {}
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc))
    for fl in STAGES:
        want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
        got = refdrv.oracle_reconstruct(d, refs, flags=fl)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "flags %d comp %d: %d differ" % (fl, c, int((got[c] != want[c]).sum()))
    # the job's edge table == what the reference derives from its own objects with the boundaries in its picture header
    a = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_DBK)["planes"]
    b = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_DBK | refdrv.DERIVE_LFP)["planes"]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # the boundaries matter
    final = want
    d.hdr.num_ver_vb = d.hdr.num_hor_vb = 0
    other = refdrv.oracle_reconstruct(d, refs, flags=0)
    assert any(not np.array_equal(x, y) for x, y in zip(other, final))


@pytest.mark.parametrize("W,H,l2,idx,seed,kw", [(384, 256, 7, 2, 281, dict(p_affine=0.6, p_intra=0.05)), (416, 240, 6, 1, 282, dict(p_affine=0.5, mv_sigma=30.0, p_intra=0.1)),
                                                  (256, 128, 5, 3, 283, dict(p_affine=0.7, p_intra=0.0, mv_sigma=60.0))])
def test_affine_motion_spanned_by_the_reference(built, W, H, l2, idx, seed, kw):
    """the sub-block MVs the descriptions carry for affine CUs are what the reference's own PU::setAllAffineMv spans from the control-point MVs (incl.
    the single fallback vector when the sub-block vectors spread too far): the reference fills the motion of the affine CUs itself and ends with the
    same motion field and the same picture.  Pins the restatement the device-side spanning (VVR_TOOL_AFFINE_MV_ON_DEVICE) is tested against."""
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=ALL | abi.TOOL_PROF, log2_ctu=l2, **kw)
    aff = (d.cu["flags"] & abi.CU_AFFINE) != 0
    assert aff.sum() >= 3
    refs = {}
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc))
    a, ma = refdrv.reconstruct_with_motion(d, refs, flags=0)
    b, mb = refdrv.reconstruct_with_motion(d, refs, flags=refdrv.SPAN_AFFINE)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert ma.tobytes() == mb.tobytes()


WRAP_CASES = [
    # W, H, l2, idx, seed, wrap offset, extra tools, generator parameters
    (512, 256, 6, 2, 291, 512, 0, dict(p_intra=0.05, mv_sigma=12.0)),                                                       # the period is the picture width (360-degree video)
    (512, 256, 6, 1, 292, 480, abi.TOOL_BDOF | abi.TOOL_DMVR, dict(p_intra=0.05, p_bi=0.9, mv_sigma=4.0)),                   # BDOF / DMVR sub-blocks
    (640, 256, 7, 3, 293, 640, abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF, dict(p_intra=0.1, p_affine=0.4, p_sbtmvp=0.2, p_geo=0.2, p_ciip=0.1)),
    (384, 256, 5, 2, 294, 320, abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_LMCS, dict(p_intra=0.1, p_affine=0.3, p_bi=0.8, mv_sigma=20.0)),
    (416, 240, 6, 3, 295, 416, abi.TOOL_WP, dict(p_intra=0.1, p_bi=0.7)),
    # vectors several wrap periods out (a parsed stream with AMVR carries them; mv_window lets the generator keep them): every clip path is taken with a move by a
    # period AND a clamp - the case in which the DMVR start vectors must be clipped against the CU, not the sub-block (round 4, tests/bitstreams/wraparound_*)
    (384, 256, 6, 2, 296, 368, abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF, dict(p_intra=0.05, p_bi=0.9, mv_sigma=1500.0, mv_window=2000)),
    (384, 256, 6, 3, 297, 368, abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_LMCS, dict(p_intra=0.1, p_bi=0.8, p_affine=0.2, p_geo=0.2, p_ciip=0.1, mv_sigma=3000.0, mv_window=4000)),
    # SbTMVP with such vectors: the pieces xSubPuMC joins decide the wrap clip (wide, tall and square CUs; cut runs)
    (384, 256, 6, 2, 298, 368, 0, dict(p_intra=0.05, p_sbtmvp=0.5, p_bi=0.5, mv_sigma=1500.0, mv_window=2000)),
    (512, 256, 7, 1, 299, 512, abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF, dict(p_intra=0.05, p_sbtmvp=0.4, p_affine=0.2, p_bi=0.7, mv_sigma=2500.0, mv_window=4000)),
]


@pytest.mark.parametrize("W,H,l2,idx,seed,off,extra,kw", WRAP_CASES)
def test_oracle_equals_reference_wrap_around(built, W, H, l2, idx, seed, off, extra, kw):
    """horizontal reference wrap-around (pps_ref_wraparound_enabled_flag): every MC path reads the wrap copy of the reference pictures or, after an MV
    was moved by one period, the ordinary one (wrapClipMv; regular, BDOF, DMVR prefetch / start / final, affine per sub-block, GPM, SbTMVP, CIIP)"""
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=ALL | extra, log2_ctu=l2, wrap_offset=off, **kw)
    assert d.hdr.wrap_offset == off
    refs = {}
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc))
    nd = d.num_dmvr
    r = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_RECO, want_dmvr=nd)
    got = refdrv.oracle_reconstruct(d, refs, flags=refdrv.STOP_AFTER_RECO)
    for c in range(3):
        assert np.array_equal(got[c], r["planes"][c]), "comp %d: %d differ" % (c, int((got[c] != r["planes"][c]).sum()))
    if nd:
        assert np.array_equal(refdrv.oracle_dmvr(nd), r["dmvr"][:nd])
    want = refdrv.reconstruct(d, refs, flags=0)["planes"]
    got = refdrv.oracle_reconstruct(d, refs, flags=0)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    # the wrap-around matters: the same records without it give another picture
    d.hdr.wrap_offset = 0
    other = refdrv.oracle_reconstruct(d, refs, flags=0)
    assert any(not np.array_equal(a, b) for a, b in zip(other, want))


SUBPIC_CASES = [
    # W, H, l2, idx, seed, subpics (bit 0 on; bits 1-2 treated as a picture: 0 none 1 all 2 some; bits 3-4 loop filters across: 0 all 1 none 2 some), extra tools, generator parameters
    (512, 384, 6, 2, 301, 1 | (1 << 1) | (1 << 3), 0, dict(tile_cols=2, tile_rows=2, p_intra=0.1, mv_sigma=24.0)),
    (512, 384, 6, 1, 302, 1 | (2 << 1) | (2 << 3), abi.TOOL_BDOF | abi.TOOL_DMVR, dict(tile_cols=2, tile_rows=2, p_intra=0.1, p_bi=0.9, mv_sigma=16.0)),
    (640, 256, 5, 3, 303, 1 | (1 << 1) | (0 << 3), abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF, dict(tile_cols=3, tile_rows=2, p_intra=0.1, p_affine=0.4, p_sbtmvp=0.2, p_geo=0.2, p_ciip=0.1, mv_sigma=30.0)),
    (384, 256, 7, 2, 304, 1 | (2 << 1) | (1 << 3), abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_NO_LF_ACROSS_SLICES, dict(tile_cols=3, tile_rows=1, p_intra=0.3, p_cclm=0.3, mv_sigma=40.0)),
    (512, 384, 6, 0, 305, 1 | (1 << 1) | (1 << 3), 0, dict(tile_cols=2, tile_rows=3, dual_tree=1.0)),                    # I picture: only the loop filters know about sub-pictures
]


@pytest.mark.parametrize("W,H,l2,idx,seed,sp,extra,kw", SUBPIC_CASES)
def test_oracle_equals_reference_subpictures(built, W, H, l2, idx, seed, sp, extra, kw):
    """sub-pictures (one per tile, one slice each): CUs of a sub-picture that is treated as a picture are predicted from that sub-picture of the reference
    pictures only (clipMvInSubpic + the sub-picture copies with their own border), SAO and ALF of a sub-picture whose flag says so do not look into other
    sub-pictures, deblocking across a boundary needs the flag of both sides (host table, checked against the reference's derivation)"""
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=ALL | extra, log2_ctu=l2, subpics=sp, **kw)
    assert d.subpics is not None and len(d.subpics) == kw["tile_cols"] * kw["tile_rows"]
    refs = {}
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc))
    for fl in STAGES:
        want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
        got = refdrv.oracle_reconstruct(d, refs, flags=fl)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "flags %d comp %d: %d differ" % (fl, c, int((got[c] != want[c]).sum()))
    a = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_DBK)["planes"]
    b = refdrv.reconstruct(d, refs, flags=refdrv.STOP_AFTER_DBK | refdrv.DERIVE_LFP)["planes"]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # the sub-pictures matter
    final = want
    d.subpics = None
    other = refdrv.oracle_reconstruct(d, refs, flags=0)
    assert any(not np.array_equal(x, y) for x, y in zip(other, final))


@pytest.mark.parametrize("bd,cf,W,H", [(10, 1, 136, 72), (8, 1, 136, 72), (10, 0, 136, 72), (10, 1, 520, 264)])
def test_picture_hash_restatement_equals_reference(built, bd, cf, W, H):
    """tests/refdrv.py's restatement of the decoded-picture-hash functions (the checker of the GPU output stage) against the reference's own
    calcMD5 / calcCRC / calcChecksum (PicYuvMD5.cpp), incl. a picture wider and taller than 256 samples (the checksum's mask folds x >> 8, y >> 8)"""
    import ctypes as C
    rng = np.random.default_rng(bd * 100 + cf * 10 + W)
    planes = [rng.integers(0, 1 << bd, (H >> s, W >> s)).astype(np.uint16) for s in ((0, 1, 1) if cf else (0,))]
    L = refdrv.lib()
    ptrs = (C.POINTER(C.c_uint16) * 3)()
    for c, pl in enumerate(planes):
        ptrs[c] = pl.ctypes.data_as(C.POINTER(C.c_uint16))
    for method, length in ((0, 16), (1, 2), (2, 4)):
        ref = (C.c_uint8 * 48)()
        assert L.vvref_picture_hash(ptrs, W, H, cf, bd, method, ref) == length
        assert b"".join(refdrv.picture_hash(planes, bd, method)) == bytes(ref[:length * len(planes)]), "method %d" % method


def rpr_case(W, H, l2, idx, seed, specs, win=(0, 0), colloc=(1, 1), tools=ALL, bit_depth=10, chroma_format=1, **kw):
    """a picture of the plan whose k-th distinct reference picture (in list order) is the scaled picture specs[k] (None: an ordinary one):
    description with its vvr_rpr_params, reference pictures of their own sizes"""
    import ctypes as C
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    slots = []
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            if slot not in slots:
                slots.append(slot)
    by_slot = {s: (specs[k] if k < len(specs) else None) for k, s in enumerate(slots)}
    masks, refs_spec = [0, 0], {}
    for l, lst in enumerate(pl.ref_slots):
        for i, (slot, poc) in enumerate(lst):
            if by_slot[slot]:
                masks[l] |= 1 << i
                refs_spec[(l, i)] = by_slot[slot]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, log2_ctu=l2, bit_depth=bit_depth, chroma_format=chroma_format, scaled_refs=(C.c_uint16 * 2)(*masks), **kw)
    synth.attach_rpr(d, refs_spec, win=win, colloc=colloc)
    refs = {}
    for lst in pl.ref_slots:
        for (slot, poc) in lst:
            sz = (by_slot[slot] or {}).get("size", (W, H))
            refs.setdefault(slot, [p if chroma_format or c == 0 else None for c, p in enumerate(synth.natural_picture(sz[0], sz[1], seed + 100 + poc, bit_depth=bit_depth))][:3 if chroma_format else 1])
    return d, refs


R1 = 1 << 14
RPR_CASES = [
    # W, H, l2, idx, seed, scaled reference pictures, window of the current picture, collocated flags, generator parameters
    # same size, another window: 1.5x (the first low-pass filter set), one reference picture of two
    (384, 256, 7, 2, 301, [dict(ratio=(R1 * 3 // 2, R1 * 3 // 2))], (0, 0), (1, 1), dict(p_intra=0.1)),
    # a reference picture of twice the size: 2x (the second low-pass set), both reference pictures scaled
    (256, 128, 6, 2, 302, [dict(ratio=(2 * R1, 2 * R1), size=(512, 256)), dict(ratio=(2 * R1, 2 * R1), size=(512, 256))], (0, 0), (1, 1), dict(p_intra=0.1)),
    # half the size: up-sampling with the regular filters; affine, GPM, CIIP, SbTMVP, BCW, half-sample AMVR among the CUs
    (384, 256, 7, 3, 303, [dict(ratio=(R1 // 2, R1 // 2), size=(192, 128))], (0, 0), (1, 1), dict(p_intra=0.1, p_affine=0.3, p_geo=0.15, p_ciip=0.15, p_sbtmvp=0.2, p_bcw=0.5, p_imv_hpel=0.3)),
    # ratios that differ by direction, windows with offsets, chroma samples not collocated, weighted prediction; the second picture at 1.3 x 1.8
    (400, 208, 6, 2, 304, [dict(ratio=(int(R1 * 1.3), int(R1 * 0.8)), size=(520, 168), win=(16, 6)), dict(ratio=(int(R1 * 1.3), int(R1 * 1.8)), size=(512, 376), win=(-8, 2))], (8, 4), (0, 0),
     dict(p_intra=0.15, p_affine=0.3, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.2, mv_sigma=6.0, tool_flags_extra=abi.TOOL_WP)),
    # a P picture (key picture of the plan) from a larger picture, 8 bit
    (320, 192, 5, 1, 305, [dict(ratio=(int(R1 * 1.6), int(R1 * 1.25)), size=(512, 240))], (0, 0), (1, 0), dict(p_intra=0.1, p_affine=0.3, bit_depth=8)),
    # 4:0:0
    (256, 128, 6, 2, 306, [dict(ratio=(R1 * 7 // 4 + 1, R1 * 5 // 4 + 1), size=(448, 160))], (4, 2), (1, 1), dict(p_intra=0.1, p_affine=0.3, p_geo=0.1, chroma_format=0)),
]


@pytest.mark.parametrize("W,H,l2,idx,seed,specs,win,colloc,kw", RPR_CASES)
def test_oracle_equals_reference_scaled_reference_pictures(built, W, H, l2, idx, seed, specs, win, colloc, kw):
    """Reference picture resampling (vvr_picture.rpr): predictions from reference pictures of another size / scaling window go through
    InterPrediction::xPredInterBlkRPR in the reference - positions advancing by the scaling ratio, the three filter sets per component,
    the affine variants, no MV clipping, no BDOF / DMVR / PROF for those CUs."""
    kw = dict(kw)
    tools = ALL | kw.pop("tool_flags_extra", 0)
    d, refs = rpr_case(W, H, l2, idx, seed, specs, win=win, colloc=colloc, tools=tools, **kw)
    inter = d.cu[d.cu["pred_mode"] == abi.PRED_INTER]
    assert len(inter) > 10
    for fl in (refdrv.STOP_AFTER_RECO, 0):
        want = refdrv.reconstruct(d, refs, flags=fl)["planes"]
        got = refdrv.oracle_reconstruct(d, refs, flags=fl)
        for c in range(len(want)):
            assert np.array_equal(got[c], want[c]), "flags %d comp %d: %d differ" % (fl, c, int((got[c] != want[c]).sum()))
    # the table matters: without it (every reference picture read as an ordinary one of its own size) the picture differs
    if all("size" not in (s or {}) for s in specs):
        keep, d.rpr = d.rpr, None
        other = refdrv.oracle_reconstruct(d, refs, flags=0)
        d.rpr = keep
        assert any(not np.array_equal(a, b) for a, b in zip(other, want))


def test_random_scaled_reference_pictures_fixed_seed_slice(built):
    """a fixed-seed slice of tools/fuzz_rpr.py (random sizes, ratios from 1/8 to 2 incl. the filter-set thresholds, windows with negative offsets, sample
    formats, large MVs): oracle == reference"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_rpr", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_rpr.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    n = 0
    for it, (d, refs), l2 in fz.cases(3, 8):
        want = refdrv.reconstruct(d, refs)["planes"]
        got = refdrv.oracle_reconstruct(d, refs)
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), "case %d" % it
        n += 1
    assert n >= 5


@pytest.mark.parametrize("threads", [0, 3])
def test_reference_scheduler_equals_the_serial_stages(threads):
    """the CPU baseline of bench.py drives the reference's own scheduler - DecLibRecon's set-up and ctuTask state machine on its ThreadPool, started at
    LF_INIT (oracle/ref_harness.cpp::decompressFromLfInit) -: its pictures equal what the harness gets by calling the stage functions one after the
    other (which is what the oracle is pinned against), for an I picture and a B picture with every tool"""
    W, H = 416, 240
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    ALL = (abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST | abi.TOOL_BDOF | abi.TOOL_DMVR |
           abi.TOOL_PROF | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE)
    for idx, kw in ((0, dict(p_cclm=0.3, p_mip=0.2, p_isp=0.1)), (2, dict(p_intra=0.2, p_affine=0.15, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.1, p_bcw=0.1, p_jccr=0.2))):
        pl = plans[idx]
        d = synth.picture_for_plan(pl, W, H, seed=905 + idx, tool_flags=ALL, log2_ctu=6, **kw)
        refs = {slot: synth.natural_picture(W, H, 910 + poc) for lst in pl.ref_slots for (slot, poc) in lst}
        serial = refdrv.reconstruct(d, refs, flags=refdrv.SIMD | refdrv.DERIVE_LFP)
        threaded = refdrv.reconstruct_threaded(d, refs, threads=threads)
        assert threaded["ms"] > 0
        for c in range(3):
            assert np.array_equal(serial["planes"][c], threaded["planes"][c]), "component %d of picture %d differs between the reference's scheduler and its stages called in turn" % (c, idx)
