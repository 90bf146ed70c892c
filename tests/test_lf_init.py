"""VVR_TOOL_LFP_ON_DEVICE: the deblocking edge parameters derived by the back-end (vvdec_amd/csrc/vvr_lf_init.h, k_lf_init) instead of supplied by the
caller.  The derivation is one source for host and device; here the product's host code runs on the stand-in runtime of tests/hoststub, whose
launch_lf_init executes that source on the CPU, and the tables it leaves in the picture's image are compared with the tables of the generated
description - which tests/test_extractor_roundtrip.py pins byte for byte to what LoopFilter::calcFilterStrengthsCTU derives for the same pictures
(oracle/_ref).  Compared: everything the filter kernels read of an entry (strengths, luma filter lengths, QPs of filtered components, the long-chroma
flag); entries of edges that are not filtered carry lengths and QPs in the reference's table that nothing reads.  The GPU side of the same path:
tests/test_gpu_parity.py::test_edge_parameters_derived_on_the_device."""
import ctypes as C
import numpy as np
import pytest
from vvdec_amd import abi, stream, synth
import test_host_glue as T
from test_host_glue import stub, Ctx          # noqa: F401  (fixture)
from test_extractor_roundtrip import CASES

pytestmark = T.pytestmark


def effective_differences(want, got, w4, h4, chroma):
    """cells where the two edge-parameter tables would make the deblocking kernels do different things -> list of texts"""
    bad = []
    for dr in range(2):
        a, b = want[dr].reshape(h4, w4), got[dr].reshape(h4, w4)
        bsa, bsb = a["bs"].astype(int), b["bs"].astype(int)
        pos = (np.arange(w4)[None, :] if dr == 0 else np.arange(h4)[:, None]) * 4 + np.zeros((h4, w4), int)
        grid = (pos % 16 == 0) if chroma else np.zeros((h4, w4), bool)
        def report(mask, what):
            if mask.any():
                y, x = np.argwhere(mask)[0]
                bad.append("dir %d: %s at %d cells, first (x4 %d, y4 %d): want %s got %s" % (dr, what, int(mask.sum()), x, y, a[y, x], b[y, x]))
        report((bsa & 3) != (bsb & 3), "luma strength")
        on = (bsa & 3) != 0
        report(on & (a["qp"][..., 0] != b["qp"][..., 0]), "luma qp")
        report(on & ((a["side_max_filt_length"] & 0x77) != (b["side_max_filt_length"] & 0x77)), "luma filter lengths")
        for c in (1, 2):
            sa, sb = (bsa >> (2 * c)) & 3, (bsb >> (2 * c)) & 3
            report(grid & (sa != sb), "chroma %d strength" % c)
            report(grid & (sa != 0) & (a["qp"][..., c] != b["qp"][..., c]), "chroma %d qp" % c)
            report(grid & (sa != 0) & ((a["flags"] & 0x20) != (b["flags"] & 0x20)), "chroma %d long-filter flag" % c)
    return bad


def derive(stub, d):
    """the picture through the product's host code with VVR_TOOL_LFP_ON_DEVICE -> the two tables launch_lf_init derived"""
    ctx = Ctx(stub, d.hdr.width, d.hdr.height, 8, log2_ctu=d.hdr.log2_ctu, bit_depth=d.hdr.bit_depth, chroma_format=d.hdr.chroma_format)
    try:
        d.hdr.tool_flags |= abi.TOOL_LFP_ON_DEVICE
        h = ctx.prepare(d)
        assert stub.vvr_submit_prepared(ctx.ctx, h) >= 0, stub.vvr_last_error(ctx.ctx).decode()
        assert stub.vvr_sync(ctx.ctx) == abi.VVR_OK, stub.vvr_last_error(ctx.ctx).decode()
        p0, p1 = C.c_void_p(), C.c_void_p()
        stub.vvt_last_lfp.argtypes = [C.c_void_p, C.c_void_p]
        n = stub.vvt_last_lfp(C.byref(p0), C.byref(p1))
        assert n == d.w4 * d.h4
        dt = d.lfp[0].dtype
        got = [np.frombuffer(C.string_at(p.value, n * dt.itemsize), dt).copy() for p in (p0, p1)]
        stub.vvr_free_prepared(ctx.ctx, h)
        return got
    finally:
        d.hdr.tool_flags &= ~abi.TOOL_LFP_ON_DEVICE
        ctx.close()


@pytest.mark.parametrize("name,W,H,l2,idx,seed,tools,kw", [c for c in CASES if not (c[6] & abi.TOOL_DEBLOCK_OFF)], ids=[c[0] for c in CASES if not (c[6] & abi.TOOL_DEBLOCK_OFF)])
@pytest.mark.parametrize("affine_on_device", [False, True], ids=["motion_field", "affine_mv_on_device"])
def test_derived_edge_parameters_equal_the_reference_tables(stub, name, W, H, l2, idx, seed, tools, kw, affine_on_device):
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    d = synth.picture_for_plan(plans[idx], W, H, seed=seed, tool_flags=tools | (abi.TOOL_AFFINE_MV_ON_DEVICE if affine_on_device else 0), log2_ctu=l2, **kw)
    got = derive(stub, d)
    bad = effective_differences(d.lfp, got, d.w4, d.h4, d.hdr.chroma_format != 0)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("seed,kw", [
    (701, dict(p_intra=0.1, p_affine=0.35, p_sbtmvp=0.3, p_sbt=0.3, p_geo=0.15, p_ciip=0.1, p_coded=0.3)),            # sub-block edges next to transform edges, few coded blocks: motion decides
    (702, dict(p_intra=0.3, p_isp=0.5, p_bdpcm=0.3, p_mip=0.2, min_cu_log2=2, p_split_scale=1.7)),                       # intra sub-partitions, BDPCM on both sides of an edge, 4-wide CUs
    (703, dict(p_intra=0.2, num_slices=4, tile_cols=3, tile_rows=2, p_affine=0.2, p_sbtmvp=0.2)),                         # slice and tile boundaries the filter may cross
    (704, dict(p_intra=0.05, p_affine=0.5, p_split_scale=0.4, p_coded=0.2)),                                              # large affine CUs: sub-block edges 8 apart, long filters capped
])
def test_derived_edge_parameters_on_b_pictures_with_many_edge_kinds(stub, seed, kw):
    tools = T.TOOLS | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    for pl in plans[1:]:
        d = synth.picture_for_plan(pl, 416, 240, seed=seed, tool_flags=tools, log2_ctu=7, **kw)
        got = derive(stub, d)
        bad = effective_differences(d.lfp, got, d.w4, d.h4, True)
        assert not bad, "POC %d\n%s" % (pl.poc, "\n".join(bad))


def test_tables_are_not_uploaded(stub):
    """with the flag the description's tables do not travel: the upload shrinks by their size (8 bytes per cell and direction)"""
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    d = synth.picture_for_plan(plans[2], 416, 240, seed=705, tool_flags=T.TOOLS, log2_ctu=7, p_intra=0.2)
    stub.vvt_take_h2d.argtypes = [C.c_void_p, C.c_void_p]
    sizes = []
    for flag in (0, abi.TOOL_LFP_ON_DEVICE):
        ctx = Ctx(stub, 416, 240, 8)
        d.hdr.tool_flags |= flag
        cnt, nbytes = C.c_size_t(), C.c_size_t()
        stub.vvt_take_h2d(C.byref(cnt), C.byref(nbytes))
        pic = d.c()
        stub.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]
        assert stub.vvr_submit(ctx.ctx, C.byref(pic)) >= 0, stub.vvr_last_error(ctx.ctx).decode()
        assert stub.vvr_sync(ctx.ctx) == abi.VVR_OK
        stub.vvt_take_h2d(C.byref(cnt), C.byref(nbytes))
        sizes.append(nbytes.value)
        d.hdr.tool_flags &= ~abi.TOOL_LFP_ON_DEVICE
        ctx.close()
    saved = sizes[0] - sizes[1]
    assert 0.8 * 16 * d.w4 * d.h4 < saved <= 16 * d.w4 * d.h4 + 512, sizes


def test_parser_fed_pictures_against_the_reference_lf_init():
    """every picture of the bitstreams in tests/bitstreams (written by tools/mini_vvenc.py, parsed by the reference's own parser): the drop-in decoder library in
    its self-check mode (VVDEC_AMD_LF_INIT=2, integration/DecLibReconDropIn.cpp) runs the reference's LF_INIT task AND the back-end's derivation (the source
    k_lf_init is compiled from) on the flattened picture and counts the table entries the deblocking filter would see differently - none.  This is where the
    generated pictures of the tests above cannot reach: what the real parser leaves in CodingStructure (found this way in round 4: SbTMVP CUs carry the
    affine flag, LoopFilter.cpp:920; the chroma QPs of an ISP CU live in its last transform unit, :1121-1123).  No GPU: the back-end is the stand-in runtime."""
    import os, re, sys
    import refdrv
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import dropin_decode as dd
    streams = dd.find_streams(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bitstreams"))
    if not refdrv.dropin_available() or not os.path.exists(dd.APP_DROPIN) or not streams:
        pytest.skip("oracle/_ref/vvdecapp_dropin not built (needs /root/reference) or no bitstreams")
    stub_lib = T.build_stub()
    env0 = os.environ.get("VVDEC_AMD_LF_INIT")
    os.environ["VVDEC_AMD_LF_INIT"] = "2"
    try:
        checked = differ = 0
        bad = []
        for b in streams:
            r, _ = dd.run_app(dd.APP_DROPIN, ["-b", b, "-t", "4", "-v", "0"], preload=stub_lib)
            m = re.findall(r"edge parameters: (\d+) entries checked against the reference's LF_INIT, (\d+) differ", r.stdout + r.stderr)
            c, d = sum(int(a) for a, _ in m), sum(int(x) for _, x in m)
            checked += c
            differ += d
            if d or (not c and "nodeblock" not in b):            # (a stream that switches deblocking off has no table)
                bad.append((os.path.basename(b), c, d))
    finally:
        if env0 is None:
            del os.environ["VVDEC_AMD_LF_INIT"]
        else:
            os.environ["VVDEC_AMD_LF_INIT"] = env0
    assert not bad and differ == 0 and checked > 2_000_000, (checked, differ, bad[:5])


@pytest.mark.parametrize("mode", ["0", "1"])
def test_dropin_lf_init_modes_on_the_stand_in_runtime(mode):
    """the drop-in decodes a stream end to end on the stand-in runtime with LF_INIT left to the back-end (the default: no edge-parameter table exists on the host) and
    with the reference's own LF_INIT (VVDEC_AMD_LF_INIT=1: its tables are copied and handed over): both ways every picture passes the back-end's checks and comes back"""
    import os, re, sys
    import refdrv
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import dropin_decode as dd
    bit = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bitstreams", "mini_all_tools_ctu128_384x256", "mini_all_tools_ctu128_384x256.bit")
    if not refdrv.dropin_available() or not os.path.exists(dd.APP_DROPIN) or not os.path.exists(bit):
        pytest.skip("oracle/_ref/vvdecapp_dropin not built (needs /root/reference) or the stream missing")
    env0 = os.environ.get("VVDEC_AMD_LF_INIT")
    os.environ["VVDEC_AMD_LF_INIT"] = mode
    try:
        r, _ = dd.run_app(dd.APP_DROPIN, ["-b", bit, "-t", "4", "-v", "3"], preload=T.build_stub())
    finally:
        if env0 is None:
            del os.environ["VVDEC_AMD_LF_INIT"]
        else:
            os.environ["VVDEC_AMD_LF_INIT"] = env0
    out = r.stdout + r.stderr
    frames, _ = dd.frames_and_fps(out)
    assert r.returncode == 0 and frames and frames >= 5 and "vvdec_amd:" not in out, (r.returncode, out[-600:])
