"""GPU parity tests (run on the MI355X box with -m gpu): HIP path vs the CPU oracle, through the C ABI.

The oracle (oracle/libvvoracle.so, plain-C restatement pinned against the real reference classes) is the checker;
the thing under test is vvdec_amd/libvvdec_amd.so.  All comparisons are bit-exact (VVC is an integer specification)."""
import hashlib
import numpy as np
import pytest

import refdrv
from vvdec_amd import abi, synth, stream

pytestmark = pytest.mark.gpu

TOOLS = abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS


def _run_stream(W, H, frames, gop, seed, tools, streams=2, log2_ctu=7, check=True, intra=False, bit_depth=10, chroma_format=1, post=None, kernels=None, **kw):
    """intra=False: POC 0 is an uploaded picture and all CUs are inter; intra=True: POC 0 is an I picture reconstructed by the
    back-end and the B pictures contain intra CUs (p_intra)."""
    import vvdec_amd
    plans, nslots = stream.ra_plan(frames, gop=gop, seed_poc0_is_external=not intra)
    ncomp = 3 if chroma_format else 1
    geo = dict(log2_ctu=log2_ctu, bit_depth=bit_depth, chroma_format=chroma_format)
    rec = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=streams, host_threads=3, **geo)     # pictures prepared by worker threads
    seed_pic = synth.natural_picture(W, H, seed, bit_depth=bit_depth)[:ncomp]
    cpu = {}
    if not intra:
        rec.write_picture(0, seed_pic)
        cpu = {0: seed_pic}
        kw.setdefault("p_intra", 0.0)
    hashes = []
    descs = [synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, alloc=rec.host_array, **geo, **kw) for pl in plans]     # records in pinned memory of `rec`
    if post:
        for d in descs:
            post(d)
    jobs = [rec.decompress_picture(d) for d in descs]          # everything in flight: the back-end orders the dependencies
    rec.sync()
    # verify in decode order; the CPU oracle consumes its own previous outputs as references.  A slot is overwritten
    # later in the stream, so pictures are read back in a second, serial pass.
    rec2 = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=1, **geo)
    if kernels is not None:
        rec2.enable_stats()         # which kernels the pictures take (HIP events around every launch)
    if not intra:
        rec2.write_picture(0, seed_pic)
    for pl, d in zip(plans, descs):
        job = rec2.decompress_picture(d)
        rec2.wait(job)
        got = rec2.read_picture(pl.slot)
        hashes.append(hashlib.md5(b"".join(p.tobytes() for p in got)).hexdigest())
        if check:
            want = refdrv.oracle_reconstruct(d, cpu)
            for c in range(ncomp):
                assert np.array_equal(got[c], want[c]), "POC %d comp %d: %d samples differ" % (pl.poc, c, int((got[c] != want[c]).sum()))
            cpu[pl.slot] = want
            nd = getattr(d, "num_dmvr", 0)
            if nd:      # refined MVs handed back to the host (DecCu::TaskFinishMotionInfo)
                assert np.array_equal(rec2.read_dmvr(job, nd), refdrv.oracle_dmvr(nd)), "POC %d: DMVR delta MVs differ" % pl.poc
    # the pipelined run must have produced the same final pictures in the slots that were not overwritten
    last = {}
    for pl in plans:
        last[pl.slot] = pl
    for slot, pl in last.items():
        a = rec.read_picture(slot)
        b = rec2.read_picture(slot)
        for c in range(ncomp):
            assert np.array_equal(a[c], b[c]), "pipelined vs serial differ in slot %d" % slot
    if kernels is not None:
        for e in rec2.stats():
            kernels[e["name"]] = kernels.get(e["name"], 0) + e["launches"]
    rec.close()
    rec2.close()
    return hashes


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_small_stream_bit_exact(built, seed):
    _run_stream(256, 128, 9, 4, seed, TOOLS)


def test_non_ctu_multiple_size(built):
    _run_stream(416, 240, 5, 4, 11, TOOLS)


def test_ctu64_and_ctu32(built):
    _run_stream(256, 192, 5, 4, 5, TOOLS, log2_ctu=6)
    _run_stream(128, 96, 5, 4, 6, TOOLS, log2_ctu=5)


def test_stage_subsets(built):
    _run_stream(256, 128, 5, 4, 21, abi.TOOL_DEP_QUANT)                                     # deblock only
    _run_stream(256, 128, 5, 4, 22, abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA)                # SAO without ALF
    _run_stream(256, 128, 5, 4, 23, abi.TOOL_ALF | abi.TOOL_DEBLOCK_OFF)                    # ALF without SAO / deblock
    _run_stream(256, 128, 5, 4, 24, TOOLS, p_coded=0.9, p_coded_chroma=0.8, p_small_corner=0.2, p_mts=0.5, p_ts=0.2)   # residual heavy


def test_the_kernels_beside_the_fused_passes_are_exercised(built):
    """Pictures the fused passes do not cover - CTUs of 32, picture-header virtual boundaries, a picture with only one of SAO / ALF, one without deblocking
    but with LMCS, the stage-wise `stop_after` runs - take k_lmcs, k_deblock4 (in place), k_sao, k_alf_luma + k_alf_chroma(_tile) and k_copy; an inter picture
    that takes the CTU-tile path of the intra stage (IBC) with LMCS chroma scaling takes k_resi_add.  Every one of them is run here, bit-exact against the
    oracle, and the back-end's statistics say that it was THAT kernel (round-5 verdict: a GPU test per such branch, or the kernels go)"""
    T = TOOLS | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE
    k = {}
    _run_stream(256, 192, 5, 4, 71, T, intra=True, log2_ctu=5, p_intra=0.2, kernels=k)                      # CTU 32: no fused pass covers it
    assert all(k.get(n, 0) > 0 for n in ("k_lmcs", "k_deblock4", "k_sao", "k_alf_planes")), k
    assert not k.get("k_alf") and not k.get("k_deblock_v"), k
    k = {}
    _run_stream(512, 384, 3, 2, 72, T, intra=True, log2_ctu=6, p_intra=0.2, virtual_boundaries=1 | (1 << 2), kernels=k)      # picture-header virtual boundaries
    assert all(k.get(n, 0) > 0 for n in ("k_lmcs", "k_deblock4", "k_sao", "k_alf_planes")), k
    k = {}
    _run_stream(256, 128, 3, 2, 73, abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_DEP_QUANT, log2_ctu=5, kernels=k)     # SAO without ALF, CTU 32: SAO into the scratch picture, copied back
    assert k.get("k_sao", 0) > 0 and k.get("k_copy", 0) > 0 and not k.get("k_alf_planes"), k
    k = {}
    _run_stream(256, 128, 3, 2, 74, abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEBLOCK_OFF | abi.TOOL_LMCS, log2_ctu=5, kernels=k)      # ALF alone, no deblocking: the inverse luma map as a pass of its own
    assert k.get("k_alf_planes", 0) > 0 and k.get("k_copy", 0) > 0 and k.get("k_lmcs", 0) > 0 and not k.get("k_deblock4") and not k.get("k_sao"), k
    k = {}
    _run_stream(256, 128, 3, 2, 75, TOOLS | abi.TOOL_LMCS | abi.TOOL_DEBLOCK_OFF, kernels=k)                 # CTU 128 without deblocking: fused SAO + ALF behind k_lmcs
    assert k.get("k_lmcs", 0) > 0 and k.get("k_alf", 0) > 0 and not k.get("k_deblock_v") and not k.get("k_deblock4"), k
    k = {}
    _run_stream(256, 128, 5, 4, 76, T | abi.TOOL_IBC, intra=True, p_intra=0.2, p_ibc=0.2, p_coded=0.8, p_coded_chroma=0.7, kernels=k)      # IBC in B pictures: CTU tiles, scaled inter chroma residuals between the launches
    assert k.get("k_resi_add", 0) > 0 and k.get("k_intra", 0) > 0, k


def test_1080p_frame(built):
    _run_stream(1920, 1080, 3, 2, 31, TOOLS)


def test_4k_determinism_and_oracle(built):
    """Full-size property test: the 4K pictures of BASELINE config 2 reconstruct identically with 1 and 4 streams in
    flight (checked inside _run_stream) and the first B picture equals the oracle."""
    h1 = _run_stream(3840, 2160, 3, 2, 41, TOOLS, streams=4, check=True)
    h2 = _run_stream(3840, 2160, 3, 2, 41, TOOLS, streams=1, check=False)
    assert h1 == h2


TOOLS_I = TOOLS | abi.TOOL_LFNST
TOOLS_B = TOOLS_I | abi.TOOL_BDOF
TOOLS_D = TOOLS_I | abi.TOOL_DMVR
TOOLS_DB = TOOLS_I | abi.TOOL_DMVR | abi.TOOL_BDOF
TOOLS_A = TOOLS_DB | abi.TOOL_PROF


@pytest.mark.parametrize("seed", [51, 52])
def test_intra_stream_bit_exact(built, seed):
    """I picture + B pictures with intra CUs: planar/DC/angular (+wide angle), PDPC, MRL, reference smoothing, BDPCM, LFNST."""
    _run_stream(256, 128, 9, 4, seed, TOOLS_I, intra=True)


def test_intra_heavy(built):
    _run_stream(416, 240, 5, 4, 61, TOOLS_I, intra=True, p_intra=0.5, p_coded=0.7, p_coded_chroma=0.6, p_small_corner=0.3, p_lfnst=0.6, p_mrl=0.4, p_bdpcm=0.1)
    _run_stream(128, 128, 3, 2, 62, TOOLS_I, intra=True, p_intra=1.0, p_coded=0.9, p_coded_chroma=0.9, p_lfnst=0.7, p_split_scale=1.5)
    _run_stream(256, 192, 3, 2, 63, TOOLS_I, intra=True, log2_ctu=6)
    _run_stream(128, 96, 3, 2, 64, TOOLS_I, intra=True, log2_ctu=5)


def test_1080p_intra_stream(built):
    _run_stream(1920, 1080, 3, 2, 71, TOOLS_I, intra=True, streams=3)


@pytest.mark.parametrize("seed", [81, 82])
def test_bdof_stream(built, seed):
    """bi-predicted CUs with equal POC distance in opposite directions take the BDOF path (InterPrediction.cpp:1407-1427)"""
    _run_stream(256, 128, 9, 8, seed, TOOLS_B, intra=True)
    _run_stream(416, 240, 5, 4, seed + 10, TOOLS_B, intra=True, p_bi=0.9, p_intra=0.05, mv_sigma=3.0)


def test_bdof_1080p(built):
    _run_stream(1920, 1080, 3, 2, 91, TOOLS_B, intra=True, streams=3)


@pytest.mark.parametrize("tools,seed", [(TOOLS_D, 101), (TOOLS_DB, 102)])
def test_dmvr_stream(built, tools, seed):
    """merge-mode bi-predicted CUs with mirrored references run DMVR (+BDOF per sub-block); delta MVs are read back"""
    _run_stream(256, 128, 9, 8, seed, tools, intra=True)
    _run_stream(416, 240, 5, 4, seed + 10, tools, intra=True, p_bi=0.9, p_intra=0.05, mv_sigma=2.0)


def test_dmvr_bdof_1080p(built):
    _run_stream(1920, 1080, 3, 2, 111, TOOLS_DB, intra=True, streams=3)


@pytest.mark.parametrize("tools,seed", [(TOOLS_DB, 121), (TOOLS_A, 122)])
def test_affine_stream(built, tools, seed):
    """4- and 6-parameter affine CUs (sub-block MVs from the motion field), with and without PROF, uni and bi"""
    _run_stream(256, 128, 9, 8, seed, tools, intra=True, p_affine=0.4)
    _run_stream(416, 240, 5, 4, seed + 10, tools, intra=True, p_bi=0.7, p_intra=0.05, mv_sigma=2.0, p_affine=0.6)


def test_affine_prof_1080p(built):
    _run_stream(1920, 1080, 3, 2, 131, TOOLS_A, intra=True, streams=3, p_affine=0.3)


def test_gpm_stream(built):
    """geometric partitioning: two uni-predictions blended with the angle/offset dependent weight masks"""
    _run_stream(256, 128, 9, 8, 141, TOOLS_A, intra=True, p_geo=0.4, p_affine=0.1)
    _run_stream(416, 240, 5, 4, 142, TOOLS_A, intra=True, p_bi=0.7, p_intra=0.05, mv_sigma=2.0, p_geo=0.6)
    _run_stream(1920, 1080, 3, 2, 143, TOOLS_A, intra=True, streams=3, p_geo=0.3)


def test_ciip_stream(built):
    """combined inter/intra prediction: inter prediction from k_mc, planar intra + blend + residual in the intra wavefront"""
    _run_stream(256, 128, 9, 8, 151, TOOLS_A, intra=True, p_ciip=0.4, p_intra=0.2)
    _run_stream(416, 240, 5, 4, 152, TOOLS_A, intra=True, p_bi=0.7, p_intra=0.3, mv_sigma=2.0, p_ciip=0.6, p_coded=0.7)
    _run_stream(1920, 1080, 3, 2, 153, TOOLS_A, intra=True, streams=3, p_ciip=0.3, p_geo=0.1, p_affine=0.1)


def test_sbtmvp_and_all_inter_tools(built):
    """SbTMVP (per-8x8 motion, uni/bi/identical) and a mix of every inter tool in one stream"""
    _run_stream(256, 128, 9, 8, 161, TOOLS_A, intra=True, p_sbtmvp=0.4, p_intra=0.1)
    _run_stream(416, 240, 5, 4, 162, TOOLS_A, intra=True, p_bi=0.7, p_intra=0.15, p_affine=0.2, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.15)
    _run_stream(1920, 1080, 3, 2, 163, TOOLS_A, intra=True, streams=3, p_affine=0.15, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.15)


def test_sub_block_transform(built):
    """cu_sbt_flag: residual in one half / quarter of an inter CU (DST-7 / DCT-8 pairs by position, 2-wide chroma blocks), with and
    without LMCS chroma residual scaling"""
    _run_stream(256, 128, 5, 4, 171, TOOLS_A, intra=True, p_sbt=0.6, p_intra=0.1, p_coded_chroma=0.5, p_jccr=0.2)
    T = TOOLS_A | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE
    _run_stream(416, 240, 5, 4, 172, T, intra=True, log2_ctu=6, p_sbt=0.5, p_intra=0.15, p_affine=0.2, p_geo=0.1, p_coded_chroma=0.5)
    _run_stream(1920, 1080, 3, 2, 173, T, intra=True, streams=3, p_sbt=0.3)


def test_weighted_prediction(built):
    """explicit weighted prediction: B pictures of a stream and one P picture (weights / offsets per reference and component)"""
    import vvdec_amd
    T = TOOLS_A | abi.TOOL_WP
    _run_stream(256, 128, 5, 4, 181, T, intra=True, p_intra=0.1, p_affine=0.2, p_sbtmvp=0.15, p_ciip=0.1, p_geo=0.1, p_bcw=0.2)
    _run_stream(1920, 1080, 3, 2, 182, T | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, intra=True, streams=3, p_affine=0.1, p_ciip=0.05)
    W, H = 416, 240
    rec = vvdec_amd.Reconstructor(W, H, num_slots=3, num_streams=1)
    refs = {1: synth.natural_picture(W, H, 183), 2: synth.natural_picture(W, H, 184)}
    for slot, pic in refs.items():
        rec.write_picture(slot, pic)
    p = synth.default_params(width=W, height=H, seed=185, tool_flags=T, slice_type=abi.SLICE_P, p_affine=0.2, p_sbtmvp=0.15, p_ciip=0.1, p_intra=0.1)
    p.poc, p.out_slot = 4, 0
    synth.set_refs(p, [(1, 0), (2, 8)], [])
    d = synth.generate(p)
    rec.wait(rec.decompress_picture(d))
    got = rec.read_picture(0)
    want = refdrv.oracle_reconstruct(d, refs)
    for c in range(3):
        assert np.array_equal(got[c], want[c]), "P picture comp %d: %d samples differ" % (c, int((got[c] != want[c]).sum()))
    rec.close()


@pytest.mark.parametrize("no_lfnst", [0, 1])
def test_scaling_lists(built, no_lfnst):
    """explicit scaling lists in the dequantisation (all matrix sizes, rectangular blocks, DC entries; flat for transform skip and,
    with sps_scaling_matrix_for_lfnst_disabled_flag, for LFNST blocks)"""
    T = TOOLS_A | abi.TOOL_SCALING_LIST | (abi.TOOL_SCALING_LIST_NO_LFNST if no_lfnst else 0)
    _run_stream(256, 128, 5, 4, 191, T, intra=True, p_intra=0.3, p_coded=0.8, p_coded_chroma=0.6, p_lfnst=0.5, p_sbt=0.2, p_jccr=0.2)
    _run_stream(416, 240, 3, 2, 192, T, intra=True, log2_ctu=5, p_coded=0.9, p_mts=0.5, p_ts=0.2, p_small_corner=0.2)
    _run_stream(1920, 1080, 3, 2, 193, T | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, intra=True, streams=3, p_coded=0.6)


def test_intra_sub_partitions(built):
    """ISP: four luma partitions predicted one after the other from the CU's reference line (2-wide partitions in pairs), implicit
    DST-7, LFNST, unsplit chroma incl. CCLM and LMCS chroma scaling"""
    _run_stream(256, 128, 5, 4, 211, TOOLS_A, intra=True, p_isp=0.6, p_intra=0.4, p_lfnst=0.4, p_coded=0.7)
    _run_stream(416, 240, 3, 2, 212, TOOLS_A | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, intra=True, log2_ctu=5, p_isp=0.5, p_intra=0.4, p_cclm=0.3, p_jccr=0.3, p_coded_chroma=0.6)
    _run_stream(1920, 1080, 3, 2, 213, TOOLS_A, intra=True, streams=3, p_isp=0.3, p_cclm=0.2, p_mip=0.1)
    _run_stream(256, 192, 3, 2, 214, TOOLS_A, intra=True, log2_ctu=6, p_isp=0.8, p_split_scale=1.6)


def test_cclm_collocated_chroma(built):
    """CCLM / MDLM with sps_chroma_vertical_collocated_flag = 1: the 5-tap cross down-sampling of the luma"""
    T = TOOLS_A | abi.TOOL_CCLM_COLLOC
    _run_stream(256, 128, 5, 4, 221, T, intra=True, p_cclm=0.6, p_intra=0.4)
    _run_stream(416, 240, 3, 2, 222, T | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, intra=True, log2_ctu=5, p_cclm=0.5, p_intra=0.4, p_isp=0.3)
    _run_stream(1920, 1080, 2, 1, 223, T, intra=True, p_cclm=0.4)


def test_config3_8k_and_config5_all_intra(built):
    """BASELINE configs 3 and 5 as parity cases: one 8K I + B pair, and 4K all-intra pictures with the intra / LFNST heavy mix
    and dual-tree chroma"""
    _run_stream(7680, 4320, 2, 1, 231, TOOLS_A | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, intra=True, streams=2, p_coded=0.5, p_affine=0.06, p_geo=0.03, p_ciip=0.03, p_cclm=0.1, p_mip=0.05, p_isp=0.05)
    import vvdec_amd
    W, H = 3840, 2160
    rec = vvdec_amd.Reconstructor(W, H, num_slots=2, num_streams=2)
    for k in range(2):
        p = synth.default_params(width=W, height=H, seed=240 + k, tool_flags=TOOLS_A, slice_type=abi.SLICE_I, base_qp=22, p_coded=0.7, p_small_corner=0.5,
                                 p_lfnst=0.4, p_isp=0.1, p_mip=0.1, p_cclm=0.15, dual_tree=1.0)
        p.poc, p.out_slot = k, k
        d = synth.generate(p)
        rec.wait(rec.decompress_picture(d))
        got = rec.read_picture(k)
        want = refdrv.oracle_reconstruct(d, {})
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "all-intra picture %d comp %d: %d samples differ" % (k, c, int((got[c] != want[c]).sum()))
    rec.close()


def _pictures_against_the_oracle(W, H, plans, nslots, mix, tools, seed, lanes, threads):
    """the pictures of a plan through the asynchronous path (worker threads, several pictures in flight, records in pinned memory), every picture
    kept in a slot of its own; then picture by picture against the CPU oracle, which consumes its own previous outputs as reference pictures"""
    import vvdec_amd
    rec = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=lanes, host_threads=threads)
    assert len({pl.slot for pl in plans}) == len(plans), "every picture needs a slot of its own here"
    descs = [synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, alloc=rec.host_array, **mix) for pl in plans]
    jobs = [rec.decompress_picture(d) for d in descs]
    rec.sync()
    cpu = {}
    for pl, d, job in zip(plans, descs, jobs):
        got = rec.read_picture(pl.slot)
        want = refdrv.oracle_reconstruct(d, cpu)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "%dx%d POC %d comp %d: %d samples differ" % (W, H, pl.poc, c, int((got[c] != want[c]).sum()))
        cpu[pl.slot] = want
        nd = getattr(d, "num_dmvr", 0)
        if nd:
            assert np.array_equal(rec.read_dmvr(job, nd), refdrv.oracle_dmvr(nd)), "POC %d: DMVR delta MVs differ" % pl.poc
    rec.close()
    return len(plans)


def test_baseline_sizes_picture_by_picture(built):
    """Parity at BASELINE.json's sizes, every picture against the oracle, on the benchmark's own streams (bench.CONFIGS / bench.MIX): one full GOP 32 of
    config 2 (the IRAP + 32 B pictures of six temporal layers at 3840x2160), 8 pictures of config 5 (4K all-intra, dual tree), 4 pictures of config 3
    (7680x4320)"""
    import bench
    tools = bench._tools(abi)
    W, H, mix, _, _ = bench.CONFIGS["4k"]
    plans, nslots = stream.ra_plan(33, gop=32, seed_poc0_is_external=False, pool=40)
    assert _pictures_against_the_oracle(W, H, plans, max(nslots, 40), mix, tools, 1234, lanes=4, threads=8) == 33
    W, H, mix, _, _ = bench.CONFIGS["allintra"]
    plans = [stream.PicPlan(poc=i, layer=0, slice_type=abi.SLICE_I, slot=i, ref_slots=([], [])) for i in range(8)]
    assert _pictures_against_the_oracle(W, H, plans, 8, mix, tools, 1234, lanes=4, threads=8) == 8
    W, H, mix, _, _ = bench.CONFIGS["8k"]
    plans, nslots = stream.ra_plan(5, gop=4, seed_poc0_is_external=False, pool=8)
    assert _pictures_against_the_oracle(W, H, plans[:4], max(nslots, 8), mix, tools, 1234, lanes=2, threads=8) == 4


def test_dual_tree_intra_pictures(built):
    """I pictures with separate luma and chroma coding trees (qtbtt_dual_tree_intra_flag): chroma CUs with derived / CCLM modes and
    their own LFNST, chroma edges of the deblocking from the chroma tree; the B pictures that follow are single tree"""
    _run_stream(256, 128, 5, 4, 241, TOOLS_A, intra=True, dual_tree=1.0, p_cclm=0.4, p_lfnst=0.5, p_isp=0.3, p_mip=0.2, p_coded_chroma=0.6)
    _run_stream(416, 240, 3, 2, 242, TOOLS_A | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, intra=True, log2_ctu=5, dual_tree=1.0, p_cclm=0.4, p_jccr=0.3, p_coded_chroma=0.6)
    _run_stream(1920, 1080, 3, 2, 243, TOOLS_A, intra=True, streams=3, log2_ctu=6, dual_tree=1.0, p_cclm=0.2, p_isp=0.1)
    # luma CUs down to 4x4 (4x4 MIP / LFNST / BDPCM blocks), implicit MTS
    _run_stream(256, 128, 3, 2, 244, TOOLS_A | abi.TOOL_IMPLICIT_MTS, intra=True, dual_tree=2.0, p_cclm=0.3, p_lfnst=0.4, p_isp=0.2, p_mip=0.3, p_coded=0.7, p_split_scale=1.5)
    _run_stream(1920, 1080, 2, 1, 245, TOOLS_A | abi.TOOL_IMPLICIT_MTS, intra=True, dual_tree=2.0, p_mip=0.1, p_isp=0.1, p_split_scale=1.3)
    # ISP on 4xN / Nx4 CUs: 1xN, Nx1 (1-D transforms), 2xN, Nx2 partitions, groups of four 1-wide partitions predicted together
    _run_stream(256, 128, 3, 2, 246, TOOLS_A | abi.TOOL_IMPLICIT_MTS, intra=True, dual_tree=3.0, p_isp=0.7, p_lfnst=0.3, p_coded=0.8, p_split_scale=1.8)
    _run_stream(416, 240, 2, 1, 247, TOOLS_A, intra=True, log2_ctu=5, dual_tree=3.0, p_isp=0.6, p_coded=0.8, p_split_scale=2.0)


def test_small_cus_and_local_dual_tree(built):
    """minimum CU size 4 in single-tree pictures: 4xN inter CUs with 2xN chroma blocks, Nx4 intra CUs with Nx2 chroma blocks, intra-only
    sub-trees with a local dual tree (luma-tree CUs down to 4x4, one chroma-tree CU per node), in I and B pictures"""
    T = TOOLS_A | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE
    _run_stream(256, 128, 5, 4, 251, TOOLS_A, intra=True, min_cu_log2=2, p_intra=0.3, p_split_scale=1.8, p_cclm=0.3, p_isp=0.2, p_mip=0.2)
    _run_stream(416, 240, 3, 2, 252, T, intra=True, log2_ctu=5, min_cu_log2=2, p_intra=0.25, p_split_scale=2.0, p_sbt=0.2, p_cclm=0.3, p_jccr=0.2, p_coded_chroma=0.5)
    _run_stream(1920, 1080, 3, 2, 253, T, intra=True, streams=3, min_cu_log2=2, p_split_scale=1.5, p_affine=0.1, p_ciip=0.05)
    # 4xN / Nx4 CIIP CUs: the 2-wide chroma blocks stay pure inter, the Nx2 ones are blended
    _run_stream(256, 128, 5, 4, 254, T, intra=True, min_cu_log2=2, p_intra=0.2, p_split_scale=1.8, p_ciip=0.6, p_coded=0.6, p_coded_chroma=0.5)


@pytest.mark.parametrize("bit_depth,chroma_format", [(8, 1), (10, 0), (8, 0)], ids=["8bit_420", "10bit_400", "8bit_400"])
def test_bit_depth_8_and_monochrome(built, bit_depth, chroma_format):
    """the other sample formats of the Main 10 profile: 8-bit samples (interpolation head-room, deblocking tc, SAO / ALF / LMCS scales
    change with the bit depth) and 4:0:0 pictures (luma only), every tool on, I and B pictures"""
    T = TOOLS_A | abi.TOOL_LMCS | (abi.TOOL_LMCS_CSCALE if chroma_format else 0)
    mix = dict(p_intra=0.25, p_affine=0.15, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.1, p_mip=0.2, p_isp=0.2, p_sbt=0.15, p_bcw=0.2)
    if chroma_format:
        mix.update(p_cclm=0.3, p_jccr=0.2, p_coded_chroma=0.5)
    _run_stream(256, 128, 5, 4, 271, T, intra=True, bit_depth=bit_depth, chroma_format=chroma_format, **mix)
    _run_stream(416, 240, 3, 2, 272, TOOLS_A | abi.TOOL_WP, intra=True, log2_ctu=6, bit_depth=bit_depth, chroma_format=chroma_format, **mix)


_IBC = abi.TOOL_IBC
_IBC_L = abi.TOOL_IBC | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE


@pytest.mark.parametrize("W,H,frames,gop,seed,extra,kw", [
    (256, 128, 5, 4, 261, _IBC, dict(p_ibc=0.5, p_intra=0.4, p_coded=0.5)),
    (416, 240, 3, 2, 262, _IBC_L, dict(log2_ctu=6, p_ibc=0.5, p_intra=0.3, p_split_scale=1.4, p_cclm=0.3, p_jccr=0.3, p_coded_chroma=0.5, p_ciip=0.1)),
    (256, 192, 3, 2, 263, _IBC_L, dict(log2_ctu=5, p_ibc=0.6, p_intra=0.5, p_isp=0.2, p_mip=0.2)),
    (256, 128, 3, 2, 264, _IBC_L, dict(dual_tree=2.0, p_ibc=0.5, p_split_scale=1.5, p_cclm=0.3)),
    (416, 240, 3, 2, 265, _IBC, dict(min_cu_log2=2, p_ibc=0.6, p_intra=0.4, p_split_scale=1.6)),
    (1920, 1080, 3, 2, 266, _IBC_L, dict(streams=3, p_ibc=0.3, p_affine=0.1, p_ciip=0.05)),
], ids=["ctu128", "ctu64_lmcs", "ctu32_lmcs", "dual_tree", "small_cus", "1080p"])
def test_intra_block_copy(built, W, H, frames, gop, seed, extra, kw):
    """IBC CUs: copies of reconstructed samples of the current picture (block vectors into the CTU itself and into the CTUs left of it),
    chroma at the halved vector, luma-only CUs of dual / local dual trees, LMCS chroma residual scaling, I and B pictures"""
    _run_stream(W, H, frames, gop, seed, TOOLS_A | extra, intra=True, **kw)


def test_joint_cbcr(built):
    """tu_joint_cbcr_residual: one coded chroma block, the other derived (all three modes, both signs)"""
    _run_stream(256, 128, 5, 4, 171, TOOLS_A, intra=True, p_jccr=0.7, p_coded_chroma=0.7, p_intra=0.3)
    _run_stream(416, 240, 5, 4, 172, TOOLS_A | abi.TOOL_JCCR_SIGN, intra=True, p_jccr=0.7, p_coded_chroma=0.7, p_intra=0.3)


def test_cclm(built):
    """chroma predicted from reconstructed luma: CCLM, MDLM_L, MDLM_T; the chroma CTU waits for the luma CTUs it reads"""
    _run_stream(256, 128, 5, 4, 181, TOOLS_A, intra=True, p_cclm=0.5, p_intra=0.3)
    _run_stream(416, 240, 5, 4, 182, TOOLS_A, intra=True, p_cclm=0.6, p_intra=0.5, log2_ctu=6, p_coded=0.6, p_coded_chroma=0.5)
    _run_stream(200, 136, 3, 2, 183, TOOLS_A, intra=True, p_cclm=0.7, p_intra=0.6, log2_ctu=5)
    _run_stream(1920, 1080, 3, 2, 184, TOOLS_A, intra=True, streams=3, p_cclm=0.4)


def test_mip(built):
    """matrix-based intra prediction: all three size classes, transposition, up-sampling in one or both directions, LFNST on top"""
    _run_stream(256, 128, 5, 4, 191, TOOLS_A, intra=True, p_mip=0.5, p_intra=0.3, p_cclm=0.2)
    _run_stream(416, 240, 5, 4, 192, TOOLS_A, intra=True, p_mip=0.6, p_intra=0.5, log2_ctu=6, p_coded=0.6, p_lfnst=0.5)
    _run_stream(200, 136, 3, 2, 193, TOOLS_A, intra=True, p_mip=0.7, p_intra=0.6, log2_ctu=5, p_split_scale=1.5)
    _run_stream(1920, 1080, 3, 2, 194, TOOLS_A, intra=True, streams=3, p_mip=0.4, p_cclm=0.2)


@pytest.mark.parametrize("cscale", [0, 1])
def test_lmcs(built, cscale):
    """LMCS: inter prediction forward-mapped, intra in the mapped domain, inverse mapping before the loop filters; with cscale
    the chroma residuals are scaled by a factor looked up from the reconstructed luma around the 64x64 VPDU"""
    T = TOOLS_A | abi.TOOL_LMCS | (abi.TOOL_LMCS_CSCALE if cscale else 0)
    _run_stream(256, 128, 5, 4, 201, T, intra=True, p_intra=0.3, p_cclm=0.2, p_mip=0.2, p_ciip=0.1, p_jccr=0.3)
    _run_stream(416, 240, 5, 4, 202, T, intra=True, p_intra=0.2, log2_ctu=6, p_affine=0.1, p_geo=0.1, p_sbtmvp=0.1, p_ciip=0.1)
    _run_stream(1920, 1080, 3, 2, 203, T, intra=True, streams=3)


def test_luma_adaptive_deblocking(built):
    """LADF: QP offset of every luma edge segment from the local luma level (LoopFilter::deriveLADFShift)"""
    T = TOOLS_A | abi.TOOL_LADF | abi.TOOL_LMCS
    _run_stream(256, 128, 5, 4, 211, T, intra=True, p_intra=0.3)
    _run_stream(416, 240, 5, 4, 212, T, intra=True, p_intra=0.15, log2_ctu=6, p_affine=0.2, p_sbtmvp=0.1)
    _run_stream(1920, 1080, 3, 2, 213, TOOLS_A | abi.TOOL_LADF, intra=True, streams=3)


@pytest.mark.parametrize("extra,kw", [
    (abi.TOOL_NO_LF_ACROSS_SLICES, dict(num_slices=3)),
    (0, dict(num_slices=3, tile_cols=2, tile_rows=2)),
    (abi.TOOL_NO_LF_ACROSS_TILES, dict(tile_cols=3, tile_rows=2)),
    (abi.TOOL_NO_LF_ACROSS_SLICES | abi.TOOL_NO_LF_ACROSS_TILES, dict(num_slices=3, tile_cols=2, tile_rows=2)),
])
def test_slices_and_tiles(built, extra, kw):
    """several slices / tiles per picture: availability of intra references, CCLM templates and the chroma-scaling neighbourhood ends at their
    boundaries (host glue), SAO and ALF stop there when the loop filters may not cross them (k_sao, k_alf_*)"""
    T = TOOLS_A | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | extra
    _run_stream(512, 384, 5, 4, 221, T, intra=True, log2_ctu=6, p_intra=0.3, p_cclm=0.3, p_ciip=0.1, p_coded_chroma=0.5, **kw)
    _run_stream(640, 256, 3, 2, 222, T, intra=True, log2_ctu=5, p_intra=0.2, p_affine=0.2, **kw)
    _run_stream(1920, 1080, 3, 2, 223, TOOLS_A | extra, intra=True, streams=3, **kw)


@pytest.mark.parametrize("extra,kw", [
    (abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_SCALING_LIST, dict(num_slices=3)),
    (abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_WP | abi.TOOL_NO_LF_ACROSS_SLICES, dict(num_slices=4, p_ciip=0.2, p_affine=0.2, p_sbtmvp=0.1)),
    (abi.TOOL_SCALING_LIST | abi.TOOL_WP | abi.TOOL_NO_LF_ACROSS_TILES, dict(num_slices=4, tile_cols=3, tile_rows=2)),
    (abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_WP | abi.TOOL_NO_LF_ACROSS_SLICES, dict(num_slices=4, intra_slices=0b0110)),      # I slices in B pictures
])
def test_slice_headers(built, extra, kw):
    """slices with headers of their own (vvr_picture::slices): dependent quantisation, LMCS, chroma residual scaling, scaling lists per slice
    (k_itrans, k_intra, the MC kernels' forward mapping, k_lmcs), deblocking offsets (k_deblock), ALF tables (k_alf_*), prediction weights (MC)"""
    T = TOOLS_A | extra
    seen = []

    def vary(d):
        synth.vary_slices(d, 230 + d.hdr.poc, intra_slices=kw.get("intra_slices", 0))
        seen.append(len(set(int(f) for f in d.slices["tool_flags"])))
    _run_stream(512, 384, 5, 4, 231, T, intra=True, log2_ctu=6, p_intra=0.3, p_cclm=0.3, p_coded=0.8, p_coded_chroma=0.6, post=vary, **kw)
    _run_stream(640, 256, 3, 2, 232, T, intra=True, log2_ctu=5, p_intra=0.2, p_coded=0.8, post=vary, **kw)
    _run_stream(1920, 1080, 3, 2, 233, T, intra=True, streams=3, p_coded=0.7, post=vary, **kw)
    assert max(seen) > 1


@pytest.mark.parametrize("vb,extra,kw", [
    (1 | (1 << 2), 0, dict()),
    (3 | (3 << 2), 0, dict(p_affine=0.3, p_sbtmvp=0.2)),
    (2 | (2 << 2) | 16, 0, dict()),
    (3 | (2 << 2) | 16, abi.TOOL_NO_LF_ACROSS_SLICES | abi.TOOL_NO_LF_ACROSS_TILES, dict(num_slices=3, tile_cols=2, tile_rows=2)),
])
def test_virtual_boundaries(built, vb, extra, kw):
    """virtual boundaries of the picture header: SAO leaves the columns / rows next to them alone (k_sao), ALF filters the parts of a CTU they cut out
    with a border of their own (k_alf_luma: one pass per part of a tile; k_alf_chroma and CC-ALF: per-sample clip); deblocking follows the host table"""
    T = TOOLS_A | abi.TOOL_LMCS | extra
    _run_stream(512, 384, 5, 4, 231, T, intra=True, log2_ctu=6, p_intra=0.3, p_cclm=0.3, virtual_boundaries=vb, **kw)
    _run_stream(640, 256, 3, 2, 232, T, intra=True, log2_ctu=5, p_intra=0.2, virtual_boundaries=vb, **kw)
    _run_stream(1920, 1080, 3, 2, 233, TOOLS_A | extra, intra=True, streams=3, virtual_boundaries=vb, **kw)


@pytest.mark.parametrize("W,H,off,kw", [
    (512, 256, 512, dict(log2_ctu=6, mv_sigma=12.0)),
    (512, 256, 480, dict(log2_ctu=6, p_bi=0.9, mv_sigma=4.0)),
    (640, 256, 640, dict(p_affine=0.4, p_sbtmvp=0.2, p_geo=0.2, p_ciip=0.1)),
    (384, 256, 320, dict(log2_ctu=5, p_affine=0.3, p_bi=0.8, mv_sigma=20.0)),
    (1920, 1080, 1920, dict(p_affine=0.1, p_geo=0.05, p_sbtmvp=0.05)),
])
def test_reference_wrap_around(built, W, H, off, kw):
    """horizontal reference wrap-around (360-degree video): every MC kernel reads the reference with the period of the picture header where the
    reference decoder reads its wrap copy, and with the ordinary clamp after an MV was moved by one period (k_mc, k_mc_dmvr incl. its padded local
    copy, k_mc_affine per sub-block)"""
    _run_stream(W, H, 5, 4, 311, TOOLS_A | abi.TOOL_LMCS, intra=True, p_intra=0.1, wrap_offset=off, **kw)


@pytest.mark.parametrize("sp,extra,kw", [
    (1 | (1 << 1) | (1 << 3), 0, dict(tile_cols=2, tile_rows=2, mv_sigma=24.0)),
    (1 | (2 << 1) | (2 << 3), 0, dict(tile_cols=2, tile_rows=2, p_bi=0.9, mv_sigma=16.0)),
    (1 | (1 << 1) | (0 << 3), 0, dict(tile_cols=3, tile_rows=2, p_affine=0.4, p_sbtmvp=0.2, p_geo=0.2, p_ciip=0.1, mv_sigma=30.0)),
    (1 | (2 << 1) | (1 << 3), abi.TOOL_NO_LF_ACROSS_SLICES, dict(tile_cols=3, tile_rows=1, p_cclm=0.3, mv_sigma=40.0)),
])
def test_subpictures(built, sp, extra, kw):
    """sub-pictures: motion compensation of a CU in a sub-picture that is treated as a picture reads the reference pictures inside that sub-picture only (MV
    clip and clamp rectangle in k_mc, k_mc_dmvr, k_mc_affine), SAO / ALF of a sub-picture whose flag says so stop at its boundary"""
    T = TOOLS_A | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | extra
    _run_stream(512, 384, 5, 4, 321, T, intra=True, log2_ctu=6, p_intra=0.15, subpics=sp, **kw)
    _run_stream(640, 256, 3, 2, 322, T, intra=True, log2_ctu=5, p_intra=0.1, subpics=sp, **kw)
    _run_stream(1920, 1080, 3, 2, 323, TOOLS_A | extra, intra=True, streams=3, subpics=sp, **kw)


@pytest.mark.parametrize("name,W,H,frames,gop,seed,extra,kw", [
    ("inter_tools", 416, 240, 9, 4, 801, abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_intra=0.15, p_affine=0.25, p_sbtmvp=0.2, p_geo=0.15, p_ciip=0.1, p_sbt=0.2, p_coded=0.4)),
    ("affine_spanned_on_the_device", 512, 384, 5, 4, 802, abi.TOOL_AFFINE_MV_ON_DEVICE, dict(p_intra=0.05, p_affine=0.5, p_sbtmvp=0.2, p_coded=0.3, log2_ctu=6)),
    ("intra_tools_dual_tree", 384, 256, 5, 4, 803, abi.TOOL_LMCS, dict(p_intra=0.3, dual_tree=2.0, p_isp=0.3, p_bdpcm=0.2, p_cclm=0.3, p_mip=0.2, log2_ctu=6)),
    ("small_cus_local_dual_tree_ibc", 264, 200, 5, 4, 804, abi.TOOL_IBC, dict(p_intra=0.3, p_ibc=0.3, min_cu_log2=2, p_split_scale=1.7, p_isp=0.2, p_sbt=0.2)),
    ("slices_tiles_virtual_boundaries", 512, 384, 5, 4, 805, abi.TOOL_NO_LF_ACROSS_SLICES | abi.TOOL_NO_LF_ACROSS_TILES | abi.TOOL_LADF, dict(p_intra=0.2, p_affine=0.2, num_slices=3, tile_cols=2, tile_rows=2, virtual_boundaries=2 | (2 << 2) | 16, log2_ctu=6)),
    ("subpictures", 512, 384, 5, 4, 806, 0, dict(p_intra=0.15, p_sbtmvp=0.2, tile_cols=2, tile_rows=2, subpics=1 | (2 << 1) | (1 << 3), mv_sigma=24.0, log2_ctu=6)),
    ("monochrome_8bit", 256, 128, 5, 4, 807, 0, dict(p_intra=0.25, p_affine=0.2, chroma_format=0, bit_depth=8, log2_ctu=6)),
    ("1080p", 1920, 1080, 3, 2, 808, abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_AFFINE_MV_ON_DEVICE, dict(p_intra=0.15, p_affine=0.2, p_sbtmvp=0.1, p_geo=0.1, streams=3)),
])
def test_edge_parameters_derived_on_the_device(built, name, W, H, frames, gop, seed, extra, kw):
    """VVR_TOOL_LFP_ON_DEVICE (SURVEY 8(a) a22 on the GPU: the reference's LF_INIT task, LoopFilter::calcFilterStrengthsCTU): the description travels WITHOUT its
    edge-parameter tables (vvdec_amd/desc.py leaves the pointers NULL under the flag), k_lf_maps / k_lf_scatter / k_lf_init derive them from the CU / TU
    records, and the pictures are the ones the oracle reconstructs with the description's own tables (pinned to the reference's by
    tests/test_extractor_roundtrip.py).  The same derivation on the CPU, entry by entry: tests/test_lf_init.py"""
    kw = dict(kw)
    geo = {k: kw.pop(k) for k in ("log2_ctu", "bit_depth", "chroma_format", "streams") if k in kw}
    _run_stream(W, H, frames, gop, seed, TOOLS_A | abi.TOOL_LFP_ON_DEVICE | extra, intra=True, **geo, **kw)


def test_affine_motion_spanned_on_the_device(built):
    """VVR_TOOL_AFFINE_MV_ON_DEVICE: the back-end spans the sub-block MVs of affine CUs from the control-point MVs (PU::setAllAffineMv)
    instead of reading them from the motion field - with the motion of the affine CUs wiped from the description the pictures are the ones the oracle
    reconstructs from the complete description (4- and 6-parameter models, PROF, the single fallback vector of widely spread models)"""
    import vvdec_amd
    for W, H, seed, kw in ((416, 240, 291, dict(p_affine=0.6, p_intra=0.05)), (512, 384, 292, dict(p_affine=0.5, mv_sigma=40.0, log2_ctu=6)), (1920, 1080, 293, dict(p_affine=0.3))):
        plans, nslots = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
        rec = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=2, log2_ctu=kw.get("log2_ctu", 7))
        dpb = {}
        n_aff = 0
        for pl in plans:
            d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=TOOLS_A, **kw)
            refs = {slot: dpb[slot] for lst in pl.ref_slots for (slot, _) in lst}
            want = refdrv.oracle_reconstruct(d, refs)
            # wipe the motion of the affine CUs and let the device span it
            for cu in d.cu[(d.cu["flags"] & abi.CU_AFFINE) != 0]:
                x4, y4, w4, h4 = int(cu["x"]) >> 2, int(cu["y"]) >> 2, int(cu["w"]) >> 2, int(cu["h"]) >> 2
                d.motion.reshape(d.h4, d.w4)["mv"][y4:y4 + h4, x4:x4 + w4] = 0
                n_aff += 1
            d.hdr.tool_flags |= abi.TOOL_AFFINE_MV_ON_DEVICE
            rec.wait(rec.decompress_picture(d))
            got = rec.read_picture(pl.slot)
            for c in range(3):
                assert np.array_equal(got[c], want[c]), "POC %d comp %d: %d samples differ" % (pl.poc, c, int((got[c] != want[c]).sum()))
            dpb[pl.slot] = want
        rec.close()
        assert n_aff > 10


def test_collocated_motion_on_the_device(built):
    """VVR_TOOL_COL_MOTION: the collocated motion the back-end hands out - the motion field at every second 4x4 unit with the
    DMVR-refined MVs written by the DMVR kernel - is what the reference's DecCu::TaskFinishMotionInfo leaves for the same picture (its final motion
    field, sub-sampled), through vvr_submit with worker threads and through vvr_prepare / vvr_submit_prepared"""
    import vvdec_amd
    if not refdrv.available():
        pytest.skip("reference build not present")
    T = TOOLS_A | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_STILL_REF       # (still referenced: the reference finishes the motion of such pictures only, DecLibRecon.cpp:1080-1100)
    for W, H, seed, kw in ((416, 240, 301, dict(p_bi=0.9, p_intra=0.05, mv_sigma=2.0)), (512, 384, 302, dict(p_bi=0.8, p_affine=0.2, log2_ctu=6, mv_sigma=3.0))):
        plans, nslots = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
        rec = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=2, log2_ctu=kw.get("log2_ctu", 7), host_threads=2)
        dpb = {}
        refined = 0
        for k, pl in enumerate(plans):
            d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=T, **kw)
            refs = {slot: dpb[slot] for lst in pl.ref_slots for (slot, _) in lst}
            planes, motion = refdrv.reconstruct_with_motion(d, refs, flags=0)
            want = motion.reshape(d.h4, d.w4)[::2, ::2].reshape(-1)
            d.hdr.tool_flags |= abi.TOOL_COL_MOTION
            if k & 1:
                hnd = rec.prepare(d)
                job = rec.submit_prepared(hnd)
            else:
                job = rec.decompress_picture(d)
            got = rec.read_col_motion(job)
            if k & 1:
                rec.free_prepared(hnd)
            assert len(got) == len(want)
            assert np.array_equal(got["mv"], want["mv"]) and np.array_equal(got["ref_idx"], want["ref_idx"]), "POC %d: %d records differ" % (pl.poc, int((got["mv"] != want["mv"]).any(axis=(1, 2)).sum()))
            refined += int((want["mv"] != d.motion.reshape(d.h4, d.w4)[::2, ::2].reshape(-1)["mv"]).any(axis=(1, 2)).sum())
            got_planes = rec.read_picture(pl.slot)
            assert all(np.array_equal(g, w) for g, w in zip(got_planes, planes))
            dpb[pl.slot] = planes
        rec.close()
        assert refined > 0          # DMVR moved something: the test is about the refined vectors


def test_unsupported_tools_fail_loudly(built):
    import vvdec_amd
    rec = vvdec_amd.Reconstructor(128, 64, num_slots=2)
    p = synth.default_params(width=128, height=64, seed=1, tool_flags=0, slice_type=abi.SLICE_I)
    d = synth.generate(p)
    d.cu["pred_mode"][0] = abi.PRED_IBC                    # intra block copy: not reconstructed by this build
    with pytest.raises(vvdec_amd.VvrError):
        rec.decompress_picture(d)
    d = synth.generate(p)
    d.cu["tree"][0] = abi.TREE_LUMA                        # a luma-tree CU whose TUs still carry chroma: inconsistent description
    with pytest.raises(vvdec_amd.VvrError):
        rec.decompress_picture(d)
    rec.close()


def test_output_window_and_picture_hash(built):
    """the output side of the boundary on reconstructed pictures: conformance-window crop (vvr_read_output) and the decoded-picture-hash
    digests (vvr_picture_hash: MD5, CRC, checksum - k_plane_hash_rows on the device) against the restatement of PicYuvMD5.cpp in
    tests/refdrv.py (pinned against the reference's functions on the CPU) over the full planes read back through vvr_read_plane"""
    import vvdec_amd
    W, H = 256, 128
    plans, nslots = stream.ra_plan(3, gop=2, seed_poc0_is_external=False)
    rec = vvdec_amd.Reconstructor(W, H, num_slots=nslots, num_streams=2)
    for pl in plans:
        rec.decompress_picture(synth.picture_for_plan(pl, W, H, seed=281, tool_flags=TOOLS_A, p_intra=0.2))
    rec.sync()
    for pl in plans:
        full = rec.read_picture(pl.slot)
        win = rec.read_output(pl.slot, window=(16, 8, 200, 96))
        for c in range(3):
            s = 1 if c else 0
            assert np.array_equal(win[c], full[c][8 >> s:(8 + 96) >> s, 16 >> s:(16 + 200) >> s])
        for method in (0, 1, 2):
            assert rec.picture_hash(pl.slot, method) == refdrv.picture_hash(full, 10, method), "hash method %d" % method
    rec.close()


@pytest.mark.parametrize("bd,cf,W,H", [(10, 1, 200, 136), (8, 1, 200, 136), (10, 0, 200, 136), (8, 0, 136, 72), (10, 1, 584, 328)])
def test_output_stage_on_the_device(built, bd, cf, W, H):
    """f3 on the device, values not lengths: MD5 / CRC / checksum of vvr_picture_hash (CommonLib/PicYuvMD5.cpp:99-221) and the 8-bit narrowing of
    vvr_read_output (VVDecImpl::copyComp, vvdecimpl.cpp:818-880) for 8- and 10-bit samples, 4:2:0 and 4:0:0, picture sizes that are no multiple
    of anything (and one beyond 256 in both directions: the checksum mask folds x >> 8 and y >> 8), on a reconstructed picture and on random planes"""
    import vvdec_amd
    ncomp = 3 if cf else 1
    geo = dict(bit_depth=bd, chroma_format=cf, log2_ctu=6)
    rec = vvdec_amd.Reconstructor(W, H, num_slots=3, num_streams=1, **geo)
    rng = np.random.default_rng(bd * 7 + cf + W)
    rnd = [rng.integers(0, 1 << bd, rec.plane_shape(c)).astype(np.uint16) for c in range(ncomp)]
    rec.write_picture(1, rnd)
    plans, _ = stream.ra_plan(1, gop=1, seed_poc0_is_external=False)
    d = synth.picture_for_plan(plans[0], W, H, seed=977, tool_flags=TOOLS_A, **geo)
    rec.wait(rec.decompress_picture(d))
    for slot, planes in ((1, rnd), (plans[0].slot, rec.read_picture(plans[0].slot)[:ncomp])):
        for method in (0, 1, 2):
            got = rec.picture_hash(slot, method)
            assert got == refdrv.picture_hash(planes, bd, method), "slot %d hash method %d: %r" % (slot, method, got)
        wx, wy, ww, wh = 8, 4, W - 24, H - 12
        win16 = rec.read_output(slot, window=(wx, wy, ww, wh))
        for c in range(ncomp):
            s = 1 if c else 0
            assert np.array_equal(win16[c], planes[c][wy >> s:(wy + wh) >> s, wx >> s:(wx + ww) >> s])
        if bd == 8:
            win8 = rec.read_output(slot, window=(wx, wy, ww, wh), bytes_per_sample=1)
            for c in range(ncomp):
                s = 1 if c else 0
                assert win8[c].dtype == np.uint8 and np.array_equal(win8[c], planes[c][wy >> s:(wy + wh) >> s, wx >> s:(wx + ww) >> s].astype(np.uint8))
        else:
            with pytest.raises(vvdec_amd.VvrError):
                rec.read_output(slot, window=(wx, wy, ww, wh), bytes_per_sample=1)        # only 8-bit content is narrowed
    rec.close()


def _golden_files():
    import glob, os
    return sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


@pytest.mark.parametrize("path", _golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_fixtures_reference_outputs(built, path):
    """HIP path against what the REAL reference classes produced (tests/golden, written by make_golden.py), every stage."""
    import vvdec_amd
    import golden_io
    d, refs, outs = golden_io.load(path)
    h = d.hdr
    nslots = max([h.out_slot] + list(refs.keys())) + 1
    for st, stop in (("final", abi.STOP_NONE), ("reco", abi.STOP_RECO), ("dbk", abi.STOP_DEBLOCK), ("sao", abi.STOP_SAO)):
        # (a fixture with scaled reference pictures holds them at their own sizes: the context is as large as the largest picture)
        MW = max([h.width] + [r[0].shape[1] for r in refs.values()]); MH = max([h.height] + [r[0].shape[0] for r in refs.values()])
        rec = vvdec_amd.Reconstructor(MW, MH, bit_depth=h.bit_depth, log2_ctu=h.log2_ctu, num_slots=nslots, num_streams=1, stop_after=stop)
        for slot, planes in refs.items():
            full = [np.zeros(rec.plane_shape(c), np.uint16) for c in range(3)]
            for c in range(3):
                full[c][:planes[c].shape[0], :planes[c].shape[1]] = planes[c]
            rec.write_picture(slot, full)
        rec.wait(rec.decompress_picture(d))
        got = [p[:h.height >> (1 if c else 0), :h.width >> (1 if c else 0)] for c, p in enumerate(rec.read_picture(h.out_slot))]
        rec.close()
        for c in range(3):
            assert np.array_equal(got[c], outs[st][c]), "%s comp %d: %d samples differ from the reference decoder" % (st, c, int((got[c] != outs[st][c]).sum()))


def test_copy_kernel_bandwidth(built):
    """the measured HBM ceiling bench.py reports next to the nominal 8 TB/s: the library's copy kernel over the DPB (16 slots -> 8 lanes of scratch
    planes: 400 MB each way, beyond the caches)"""
    import vvdec_amd
    rec = vvdec_amd.Reconstructor(3840, 2160, num_slots=16, num_streams=8)
    pic = synth.natural_picture(3840, 2160, 5)
    rec.write_picture(0, pic)
    bps = rec.copy_bandwidth(10)
    assert 1.0e12 < bps < 8.0e12, bps
    got = rec.read_picture(0)
    assert all(np.array_equal(g, w) for g, w in zip(got, pic))          # slot 0 is only read
    rec.close()


@pytest.mark.parametrize("idx,seed,tools_extra,kw", [
    (0, 701, abi.TOOL_LMCS, dict(p_cclm=0.3, p_mip=0.2, p_isp=0.2)),
    (2, 702, abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_STILL_REF | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_intra=0.2, p_bi=0.9, p_affine=0.15, p_sbtmvp=0.1, p_ciip=0.1, p_geo=0.1)),
    (3, 703, abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_STILL_REF, dict(p_intra=0.05, p_bi=0.95, mv_sigma=2.0)),
])
def test_declibrecon_replacement_executed(built, idx, seed, tools_extra, kw):
    """SURVEY 8(f)-1 executed: reference-built objects of a picture through integration/DecLibReconAmd.h (LF_INIT by the reference's own
    calcFilterStrengthsCTU, extractor, vvr_submit, vvr_wait, DMVR delta MVs from the GPU back through DecCu::TaskFinishMotionInfo) - planes and
    motion field equal what the reference's own DecLibRecon stages produce for the same objects"""
    import os
    import vvdec_amd
    if not (refdrv.available() and refdrv.binding_available()):
        pytest.skip("the reference build (oracle/_ref) is not present")
    W, H = 384, 256
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=TOOLS | abi.TOOL_LFNST | tools_extra, **kw)
    refs = {slot: synth.natural_picture(W, H, seed + 100 + poc) for lst in pl.ref_slots for (slot, poc) in lst}
    want_planes, want_motion = refdrv.reconstruct_with_motion(d, refs, flags=refdrv.DERIVE_LFP)
    got_planes, got_motion = refdrv.run_binding(d, refs, vvdec_amd._LIBPATH)
    for c in range(3):
        assert np.array_equal(got_planes[c], want_planes[c]), "comp %d: %d samples differ from the reference's DecLibRecon" % (c, int((got_planes[c] != want_planes[c]).sum()))
    valid = want_motion["ref_idx"] >= 0
    assert np.array_equal(got_motion["ref_idx"], want_motion["ref_idx"])
    assert np.array_equal(got_motion["mv"][valid], want_motion["mv"][valid]), "motion field after TaskFinishMotionInfo differs"
    if idx == 3:
        both = valid & (d.motion["ref_idx"] >= 0)
        assert getattr(d, "num_dmvr", 0) > 0 and not np.array_equal(want_motion["mv"][both], d.motion["mv"][both]), "no DMVR refinement in this picture: the test would be vacuous"


@pytest.mark.parametrize("idx,seed,tools_extra,kw,threads", [
    (0, 711, abi.TOOL_LMCS, dict(p_cclm=0.3, p_mip=0.2, p_isp=0.2), 0),
    (2, 712, abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_STILL_REF | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_intra=0.2, p_bi=0.9, p_affine=0.15, p_sbtmvp=0.1, p_ciip=0.1, p_geo=0.1), 3),
    (3, 713, abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_STILL_REF, dict(p_intra=0.05, p_bi=0.95, mv_sigma=2.0), 2),
])
def test_dropin_declibrecon(built, idx, seed, tools_extra, kw, threads):
    """the DROP-IN executed on the GPU: the reference's class vvdec::DecLibRecon with the member functions of integration/DecLibReconDropIn.cpp, driven
    like DecLib::reconPicture drives it (create( ThreadPool*, id, upscale ) / decompressPicture / waitForPrevDecompressedPic, pool of 0 / 2 / 3
    threads; reference pictures that did not come out of this back-end are uploaded from their Picture buffers).  The planes AS THEY SIT IN THE
    PICTURE'S OWN BUFFERS afterwards - where output, hash check and film grain read them - and the motion field after TaskFinishMotionInfo equal
    what the reference's own DecLibRecon stages produce for the same objects"""
    import vvdec_amd
    if not (refdrv.available() and refdrv.dropin_available()):
        pytest.skip("the reference build (oracle/_ref) is not present")
    W, H = 384, 256
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=TOOLS | abi.TOOL_LFNST | tools_extra, **kw)
    refs = {slot: synth.natural_picture(W, H, seed + 100 + poc) for lst in pl.ref_slots for (slot, poc) in lst}
    want_planes, want_motion = refdrv.reconstruct_with_motion(d, refs, flags=refdrv.DERIVE_LFP)
    got_planes, got_motion = refdrv.run_dropin(d, refs, vvdec_amd._LIBPATH, threads=threads)
    for c in range(3):
        assert np.array_equal(got_planes[c], want_planes[c]), "comp %d: %d samples differ from the reference's DecLibRecon" % (c, int((got_planes[c] != want_planes[c]).sum()))
    valid = want_motion["ref_idx"] >= 0
    assert np.array_equal(got_motion["ref_idx"], want_motion["ref_idx"])
    assert np.array_equal(got_motion["mv"][valid], want_motion["mv"][valid]), "motion field after TaskFinishMotionInfo differs"


@pytest.mark.parametrize("idx,seed,tools_extra,kw,threads,rotate", [
    (0, 721, abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_SCALING_LIST, dict(num_slices=3, p_cclm=0.3, p_mip=0.2, p_coded=0.8), 2, False),
    (2, 722, abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_WP, dict(num_slices=4, p_intra=0.2, p_bi=0.9, p_affine=0.15, p_sbtmvp=0.1, p_ciip=0.1, p_geo=0.15), 3, True),
    (3, 723, abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_SCALING_LIST | abi.TOOL_NO_LF_ACROSS_TILES, dict(num_slices=4, tile_cols=2, tile_rows=2, p_intra=0.1, p_bi=0.9, p_geo=0.1, mv_sigma=2.0), 2, True),
    (2, 724, abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_WP, dict(num_slices=4, intra_slices=0b0101, p_intra=0.2, p_bi=0.9, p_cclm=0.3), 2, True),      # I slices among B slices
])
def test_dropin_slices_with_headers_of_their_own(built, idx, seed, tools_extra, kw, threads, rotate):
    """the drop-in on pictures whose slices differ in their headers and (rotate) in their reference picture lists: the reference's objects carry
    per-slice lists / reference indices / weights, the extractor merges them (vvr_picture::slices, union of the lists), the GPU reconstructs,
    and the Picture's buffers and motion field equal what the reference's own DecLibRecon stages give for the same objects"""
    import vvdec_amd
    if not (refdrv.available() and refdrv.dropin_available()):
        pytest.skip("the reference build (oracle/_ref) is not present")
    W, H = 512, 384
    plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    pl = plans[idx]
    d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=TOOLS | abi.TOOL_LFNST | tools_extra, log2_ctu=6, **kw)
    synth.vary_slices(d, seed, intra_slices=kw.get("intra_slices", 0))
    refs = {slot: synth.natural_picture(W, H, seed + 100 + poc) for lst in pl.ref_slots for (slot, poc) in lst}
    fl = refdrv.ROTATE_REF_LISTS if rotate else 0
    want_planes, want_motion = refdrv.reconstruct_with_motion(d, refs, flags=refdrv.DERIVE_LFP | fl)
    got_planes, got_motion = refdrv.run_dropin(d, refs, vvdec_amd._LIBPATH, threads=threads, flags=fl)
    for c in range(3):
        assert np.array_equal(got_planes[c], want_planes[c]), "comp %d: %d samples differ from the reference's DecLibRecon" % (c, int((got_planes[c] != want_planes[c]).sum()))
    valid = want_motion["ref_idx"] >= 0
    assert np.array_equal(got_motion["ref_idx"], want_motion["ref_idx"])
    assert np.array_equal(got_motion["mv"][valid], want_motion["mv"][valid]), "motion field after TaskFinishMotionInfo differs"
    # and the rotation is immaterial: the same picture as without it
    if rotate:
        plain, _ = refdrv.reconstruct_with_motion(d, refs, flags=refdrv.DERIVE_LFP)
        assert all(np.array_equal(a, b) for a, b in zip(plain, want_planes))


@pytest.mark.parametrize("threads", [0, 3])
def test_reference_slots_replicated_between_two_back_ends_on_the_device(built, threads):
    """the device-side ordering the picture-level multi-GPU split rests on (vvr_stream_wait_job / vvr_stream_wait_slot / vvr_slot_external_event with a
    torch stream and torch events, vvdec_amd.parallel.TorchDeviceRuntime), on ONE device: tests/two_back_ends_on_one_device.py, in a process of its
    own (torch brings its own HIP runtime, which has to be the first one the process initialises - the order bench.py uses)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "two_back_ends_on_one_device.py"), str(threads)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "both DPBs equal the single back-end" in r.stdout


def _rpr_ratio(ref, cur):
    return ((ref << 14) + (cur >> 1)) // cur          # CU::getRprScaling (UnitTools.cpp:101)


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,kw", [
    # size of the picture with POC k: the stream changes its resolution at POC 4 and goes back at POC 1 / 3
    ({0: (512, 384), 4: (384, 256), 2: (384, 256), 1: (512, 384), 3: (384, 256)}, dict(log2_ctu=6, p_intra=0.2)),
    ({0: (320, 192), 4: (512, 384), 2: (384, 256), 1: (320, 192), 3: (512, 384)}, dict(log2_ctu=5, p_intra=0.15, p_affine=0.3, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.2, p_bcw=0.4, p_imv_hpel=0.3)),
    ({0: (512, 384), 4: (256, 192), 2: (512, 384), 1: (256, 192), 3: (384, 288)}, dict(log2_ctu=7, p_intra=0.2, p_affine=0.3, p_sbtmvp=0.1, tool_flags_extra=abi.TOOL_WP | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE)),
])
def test_stream_that_changes_its_picture_size(built, sizes, kw):
    """Reference picture resampling: pictures of several sizes in one context (each in the top left corner of its DPB slot, every stage at the
    picture's own size), CUs predicting from reference pictures of another size through k_mc_rpr (all three filter sets, both directions of
    scaling), from same-size pictures through the ordinary kernels; the decoded picture hash covers the picture, not the slot."""
    import ctypes as C
    import vvdec_amd
    kw = dict(kw)
    tools = TOOLS_A | kw.pop("tool_flags_extra", 0)
    l2 = kw.pop("log2_ctu")
    MW, MH = max(s[0] for s in sizes.values()), max(s[1] for s in sizes.values())
    plans, nslots = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
    rec = vvdec_amd.Reconstructor(MW, MH, num_slots=nslots, num_streams=2, host_threads=2, log2_ctu=l2)
    size_of_slot, cpu, descs = {}, {}, []
    for pl in plans:
        W, H = sizes[pl.poc]
        masks, spec = [0, 0], {}
        for l, lst in enumerate(pl.ref_slots):
            for i, (slot, poc) in enumerate(lst):
                rw, rh = sizes[poc]
                if (rw, rh) != (W, H):
                    masks[l] |= 1 << i
                    spec[(l, i)] = dict(ratio=(_rpr_ratio(rw, W), _rpr_ratio(rh, H)), size=(rw, rh))
        d = synth.picture_for_plan(pl, W, H, seed=900 + len(sizes), tool_flags=tools, log2_ctu=l2, alloc=rec.host_array, scaled_refs=(C.c_uint16 * 2)(*masks), **kw)
        if pl.slice_type != abi.SLICE_I and (spec or pl.poc == 3):          # (POC 3: a table that names no scaled picture)
            synth.attach_rpr(d, spec)
        descs.append(d)
    assert any(d.rpr is not None and any(d.rpr.ref[l][i].ratio[0] > (1 << 14) * 5 // 4 for l in range(2) for i in range(d.hdr.num_ref[l])) for d in descs)
    for pl, d in zip(plans, descs):
        W, H = sizes[pl.poc]
        job = rec.decompress_picture(d)
        rec.wait(job)
        got = [p[:H >> (1 if c else 0), :W >> (1 if c else 0)] for c, p in enumerate(rec.read_picture(pl.slot))]
        want = refdrv.oracle_reconstruct(d, cpu)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), "POC %d (%dx%d) comp %d: %d samples differ" % (pl.poc, W, H, c, int((got[c] != want[c]).sum()))
        cpu[pl.slot] = want
        for method in (1, 2):
            assert rec.picture_hash(pl.slot, method) == refdrv.picture_hash(want, 10, method), "hash method %d of POC %d" % (method, pl.poc)
        # the picture back the way the drop-in takes it (vvr_read_picture: this picture only, pinned staging, rows at the caller's strides)
        for a, b in zip(rec.read_picture_into(pl.slot, (W, H), pad=24, threads=3), want):
            assert np.array_equal(a, b)
    # all in flight at once
    rec2 = vvdec_amd.Reconstructor(MW, MH, num_slots=nslots, num_streams=3, host_threads=3, log2_ctu=l2)
    for d in descs:
        rec2.decompress_picture(d)
    rec2.sync()
    last = {}
    for pl in plans:
        last[pl.slot] = pl
    for slot in last:
        for a, b in zip(rec.read_picture(slot), rec2.read_picture(slot)):
            assert np.array_equal(a, b)
    rec.close()
    rec2.close()


@pytest.mark.gpu
def test_scaled_reference_pictures_through_the_back_end(built):
    """the cases the oracle is pinned with against the reference's xPredInterBlkRPR (test_oracle_vs_ref.py: scaling windows with offsets, ratios that
    differ by direction, chroma sample locations, 8 bit, 4:0:0): one picture each, the reference pictures uploaded at their own sizes"""
    import vvdec_amd
    from test_oracle_vs_ref import RPR_CASES, rpr_case, ALL
    for (W, H, l2, idx, seed, specs, win, colloc, kw) in RPR_CASES:
        kw = dict(kw)
        tools = ALL | kw.pop("tool_flags_extra", 0)
        d, refs = rpr_case(W, H, l2, idx, seed, specs, win=win, colloc=colloc, tools=tools, **kw)
        bd, cf = d.hdr.bit_depth, d.hdr.chroma_format
        MW = max([W] + [r[0].shape[1] for r in refs.values()])
        MH = max([H] + [r[0].shape[0] for r in refs.values()])
        rec = vvdec_amd.Reconstructor(MW, MH, num_slots=max(list(refs) + [d.hdr.out_slot]) + 1, num_streams=1, log2_ctu=l2, bit_depth=bd, chroma_format=cf)
        for slot, planes in refs.items():
            full = []
            for c, p in enumerate(planes):
                a = np.zeros(rec.plane_shape(c), np.uint16)
                a[:p.shape[0], :p.shape[1]] = p
                full.append(a)
            rec.write_picture(slot, full)
        rec.wait(rec.decompress_picture(d))
        got = rec.read_picture(d.hdr.out_slot)
        want = refdrv.oracle_reconstruct(d, refs)
        for c in range(len(want)):
            g = got[c][:want[c].shape[0], :want[c].shape[1]]
            assert np.array_equal(g, want[c]), "seed %d comp %d: %d samples differ" % (seed, c, int((g != want[c]).sum()))
        rec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 3])
def test_dropin_with_scaled_reference_pictures(built, threads):
    """the drop-in on pictures whose reference pictures have another size / scaling window: the reference's objects (PPSs with scaling windows,
    Slice::m_scalingRatio, reference Pictures of their own size) are flattened by the extractor (vvr_picture.rpr), the reference pictures are
    uploaded at their own size, the GPU reconstructs, and the Picture's buffers equal what the reference's own DecLibRecon stages give"""
    import vvdec_amd
    from test_oracle_vs_ref import RPR_CASES, rpr_case, ALL
    if not (refdrv.available() and refdrv.dropin_available()):
        pytest.skip("the reference build (oracle/_ref) is not present")
    for (W, H, l2, idx, seed, specs, win, colloc, kw) in RPR_CASES:
        kw = dict(kw)
        tools = ALL | kw.pop("tool_flags_extra", 0)
        d, refs = rpr_case(W, H, l2, idx, seed, specs, win=win, colloc=colloc, tools=tools, **kw)
        want_planes, want_motion = refdrv.reconstruct_with_motion(d, refs, flags=refdrv.DERIVE_LFP)
        got_planes, got_motion = refdrv.run_dropin(d, refs, vvdec_amd._LIBPATH, threads=threads)
        for c in range(len(want_planes)):
            assert np.array_equal(got_planes[c], want_planes[c]), "seed %d comp %d: %d samples differ from the reference's DecLibRecon" % (seed, c, int((got_planes[c] != want_planes[c]).sum()))
        valid = want_motion["ref_idx"] >= 0
        assert np.array_equal(got_motion["ref_idx"], want_motion["ref_idx"]) and np.array_equal(got_motion["mv"][valid], want_motion["mv"][valid])
