#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz with the REAL reference decoder classes (oracle/_ref/libvvref.so).

Run in the build container (needs /root/reference for `make -C oracle harness`):   python tests/golden/make_golden.py
Each fixture = one synthetic pre-parsed picture (tools/synth.cpp, fixed seed) + its reference pictures + the planes the
reference's DecCu / InterPrediction / IntraPrediction / LoopFilter / SampleAdaptiveOffset / AdaptiveLoopFilter produced
after each stage (scalar code paths; the SIMD paths are checked to give the same bytes before anything is written)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np
import refdrv
import golden_io
from vvdec_amd import abi, synth, stream

ALL = abi.TOOL_SAO_LUMA | abi.TOOL_SAO_CHROMA | abi.TOOL_ALF | abi.TOOL_CCALF | abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST

# name, W, H, log2_ctu, picture index in a GOP-4 stream with an I picture at POC 0, seed, tools, generator overrides
CASES = [
    ("i_256x128_ctu128", 256, 128, 7, 0, 11, ALL, {}),
    ("i_200x136_ctu64", 200, 136, 6, 0, 12, ALL, {}),
    ("i_128x64_ctu32_nodq", 128, 64, 5, 0, 13, ALL & ~abi.TOOL_DEP_QUANT, {}),
    ("b_256x128_ctu128", 256, 128, 7, 2, 14, ALL, dict(p_intra=0.2)),
    ("b_200x136_ctu64_inter", 200, 136, 6, 3, 15, ALL, dict(p_intra=0.0)),
    ("b_256x192_ctu128_key", 256, 192, 7, 1, 16, ALL, dict(p_intra=0.3)),
    ("b_128x128_ctu32_nofilters", 128, 128, 5, 4, 17, abi.TOOL_DEP_QUANT | abi.TOOL_MTS | abi.TOOL_LFNST, dict(p_intra=0.15)),
    ("b_256x128_ctu128_bdof", 256, 128, 7, 2, 18, ALL | abi.TOOL_BDOF, dict(p_intra=0.1, p_bi=0.9)),
    ("b_200x136_ctu64_bdof", 200, 136, 6, 3, 19, ALL | abi.TOOL_BDOF, dict(p_intra=0.0, p_bi=0.8, mv_sigma=2.0)),
    ("b_256x128_ctu128_dmvr_bdof", 256, 128, 7, 2, 20, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR, dict(p_intra=0.1, p_bi=0.9)),
    ("b_200x136_ctu64_dmvr", 200, 136, 6, 3, 21, ALL | abi.TOOL_DMVR, dict(p_intra=0.0, p_bi=0.8, mv_sigma=2.0)),
    ("b_256x128_ctu128_affine_prof", 256, 128, 7, 2, 22, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF, dict(p_intra=0.1, p_affine=0.5)),
    ("b_200x136_ctu64_affine", 200, 136, 6, 1, 23, ALL, dict(p_intra=0.0, p_affine=0.5, mv_sigma=3.0)),
    ("b_256x128_ctu128_gpm", 256, 128, 7, 2, 24, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF, dict(p_intra=0.1, p_geo=0.5, p_affine=0.1)),
    ("b_256x128_ctu64_ciip", 256, 128, 6, 3, 25, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF, dict(p_intra=0.25, p_ciip=0.5, p_coded=0.6)),
    ("b_256x128_ctu64_jccr_sign", 256, 128, 6, 2, 27, ALL | abi.TOOL_JCCR_SIGN, dict(p_intra=0.3, p_jccr=0.7, p_coded_chroma=0.7)),
    ("i_256x128_ctu128_cclm", 256, 128, 7, 0, 28, ALL, dict(p_cclm=0.5)),
    ("b_200x136_ctu64_cclm", 200, 136, 6, 2, 29, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR, dict(p_cclm=0.6, p_intra=0.5)),
    ("i_256x128_ctu128_mip", 256, 128, 7, 0, 30, ALL, dict(p_mip=0.5, p_cclm=0.2, p_lfnst=0.4)),
    ("b_256x128_ctu128_lmcs", 256, 128, 7, 2, 31, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_LMCS, dict(p_intra=0.3, p_cclm=0.2, p_mip=0.2, p_ciip=0.1, p_affine=0.1)),
    ("b_256x192_ctu128_lmcs_cscale", 256, 192, 7, 3, 33, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_intra=0.3, p_cclm=0.2, p_ciip=0.1, p_coded_chroma=0.5, p_jccr=0.3)),
    ("i_200x136_ctu64_lmcs", 200, 136, 6, 0, 32, ALL | abi.TOOL_LMCS, dict(p_cclm=0.3, p_mip=0.3)),
    ("b_256x128_ctu64_sbt", 256, 128, 6, 2, 34, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_intra=0.1, p_sbt=0.6, p_coded_chroma=0.5, p_jccr=0.2, p_affine=0.1)),
    ("b_256x128_ctu64_wp", 256, 128, 6, 2, 35, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_WP, dict(p_intra=0.1, p_affine=0.2, p_sbtmvp=0.15, p_ciip=0.1, p_geo=0.1, p_bcw=0.2)),
    ("b_256x192_ctu128_scaling_list", 256, 192, 7, 2, 36, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_SCALING_LIST, dict(p_intra=0.3, p_coded=0.8, p_coded_chroma=0.6, p_sbt=0.2, p_jccr=0.2, p_lfnst=0.4)),
    ("i_256x128_ctu128_isp", 256, 128, 7, 0, 37, ALL, dict(p_isp=0.6, p_lfnst=0.4, p_coded=0.7, p_cclm=0.2)),
    ("b_200x136_ctu64_isp_lmcs", 200, 136, 6, 2, 38, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_isp=0.6, p_intra=0.5, p_cclm=0.3, p_jccr=0.3, p_coded_chroma=0.6)),
    ("i_256x128_ctu128_cclm_colloc", 256, 128, 7, 0, 39, ALL | abi.TOOL_CCLM_COLLOC, dict(p_cclm=0.6)),
    ("i_256x192_ctu128_dual_tree", 256, 192, 7, 0, 40, ALL | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(dual_tree=1.0, p_cclm=0.3, p_lfnst=0.4, p_isp=0.2, p_mip=0.2, p_coded_chroma=0.6, p_jccr=0.2)),
    ("i_200x136_ctu32_dual_tree", 200, 136, 5, 0, 41, ALL, dict(dual_tree=1.0, p_cclm=0.4, p_coded_chroma=0.6)),
    ("i_256x128_ctu64_dual_tree_4xn_implicit_mts", 256, 128, 6, 0, 42, ALL | abi.TOOL_IMPLICIT_MTS, dict(dual_tree=2.0, p_cclm=0.3, p_lfnst=0.4, p_isp=0.2, p_mip=0.3, p_coded=0.7, p_coded_chroma=0.5, p_split_scale=1.5)),
    ("i_256x128_ctu64_isp_4xn", 256, 128, 6, 0, 43, ALL | abi.TOOL_IMPLICIT_MTS, dict(dual_tree=3.0, p_isp=0.7, p_lfnst=0.3, p_coded=0.8, p_split_scale=1.8)),
    ("b_256x128_ctu64_small_cus", 256, 128, 6, 2, 44, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(min_cu_log2=2, p_intra=0.3, p_split_scale=1.8, p_cclm=0.3, p_isp=0.2, p_sbt=0.2, p_coded_chroma=0.5)),
    ("i_200x136_ctu32_small_cus", 200, 136, 5, 0, 45, ALL, dict(min_cu_log2=2, p_split_scale=1.8, p_cclm=0.3, p_mip=0.2, p_lfnst=0.3)),
    ("b_384x256_ctu128_subblock_edges", 384, 256, 7, 1, 46, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF, dict(p_intra=0.05, p_affine=0.5, p_sbtmvp=0.2, p_split_scale=0.5, p_coded=0.5)),
    ("i_256x128_ctu64_ibc_lmcs", 256, 128, 6, 0, 47, ALL | abi.TOOL_IBC | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(p_ibc=0.5, p_split_scale=1.4, p_cclm=0.3, p_jccr=0.3, p_coded_chroma=0.5)),
    ("b_264x200_ctu128_ibc_small_cus", 264, 200, 7, 3, 48, ALL | abi.TOOL_IBC | abi.TOOL_BDOF | abi.TOOL_DMVR, dict(p_ibc=0.6, p_intra=0.4, min_cu_log2=2, p_split_scale=1.6, p_ciip=0.1)),
    ("b_256x128_ctu64_8bit", 256, 128, 6, 2, 49, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE, dict(bit_depth=8, p_intra=0.25, p_affine=0.15, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.1, p_mip=0.2, p_isp=0.2, p_cclm=0.3, p_jccr=0.2, p_coded_chroma=0.5)),
    ("i_384x256_ctu64_slices", 384, 256, 6, 0, 51, ALL | abi.TOOL_NO_LF_ACROSS_SLICES, dict(num_slices=3, p_cclm=0.3, p_mip=0.2)),
    ("b_384x256_ctu64_tiles_slices", 384, 256, 6, 2, 52, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_NO_LF_ACROSS_SLICES | abi.TOOL_NO_LF_ACROSS_TILES,
     dict(num_slices=3, tile_cols=2, tile_rows=2, p_intra=0.3, p_cclm=0.3, p_ciip=0.1, p_affine=0.1, p_coded_chroma=0.5)),
    ("b_384x256_ctu64_subpictures", 384, 256, 6, 2, 55, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_LMCS, dict(subpics=1 | (2 << 1) | (2 << 3), tile_cols=2, tile_rows=2, p_intra=0.15, p_affine=0.2, p_bi=0.8, mv_sigma=24.0)),
    ("b_384x256_ctu64_wrap_around", 384, 256, 6, 3, 54, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF, dict(wrap_offset=384, p_intra=0.1, p_affine=0.3, p_bi=0.8, p_geo=0.1, mv_sigma=10.0)),
    ("b_384x256_ctu64_virtual_boundaries", 384, 256, 6, 2, 53, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_LMCS, dict(virtual_boundaries=2 | (2 << 2) | 16, p_intra=0.3, p_cclm=0.3, p_affine=0.2, p_coded_chroma=0.5)),
    ("b_256x128_ctu64_ladf", 256, 128, 6, 2, 50, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_LADF, dict(p_intra=0.3, p_affine=0.1, p_coded=0.5)),
    ("b_384x256_ctu64_slice_headers", 384, 256, 6, 2, 56, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_LMCS | abi.TOOL_LMCS_CSCALE | abi.TOOL_SCALING_LIST | abi.TOOL_WP | abi.TOOL_NO_LF_ACROSS_SLICES,
     dict(num_slices=4, vary_slices=1, p_intra=0.3, p_cclm=0.3, p_ciip=0.1, p_affine=0.1, p_coded=0.8, p_coded_chroma=0.5)),
    # reference picture resampling: two scaled reference pictures of their own sizes (1.3 x 0.8 and 1.3 x 1.8: regular and both low-pass filter sets), scaling
    # windows with offsets, chroma samples not collocated, weighted prediction, affine / GPM / CIIP / SbTMVP CUs among the ones that read them
    ("b_400x208_ctu64_scaled_references", 400, 208, 6, 2, 57, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF | abi.TOOL_WP,
     dict(rpr=dict(specs=[dict(ratio=(int(1.3 * 16384), int(0.8 * 16384)), size=(520, 168), win=(16, 6)), dict(ratio=(int(1.3 * 16384), int(1.8 * 16384)), size=(512, 376), win=(-8, 2))], win=(8, 4), colloc=(0, 0)),
          p_intra=0.15, p_affine=0.3, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.2, mv_sigma=6.0)),
    ("b_256x192_ctu128_all_inter", 256, 192, 7, 2, 26, ALL | abi.TOOL_BDOF | abi.TOOL_DMVR | abi.TOOL_PROF, dict(p_intra=0.1, p_affine=0.2, p_geo=0.1, p_ciip=0.1, p_sbtmvp=0.2)),
]


def main():
    assert refdrv.available(), "oracle/_ref is not built (make -C oracle harness needs /root/reference)"
    only = sys.argv[1:]
    for (name, W, H, l2, idx, seed, tools, kw) in CASES:
        if only and name not in only:
            continue
        plans, _ = stream.ra_plan(5, gop=4, seed_poc0_is_external=False)
        pl = plans[idx]
        kw = dict(kw)
        vary = kw.pop("vary_slices", 0)
        rpr = kw.pop("rpr", None)
        if rpr:
            from test_oracle_vs_ref import rpr_case      # (description with its vvr_rpr_params, reference pictures of their own sizes)
            d, refs = rpr_case(W, H, l2, idx, seed, rpr["specs"], win=rpr["win"], colloc=rpr["colloc"], tools=tools, **kw)
        else:
            d = synth.picture_for_plan(pl, W, H, seed=seed, tool_flags=tools, log2_ctu=l2, **kw)
            if vary:
                synth.vary_slices(d, seed)           # slices with headers of their own (vvr_slice_header)
            refs = {}
            for lst in pl.ref_slots:
                for (slot, poc) in lst:
                    refs.setdefault(slot, synth.natural_picture(W, H, seed + 100 + poc, bit_depth=kw.get("bit_depth", 10)))
        outs = {}
        for st, fl in (("reco", refdrv.STOP_AFTER_RECO), ("dbk", refdrv.STOP_AFTER_DBK), ("sao", refdrv.STOP_AFTER_SAO), ("final", 0)):
            scalar = refdrv.reconstruct(d, refs, flags=fl)["planes"]
            simd = refdrv.reconstruct(d, refs, flags=fl | refdrv.SIMD)["planes"]
            assert all(np.array_equal(a, b) for a, b in zip(scalar, simd)), "%s/%s: the reference's scalar and SIMD paths disagree" % (name, st)
            outs[st] = scalar
        path = os.path.join(HERE, name + ".npz")
        golden_io.save(path, d, refs, outs)
        print("%-32s %6d CUs %6d TUs  %7.1f KiB" % (name, len(d.cu), len(d.tu), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
