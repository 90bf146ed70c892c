#!/usr/bin/env python3
"""mini_vvenc.py - a deliberately small VVC (H.266) bitstream WRITER: TEST INFRASTRUCTURE, not an encoder anyone would ship.

Why it exists: no VVC bitstream is available offline and no encoder is installed, so until now no picture that the reference's own PARSER
(DecLibParser / HLSyntaxReader / CABACReader) produced had ever gone through the drop-in reconstruction stage.  This tool writes conforming Annex-B
streams with random content - it chooses SYNTAX ELEMENT VALUES at random (split flags, MPM flags and indices, chroma mode indices, coded block
flags, coefficient levels) and entropy-codes them; it never derives a prediction mode or reconstructs a sample.  What the values mean is decided by
the decoder: the reference's own decoder (oracle/_ref/vvdecapp_ref, built from /root/reference) decodes each stream here and its output MD5 becomes
the stream's `.yuv.md5` - the ground truth the drop-in decoder on the GPU back-end is then held to (tools/dropin_decode.py,
tests/test_dropin_library.py::test_decodes_conformance_bitstreams).

Syntax written (everything else is switched off in the parameter sets):
  * SPS / PPS (HLSyntaxReader::parseSPS / parsePPS), one IDR slice per picture with the picture header inside the slice header (parseSliceHeader /
    parsePictureHeader), Main 10, 4:2:0, 8 or 10 bit, CTU 32 / 64 / 128, single tree, quad-tree splits only;
  * per CTU (CABACReader::coding_tree_unit): split_cu_flag, intra_luma_mpm_flag / intra_luma_not_planar_flag / intra_luma_mpm_idx /
    intra_luma_mpm_remainder, intra_chroma_pred_mode, tu_cb_coded_flag / tu_cr_coded_flag / tu_y_coded_flag, residual_coding restricted to the first
    coefficient group with levels up to 3 (last_sig_coeff_{x,y}_prefix, sig_coeff_flag, abs_level_gtx_flag[0 / 1], par_level_flag, coeff_sign_flag),
    the implicit transform split of CUs larger than the maximum transform size; deblocking stays on.
  * CABAC: the arithmetic encoder of the standard (9.3.4) with the two-rate probability model of Contexts.h; initial values of the context sets are
    read from the reference's table (CommonLib/Contexts.cpp) when this tool runs.

That was the first version.  What has been added since is listed where it is used, in the comments of FIXTURES below: inter slices (skip / merge / AMVP, MMVD, affine, CIIP, GPM,
SbTMVP, AMVR, BCW, SMVD, SBT, weighted prediction), binary / ternary splits, dual trees, the intra tools with syntax of their own (MRL, ISP, MIP, CCLM - also in the chroma tree of
dual-tree pictures -, BDPCM), LFNST / MTS, transform skip and the regular residual beyond the first coefficient group, dependent quantisation, joint Cb-Cr, delta QPs and chroma QP
offsets, SAO / ALF / CC-ALF / LMCS / scaling-list APSs, slices / tiles / sub-pictures, virtual boundaries, LADF, reference picture resampling, wrap-around, 4:0:0, and - round 5 -
intra block copy, the one tool whose syntax values cannot be random (coding_unit_ibc).

  python tools/mini_vvenc.py --out tests/bitstreams [--seed N]     writes the fixture set (needs /root/reference for the context tables and
                                                                    oracle/_ref/vvdecapp_ref for the MD5s)
"""
import argparse
import hashlib
import os
import random
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("VVDEC_REFERENCE", "/root/reference")
APP_REF = os.path.join(ROOT, "oracle", "_ref", "vvdecapp_ref")


# ---------------------------------------------------------------------------------------------------------------------
# bits
# ---------------------------------------------------------------------------------------------------------------------
class Bits:
    def __init__(self):
        self.b = []

    def u(self, n, v):
        assert 0 <= v < (1 << n), (n, v)
        for i in range(n - 1, -1, -1):
            self.b.append((v >> i) & 1)

    def flag(self, v):
        self.b.append(1 if v else 0)

    def ue(self, v):
        assert v >= 0
        v += 1
        n = v.bit_length()
        self.u(n - 1, 0)
        self.u(n, v)

    def se(self, v):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def aligned(self):
        return len(self.b) % 8 == 0

    def align_zero(self):
        while not self.aligned():
            self.b.append(0)

    def trailing(self):              # rbsp_trailing_bits / byte_alignment(): a one, then zeros
        self.b.append(1)
        self.align_zero()

    def bytes(self):
        assert self.aligned()
        return bytes(int("".join(map(str, self.b[i:i + 8])), 2) for i in range(0, len(self.b), 8))


def nal(nal_type, payload, long_start=False, tid=0):
    """NAL unit header (7.3.1.2) + payload with emulation prevention, behind an Annex-B start code"""
    hdr = bytes([0, (nal_type << 3) | (tid + 1)])          # forbidden_zero_bit, nuh_reserved_zero_bit, nuh_layer_id = 0; type; temporal id + 1
    out = bytearray()
    zeros = 0
    for byte in hdr + payload:
        if zeros >= 2 and byte <= 3:
            out.append(3)
            zeros = 0
        out.append(byte)
        zeros = zeros + 1 if byte == 0 else 0
    return (b"\x00\x00\x00\x01" if long_start else b"\x00\x00\x01") + bytes(out)


NAL_IDR_N_LP, NAL_SPS, NAL_PPS = 8, 15, 16


# ---------------------------------------------------------------------------------------------------------------------
# CABAC
# ---------------------------------------------------------------------------------------------------------------------
CTX_CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mini_vvenc_contexts.json")


def load_context_tables():
    """{set name: [rows B, P, I, rate]} and the renormalisation table, from the reference's CommonLib/Contexts.cpp - or, where /root/reference does not
    exist (the GPU box), from tools/mini_vvenc_contexts.json: the same values (the CABAC initialisation tables of the standard), dumped by this function"""
    path = os.path.join(REF, "source", "Lib", "CommonLib", "Contexts.cpp")
    if not os.path.exists(path):
        import json
        c = json.load(open(CTX_CACHE))
        return c["tables"], c["renorm"]
    src = open(path).read()
    tables = {}
    for m in re.finditer(r"ContextSetCfg::(\w+)(\[\])?\s*=\s*(\{)?\s*((?:ContextSetCfg::addCtxSet\s*\(\{.*?\}\)\s*,?\s*)+)\}?;", src, re.S):
        name, sets = m.group(1), []
        for s in re.finditer(r"addCtxSet\s*\(\{(.*?)\}\)", m.group(4), re.S):
            rows = [[int(v) for v in re.findall(r"\d+", r.replace("CNU", "35").replace("DWS", "8"))] for r in re.findall(r"\{([^{}]*)\}", s.group(1))]      # (CNU 35, DWS 8: Contexts.cpp)
            sets.append(rows)
        tables[name] = sets if m.group(2) else sets[0]
    rn = re.search(r"m_RenormTable_32\s*\[\s*32\s*\]\s*=\s*\{(.*?)\}", src, re.S)
    renorm = [int(v) for v in re.findall(r"\d+", rn.group(1))]
    assert len(renorm) == 32 and "SplitFlag" in tables and len(tables["SplitFlag"][0]) == 9
    try:
        import json
        if not os.path.exists(CTX_CACHE) or json.load(open(CTX_CACHE)) != {"tables": tables, "renorm": renorm}:
            json.dump({"tables": tables, "renorm": renorm}, open(CTX_CACHE, "w"))
    except Exception:
        pass
    return tables, renorm


class Ctx:
    """BinProbModel (Contexts.h:71-160): two estimates of different adaptation rates"""
    MASK_0, MASK_1 = ((1 << 10) - 1) << 5, ((1 << 14) - 1) << 1

    def __init__(self, init_id, rate, qp):
        slope = (init_id >> 3) - 4
        offset = ((init_id & 7) * 18) + 1
        inistate = ((slope * (qp - 16)) >> 1) + offset
        p1 = min(127, max(1, inistate)) << 8
        self.s0, self.s1 = p1 & Ctx.MASK_0, p1 & Ctx.MASK_1
        r0 = 2 + ((rate >> 2) & 3)
        r1 = 3 + r0 + (rate & 3)
        self.r0, self.r1 = r0 + 5, r1 + 1
        self.d0 = [0xFFFF >> (16 - self.r0), 0xFFFF >> 1]
        self.d1 = [0xFFFF >> (16 - self.r1), 0xFFFF >> 1]

    def state(self):
        return (self.s0 + self.s1) >> 8

    def update(self, b):
        self.s0 += ((self.d0[b] - self.s0) >> self.r0) << 5
        self.s1 += ((self.d1[b] - self.s1) >> self.r1) << 1


class Cabac:
    """the arithmetic encoder (9.3.4.x as the usual encoder-side mirror: low / range / outstanding bytes)"""

    def __init__(self, tables, renorm, slice_type, qp):
        self.tables, self.renorm, self.st, self.qp = tables, renorm, slice_type, qp
        self.ctx = {}
        self.low, self.range, self.bits_left, self.buffered, self.num_buffered = 0, 510, 23, 0xFF, 0
        self.out = bytearray()
        self.nbins = 0

    def model(self, name, idx, sub=None):
        key = (name, sub, idx)
        if key not in self.ctx:
            t = self.tables[name] if sub is None else self.tables[name][sub]
            self.ctx[key] = Ctx(t[self.st][idx], t[3][idx], self.qp)
        return self.ctx[key]

    def _write_out(self):
        lead = self.low >> (24 - self.bits_left)
        self.bits_left += 8
        self.low &= 0xFFFFFFFF >> self.bits_left
        if lead == 0xFF:
            self.num_buffered += 1
        elif self.num_buffered > 0:
            carry = lead >> 8
            self.out.append((self.buffered + carry) & 0xFF)
            self.buffered = lead & 0xFF
            fill = (0xFF + carry) & 0xFF
            while self.num_buffered > 1:
                self.out.append(fill)
                self.num_buffered -= 1
        else:
            self.num_buffered = 1
            self.buffered = lead

    def _test(self):
        if self.bits_left < 12:
            self._write_out()

    def bin(self, b, name, idx, sub=None):
        m = self.model(name, idx, sub)
        self.nbins += 1
        q = m.state()
        mps = q >> 7
        if q & 0x80:
            q ^= 0xFF
        lps = (((q >> 2) * (self.range >> 5)) >> 1) + 4
        self.range -= lps
        if b != mps:
            n = self.renorm[lps >> 3]
            self.low = (self.low + self.range) << n
            self.range = lps << n
            self.bits_left -= n
        elif self.range < 256:
            self.low <<= 1
            self.range <<= 1
            self.bits_left -= 1
        m.update(b)
        self._test()

    def ep(self, b):
        self.low <<= 1
        if b:
            self.low += self.range
        self.bits_left -= 1
        self._test()

    def eps(self, v, n):
        for i in range(n - 1, -1, -1):
            self.ep((v >> i) & 1)

    def trm(self, b):
        self.range -= 2
        if b:
            self.low += self.range
            self.low <<= 7
            self.range = 2 << 7
            self.bits_left -= 7
        elif self.range >= 256:
            return
        else:
            self.low <<= 1
            self.range <<= 1
            self.bits_left -= 1
        self._test()

    def finish(self):
        """end_of_slice_one_bit was coded with trm(1): flush; returns the bits of the slice data (the caller appends rbsp_trailing_bits)"""
        if self.low >> (32 - self.bits_left):
            self.out.append((self.buffered + 1) & 0xFF)
            while self.num_buffered > 1:
                self.out.append(0)
                self.num_buffered -= 1
            self.low -= 1 << (32 - self.bits_left)
        else:
            if self.num_buffered > 0:
                self.out.append(self.buffered)
            while self.num_buffered > 1:
                self.out.append(0xFF)
                self.num_buffered -= 1
        bits = []
        for byte in self.out:
            bits += [(byte >> i) & 1 for i in range(7, -1, -1)]
        n = 24 - self.bits_left
        v = self.low >> 8
        bits += [(v >> i) & 1 for i in range(n - 1, -1, -1)]
        return bits


# ---------------------------------------------------------------------------------------------------------------------
# parameter sets and headers
# ---------------------------------------------------------------------------------------------------------------------
class Cfg:
    def __init__(self, width, height, log2_ctu=6, log2_min_qt=3, bit_depth=10, qp=30, max_tb64=True, p_split=0.6, p_cbf=0.5, p_cbf_chroma=0.3, deblock=True,
                 inter=False, tmvp=True, sbtmvp=False, bdof=True, dmvr=True, mmvd=False, affine=False, ciip=False, gpm=False, p_skip=0.3, p_intra=0.15, p_merge=0.5, max_mvd=24,
                 sao=False, lmcs=False, jccr=False, dep_quant=False, mtt_depth=0, p_mtt=0.5,
                 mrl=False, isp=False, mip=False, cclm=False, lfnst=False, mts=False, alf=False, ccalf=False, alf_aps=2, big_resi=False,
                 amvr=False, bcw=False, smvd=False, sbt=False, dqp=False, dual_tree=False, log2_min_qt_c=4, scaling=False, ts=False, bdpcm=False, ts_regular=False, part=None, lf_across=True, rpr=None, mono=False, subpic=None, chroma_qp=False, db_offsets=False, ladf=False, wrap=False, vb=False, wp=False, ibc=False, p_ibc=0.3):
        assert width % (1 << log2_ctu) == 0 and height % (1 << log2_ctu) == 0, "pictures of whole CTUs only (no implicit splits at the picture boundary)"
        self.__dict__.update(locals())
        self.log2_min_cb = 3                       # 8x8 luma / 4x4 chroma: no block below 4x4, no local dual tree
        self.log2_max_tb = 6 if (max_tb64 and log2_ctu > 5) else 5
        self.log2_max_btt = min(6, log2_ctu)           # largest block a binary / ternary split applies to (sps_log2_diff_max_bt / tt_min_qt)
        if mono:                                       # 4:0:0: nothing that only exists with chroma
            self.jccr = self.ccalf = self.cclm = self.chroma_qp = self.dual_tree = False
        if not inter:
            self.tmvp = self.sbtmvp = self.bdof = self.dmvr = self.mmvd = self.affine = self.ciip = self.gpm = False
            self.amvr = self.bcw = self.smvd = self.sbt = self.wrap = self.wp = False
        self.max_aff_merge = 5 if self.affine else (1 if (self.sbtmvp and self.tmvp) else 0)
        self.wrap_minus = 0 if width % 64 else 2       # wrap-around period: the picture's width, or two minimum coding blocks less
        self.partition = partition_of(self)
        # reference picture resampling: rpr = [(width, height, scaling window offsets or None), ...], one PPS each; a picture takes the sizes in turn every `rpr_every`
        # pictures (the first size is the sequence's maximum)
        self.sizes = [(width, height, None)] + list(rpr or [])
        for (ww, hh, _) in self.sizes:
            assert ww % (1 << log2_ctu) == 0 and hh % (1 << log2_ctu) == 0 and ww <= width and hh <= height


def partition_of(c):
    """part = None: one slice, one tile.  ("rows", [explicit slice heights in CTU rows]): one tile, rectangular slices that are bands of CTU rows (the rest of the tile in
    slices of the last explicit height, pps_exp_slice_height_in_ctus_minus1).  ("tiles", explicit column width, explicit row height): a grid of tiles (the rest of the
    picture in tiles of that size), one rectangular slice per tile.  -> dict(slices=[[(rx, ry) in coding order]], ctu=(slice, tile) per CTU) or None"""
    if not c.part:
        return None
    S = 1 << c.log2_ctu
    W, H = c.width // S, c.height // S
    slices, ctu = [], {}
    if c.part[0] == "rows":
        hs, rem = list(c.part[1]), H - sum(c.part[1])
        assert rem >= 0 and hs
        while rem >= hs[-1]:
            hs.append(hs[-1]); rem -= hs[-1]
        if rem > 0:
            hs.append(rem)
        y = 0
        for k, hh in enumerate(hs):
            cur = [(rx, ry) for ry in range(y, y + hh) for rx in range(W)]
            for xy in cur:
                ctu[xy] = (k, 0)
            slices.append(cur)
            y += hh
    else:
        cw, rh = c.part[1], c.part[2]
        cols, rows = [], []
        x = 0
        while x < W:
            cols.append((x, min(cw, W - x))); x += cw
        y = 0
        while y < H:
            rows.append((y, min(rh, H - y))); y += rh
        k = 0
        for (y0, hh) in rows:
            for (x0, ww) in cols:
                cur = [(rx, ry) for ry in range(y0, y0 + hh) for rx in range(x0, x0 + ww)]
                for xy in cur:
                    ctu[xy] = (k, k)
                slices.append(cur)
                k += 1
    return dict(slices=slices, ctu=ctu, W=W, H=H)


def write_sps(c):
    b = Bits()
    b.u(4, 0)                                        # sps_seq_parameter_set_id
    b.u(4, 0)                                        # sps_video_parameter_set_id
    b.u(3, 0)                                        # sps_max_sublayers_minus1
    b.u(2, 0 if c.mono else 1)                       # sps_chroma_format_idc: 4:0:0 / 4:2:0
    b.u(2, c.log2_ctu - 5)                           # sps_log2_ctu_size_minus5
    b.flag(1)                                        # sps_ptl_dpb_hrd_params_present_flag
    # profile_tier_level( 1, 0 )
    b.u(7, 1)                                        # general_profile_idc: Main 10
    b.flag(0)                                        # general_tier_flag
    b.u(8, 16 * 5 + 3 * 1)                           # general_level_idc: 5.1
    b.flag(1)                                        # ptl_frame_only_constraint_flag
    b.flag(0)                                        # ptl_multilayer_enabled_flag
    b.flag(0)                                        # gci_present_flag
    b.align_zero()                                   # gci_alignment_zero_bit
    b.align_zero()                                   # ptl_reserved_zero_bit
    b.u(8, 0)                                        # ptl_num_sub_profiles
    b.flag(0)                                        # sps_gdr_enabled_flag
    b.flag(len(c.sizes) > 1)                         # sps_ref_pic_resampling_enabled_flag
    if len(c.sizes) > 1:
        b.flag(1)                                    # sps_res_change_in_clvs_allowed_flag
    b.ue(c.width)                                    # sps_pic_width_max_in_luma_samples
    b.ue(c.height)
    b.flag(0)                                        # sps_conformance_window_flag
    b.flag(bool(c.subpic))                           # sps_subpic_info_present_flag
    if c.subpic:
        # one sub-picture per slice of the partition (a band of CTU rows or a tile); c.subpic = per sub-picture (treated as picture, loop filter across)
        P, S = c.partition, 1 << c.log2_ctu
        n = len(P["slices"])
        assert n > 1 and len(c.subpic) == n
        b.ue(n - 1)                                  # sps_num_subpics_minus1
        b.flag(0)                                    # sps_independent_subpics_flag
        b.flag(0)                                    # sps_subpic_same_size_flag
        bw, bh = max(0, (P["W"] - 1).bit_length()), max(0, (P["H"] - 1).bit_length())
        for i, ctus in enumerate(P["slices"]):
            x0, y0 = min(x for x, _ in ctus), min(y for _, y in ctus)
            x1, y1 = max(x for x, _ in ctus), max(y for _, y in ctus)
            if i > 0 and c.width > S:
                b.u(bw, x0)                          # sps_subpic_ctu_top_left_x
            if i > 0 and c.height > S:
                b.u(bh, y0)                          # sps_subpic_ctu_top_left_y
            if i < n - 1 and c.width > S:
                b.u(bw, x1 - x0)                     # sps_subpic_width_minus1
            if i < n - 1 and c.height > S:
                b.u(bh, y1 - y0)                     # sps_subpic_height_minus1
            b.flag(c.subpic[i][0])                   # sps_subpic_treated_as_pic_flag
            b.flag(c.subpic[i][1])                   # sps_loop_filter_across_subpic_enabled_flag
        b.ue(max(1, (n - 1).bit_length()) - 1)       # sps_subpic_id_len_minus1
        b.flag(0)                                    # sps_subpic_id_mapping_explicitly_signalled_flag
    b.ue(c.bit_depth - 8)                            # sps_bitdepth_minus8
    b.flag(0)                                        # sps_entropy_coding_sync_enabled_flag
    b.flag(0)                                        # sps_entry_point_offsets_present_flag
    b.u(4, 4)                                        # sps_log2_max_pic_order_cnt_lsb_minus4
    b.flag(0)                                        # sps_poc_msb_cycle_flag
    b.u(2, 0)                                        # sps_num_extra_ph_bytes
    b.u(2, 0)                                        # sps_num_extra_sh_bytes
    b.ue(5 if c.inter else 1)                        # dpb_max_dec_pic_buffering_minus1[0]
    b.ue(3 if c.inter else 0)                        # dpb_max_num_reorder_pics[0]
    b.ue(0)                                          # dpb_max_latency_increase_plus1[0]
    b.ue(c.log2_min_cb - 2)                          # sps_log2_min_luma_coding_block_size_minus2
    b.flag(0)                                        # sps_partition_constraints_override_enabled_flag
    b.ue(c.log2_min_qt - c.log2_min_cb)              # sps_log2_diff_min_qt_min_cb_intra_slice_luma
    b.ue(c.mtt_depth)                                # sps_max_mtt_hierarchy_depth_intra_slice_luma
    if c.mtt_depth:
        b.ue(c.log2_max_btt - c.log2_min_qt)         # sps_log2_diff_max_bt_min_qt_intra_slice_luma
        b.ue(c.log2_max_btt - c.log2_min_qt)         # sps_log2_diff_max_tt_min_qt_intra_slice_luma
    if not c.mono:
        b.flag(c.dual_tree)                          # sps_qtbtt_dual_tree_intra_flag
    if c.dual_tree:
        b.ue(c.log2_min_qt_c - c.log2_min_cb)        # sps_log2_diff_min_qt_min_cb_intra_slice_chroma
        b.ue(c.mtt_depth)                            # sps_max_mtt_hierarchy_depth_intra_slice_chroma
        if c.mtt_depth:
            b.ue(c.log2_max_btt - c.log2_min_qt_c)   # sps_log2_diff_max_bt_min_qt_intra_slice_chroma
            b.ue(c.log2_max_btt - c.log2_min_qt_c)   # sps_log2_diff_max_tt_min_qt_intra_slice_chroma
    b.ue(c.log2_min_qt - c.log2_min_cb)              # sps_log2_diff_min_qt_min_cb_inter_slice
    b.ue(c.mtt_depth)                                # sps_max_mtt_hierarchy_depth_inter_slice
    if c.mtt_depth:
        b.ue(c.log2_max_btt - c.log2_min_qt)         # sps_log2_diff_max_bt_min_qt_inter_slice
        b.ue(c.log2_max_btt - c.log2_min_qt)         # sps_log2_diff_max_tt_min_qt_inter_slice
    if c.log2_ctu > 5:
        b.flag(c.log2_max_tb == 6)                   # sps_max_luma_transform_size_64_flag
    b.flag(c.ts)                                     # sps_transform_skip_enabled_flag
    if c.ts:
        b.ue(3)                                      # sps_log2_transform_skip_max_size_minus2: 32
        b.flag(c.bdpcm)                              # sps_bdpcm_enabled_flag
    b.flag(c.mts)                                    # sps_mts_enabled_flag
    if c.mts:
        b.flag(1)                                    # sps_explicit_mts_intra_enabled_flag
        b.flag(1)                                    # sps_explicit_mts_inter_enabled_flag
    b.flag(c.lfnst)                                  # sps_lfnst_enabled_flag
    if not c.mono:
        b.flag(c.jccr)                               # sps_joint_cbcr_enabled_flag
        b.flag(1)                                    # sps_same_qp_table_for_chroma_flag
        b.se(0)                                      # sps_qp_table_start_minus26[0]
        b.ue(0)                                      # sps_num_points_in_qp_table_minus1[0]
        b.ue(0)                                      # sps_delta_qp_in_val_minus1[0][0]
        b.ue(1)                                      # sps_delta_qp_diff_val[0][0]: the identity table
    b.flag(c.sao)                                    # sps_sao_enabled_flag
    b.flag(c.alf)                                    # sps_alf_enabled_flag
    if c.alf and not c.mono:
        b.flag(c.ccalf)                              # sps_ccalf_enabled_flag
    b.flag(c.lmcs)                                   # sps_lmcs_enable_flag
    b.flag(c.wp)                                     # sps_weighted_pred_flag
    b.flag(c.wp)                                     # sps_weighted_bipred_flag
    b.flag(0)                                        # sps_long_term_ref_pics_flag
    b.flag(0)                                        # sps_idr_rpl_present_flag
    b.flag(1)                                        # sps_rpl1_same_as_rpl0_flag
    b.ue(0)                                          # sps_num_ref_pic_lists[0]
    b.flag(c.wrap)                                   # sps_ref_wraparound_enabled_flag
    b.flag(c.tmvp)                                   # sps_temporal_mvp_enabled_flag
    if c.tmvp:
        b.flag(c.sbtmvp)                             # sps_sbtmvp_enabled_flag
    b.flag(c.amvr)                                   # sps_amvr_enabled_flag
    b.flag(c.bdof)                                   # sps_bdof_enabled_flag
    if c.bdof:
        b.flag(0)                                    # sps_bdof_control_present_in_ph_flag
    b.flag(c.smvd)                                   # sps_smvd_enabled_flag
    b.flag(c.dmvr)                                   # sps_dmvr_enabled_flag
    if c.dmvr:
        b.flag(0)                                    # sps_dmvr_control_present_in_ph_flag
    b.flag(c.mmvd)                                   # sps_mmvd_enabled_flag
    if c.mmvd:
        b.flag(0)                                    # sps_mmvd_fullpel_only_flag
    b.ue(0)                                          # sps_six_minus_max_num_merge_cand: MaxNumMergeCand = 6
    b.flag(c.sbt)                                    # sps_sbt_enabled_flag
    b.flag(c.affine)                                 # sps_affine_enabled_flag
    if c.affine:
        b.ue(0)                                      # sps_five_minus_max_num_subblock_merge_cand
        b.flag(1)                                    # sps_6param_affine_enabled_flag
        if c.amvr:
            b.flag(1)                                # sps_affine_amvr_enabled_flag
        b.flag(1)                                    # sps_affine_prof_enabled_flag
        b.flag(0)                                    # sps_prof_control_present_in_ph_flag
    b.flag(c.bcw)                                    # sps_bcw_enabled_flag
    b.flag(c.ciip)                                   # sps_ciip_enabled_flag
    b.flag(c.gpm)                                    # sps_gpm_enabled_flag (MaxNumMergeCand = 6)
    if c.gpm:
        b.ue(0)                                      # sps_max_num_merge_cand_minus_max_num_gpm_cand
    b.ue(0)                                          # sps_log2_parallel_merge_level_minus2
    b.flag(c.isp)                                    # sps_isp_enabled_flag
    b.flag(c.mrl)                                    # sps_mrl_enabled_flag
    b.flag(c.mip)                                    # sps_mip_enabled_flag
    if not c.mono:
        b.flag(c.cclm)                               # sps_cclm_enabled_flag
        b.flag(0)                                    # sps_chroma_horizontal_collocated_flag
        b.flag(0)                                    # sps_chroma_vertical_collocated_flag
    b.flag(0)                                        # sps_palette_enabled_flag
    if c.ts:
        b.ue(2 if c.bit_depth > 8 else 0)            # sps_internal_bit_depth_minus_input_bit_depth (the QP floor of transform-skip blocks)
    b.flag(1 if c.ibc else 0)                        # sps_ibc_enabled_flag
    if c.ibc:
        b.ue(5)                                      # sps_six_minus_max_num_ibc_merge_cand: one candidate - no merge index, no predictor flag (the writer models candidate 0)
    b.flag(c.ladf)                                   # sps_ladf_enabled_flag
    if c.ladf:
        b.u(2, 2)                                    # sps_num_ladf_intervals_minus2: four intervals
        b.se(2)                                      # sps_ladf_lowest_interval_qp_offset
        for off, thr in ((-3, 150), (1, 250), (4, 300)):
            b.se(off)                                # sps_ladf_qp_offset[i]
            b.ue((thr >> (10 - c.bit_depth)) - 1)    # sps_ladf_delta_threshold_minus1[i]
    b.flag(c.scaling)                                # sps_explicit_scaling_list_enabled_flag
    if c.scaling and c.lfnst:
        b.flag(rseed_bit(c))                         # sps_scaling_matrix_for_lfnst_disabled_flag
    b.flag(c.dep_quant)                              # sps_dep_quant_enabled_flag
    b.flag(0)                                        # sps_sign_data_hiding_enabled_flag
    b.flag(c.vb)                                     # sps_virtual_boundaries_enabled_flag
    if c.vb:
        b.flag(1)                                    # sps_virtual_boundaries_present_flag
        S = 1 << c.log2_ctu
        xs = [p for p in (S + 24, 2 * S + S // 2) if p < c.width][:2] if c.width > S + 24 else []
        ys = [p for p in (S // 2 + 8,) if p < c.height]
        b.ue(len(xs))                                # sps_num_ver_virtual_boundaries
        for p in xs:
            b.ue(p // 8 - 1)                         # sps_virtual_boundary_pos_x_minus1
        b.ue(len(ys))                                # sps_num_hor_virtual_boundaries
        for p in ys:
            b.ue(p // 8 - 1)                         # sps_virtual_boundary_pos_y_minus1
    b.flag(0)                                        # sps_timing_hrd_params_present_flag
    b.flag(0)                                        # sps_field_seq_flag
    b.flag(0)                                        # sps_vui_parameters_present_flag
    b.flag(0)                                        # sps_extension_present_flag
    b.trailing()
    return b.bytes()


def write_pps(c, pps_id=0, win=None):
    b = Bits()
    b.u(6, pps_id)                                   # pps_pic_parameter_set_id
    b.u(4, 0)                                        # pps_seq_parameter_set_id
    b.flag(0)                                        # pps_mixed_nalu_types_in_pic_flag
    b.ue(c.width)
    b.ue(c.height)
    b.flag(0)                                        # pps_conformance_window_flag
    b.flag(win is not None)                          # pps_scaling_window_explicit_signalling_flag
    if win is not None:
        for v in win:
            b.se(v)                                  # pps_scaling_win_left / right / top / bottom_offset (chroma samples)
    b.flag(0)                                        # pps_output_flag_present_flag
    P = c.partition
    b.flag(0 if P else 1)                            # pps_no_pic_partition_flag
    b.flag(0)                                        # pps_subpic_id_mapping_present_flag
    if P:
        n = len(P["slices"])
        b.u(2, c.log2_ctu - 5)                       # pps_log2_ctu_size_minus5
        b.ue(0)                                      # pps_num_exp_tile_columns_minus1
        b.ue(0)                                      # pps_num_exp_tile_rows_minus1
        if c.part[0] == "rows":
            b.ue(P["W"] - 1)                         # pps_tile_column_width_minus1[0]: one tile
            b.ue(P["H"] - 1)                         # pps_tile_row_height_minus1[0]
            b.flag(bool(c.subpic))                   # pps_single_slice_per_subpic_flag  (one tile: rectangular slices)
            if not c.subpic:
                b.ue(n - 1)                          # pps_num_slices_in_pic_minus1
                if n - 1 > 1:
                    b.flag(0)                        # pps_tile_idx_delta_present_flag
                b.ue(len(c.part[1]))                 # pps_num_exp_slices_in_tile[0]  (slice 0: neither a width nor a height in tiles is read in a 1 x 1 grid)
                for hh in c.part[1]:
                    b.ue(hh - 1)                     # pps_exp_slice_height_in_ctus_minus1
        else:
            ncol, nrow = -(-P["W"] // c.part[1]), -(-P["H"] // c.part[2])
            b.ue(c.part[1] - 1)                      # pps_tile_column_width_minus1[0] (the other columns: the same width, the last one what is left)
            b.ue(c.part[2] - 1)                      # pps_tile_row_height_minus1[0]
            b.flag(c.lf_across)                      # pps_loop_filter_across_tiles_enabled_flag
            b.flag(1)                                # pps_rect_slice_flag
            b.flag(bool(c.subpic))                   # pps_single_slice_per_subpic_flag
            if not c.subpic:
                b.ue(n - 1)                          # pps_num_slices_in_pic_minus1
                if n - 1 > 1:
                    b.flag(0)                        # pps_tile_idx_delta_present_flag
            heights = [min(c.part[2], P["H"] - r * c.part[2]) for r in range(nrow)]
            for i in range(0 if c.subpic else n - 1):      # (the last slice takes what is left)
                tx, ty = i % ncol, i // ncol
                if tx != ncol - 1:
                    b.ue(0)                          # pps_slice_width_in_tiles_minus1
                if ty != nrow - 1 and tx == 0:
                    b.ue(0)                          # pps_slice_height_in_tiles_minus1
                if heights[ty] > 1:
                    b.ue(0)                          # pps_num_exp_slices_in_tile: the tile is one slice
        b.flag(c.lf_across)                          # pps_loop_filter_across_slices_enabled_flag
    b.flag(0)                                        # pps_cabac_init_present_flag
    b.ue(0)                                          # pps_num_ref_idx_default_active_minus1[0]
    b.ue(0)                                          # pps_num_ref_idx_default_active_minus1[1]
    b.flag(0)                                        # pps_rpl1_idx_present_flag
    b.flag(c.wp)                                     # pps_weighted_pred_flag
    b.flag(c.wp)                                     # pps_weighted_bipred_flag
    b.flag(c.wrap)                                   # pps_ref_wraparound_enabled_flag
    if c.wrap:
        b.ue(c.wrap_minus)                           # pps_pic_width_minus_wraparound_offset (in minimum coding blocks)
    b.se(0)                                          # pps_init_qp_minus26
    b.flag(c.dqp)                                    # pps_cu_qp_delta_enabled_flag
    b.flag(c.chroma_qp)                              # pps_chroma_tool_offsets_present_flag
    if c.chroma_qp:
        b.se(3)                                      # pps_cb_qp_offset
        b.se(-4)                                     # pps_cr_qp_offset
        b.flag(1)                                    # pps_joint_cbcr_qp_offset_present_flag
        b.se(2)                                      # pps_joint_cbcr_qp_offset_value
        b.flag(1)                                    # pps_slice_chroma_qp_offsets_present_flag
        b.flag(0)                                    # pps_cu_chroma_qp_offset_list_enabled_flag
    b.flag(1)                                        # pps_deblocking_filter_control_present_flag
    b.flag(0)                                        # pps_deblocking_filter_override_enabled_flag
    b.flag(0 if c.deblock else 1)                    # pps_deblocking_filter_disabled_flag
    if c.deblock:
        b.se(3 if c.db_offsets else 0)               # pps_luma_beta_offset_div2
        b.se(-2 if c.db_offsets else 0)              # pps_luma_tc_offset_div2
        if c.chroma_qp:
            b.se(-3 if c.db_offsets else 0)          # pps_cb_beta_offset_div2
            b.se(4 if c.db_offsets else 0)           # pps_cb_tc_offset_div2
            b.se(5 if c.db_offsets else 0)           # pps_cr_beta_offset_div2
            b.se(-1 if c.db_offsets else 0)          # pps_cr_tc_offset_div2
    if P:
        b.flag(0)                                    # pps_rpl_info_in_ph_flag: reference picture lists, SAO, ALF and the QP stay in the slice headers
        b.flag(0)                                    # pps_sao_info_in_ph_flag
        b.flag(0)                                    # pps_alf_info_in_ph_flag
        b.flag(0)                                    # pps_qp_delta_info_in_ph_flag
    b.flag(0)                                        # pps_picture_header_extension_present_flag
    b.flag(0)                                        # pps_slice_header_extension_present_flag
    b.flag(0)                                        # pps_extension_flag
    b.trailing()
    return b.bytes()


NAL_PREFIX_APS = 17
NAL_PH = 19
NAL_SUFFIX_SEI = 24


def write_lmcs_aps(c, rng, aps_id):
    """adaptation_parameter_set_rbsp with lmcs_data (parseAPS / parseLmcsAps): bins 1..14, code words around the default one, their sum below the range"""
    b = Bits()
    b.u(3, 1)                                        # aps_params_type: LMCS_APS
    b.u(5, aps_id)                                   # aps_adaptation_parameter_set_id
    b.flag(not c.mono)                               # aps_chroma_present_flag
    b.ue(1)                                          # lmcs_min_bin_idx
    b.ue(1)                                          # lmcs_delta_max_bin_idx: LmcsMaxBinIdx = 14
    org = (1 << c.bit_depth) // 16
    prec = max(1, (org // 4).bit_length())
    b.ue(prec - 1)                                   # lmcs_delta_cw_prec_minus1
    for i in range(1, 15):
        d = rng.randrange(-(org // 4) + 1, org // 4)
        b.u(prec, abs(d))                            # lmcs_delta_abs_cw[i]
        if d:
            b.flag(d < 0)                            # lmcs_delta_sign_cw_flag[i]
    crs = rng.randrange(-3, 4)
    if not c.mono:
        b.u(3, abs(crs))                             # lmcs_delta_abs_crs
        if crs:
            b.flag(crs < 0)                          # lmcs_delta_sign_crs_flag
    b.flag(0)                                        # aps_extension_flag
    b.trailing()
    return b.bytes()


def rseed_bit(c):
    return (c.width // 8 + c.qp) & 1                 # (a configuration-dependent constant: both values of the flag appear among the fixtures)


def diag_scan(n):
    """positions (x, y) of an n x n block in the diagonal scan (ScanGenerator, Rom.cpp:130-172)"""
    out = []
    for d in range(2 * n - 1):
        for y in range(min(d, n - 1), -1, -1):
            if d - y < n:
                out.append((d - y, y))
    return out


def write_scaling_aps(c, rng, aps_id):
    """adaptation_parameter_set_rbsp with scaling_list_data (parseScalingList / decodeScalingList): the 28 matrices - flat, copied or predicted from an earlier one of their
    size, or coded afresh -, DC entries from 16x16 on, no coefficients for the zeroed-out quadrant of the 64x64 matrices"""
    b = Bits()
    b.u(3, 2)                                        # aps_params_type: SCALING_APS
    b.u(5, aps_id)                                   # aps_adaptation_parameter_set_id
    b.flag(not c.mono)                               # aps_chroma_present_flag
    rec, dc = {}, {}
    for sid in range(28):
        if c.mono and not (sid % 3 == 2 or sid == 27):                         # (without chroma only the luma matrices: ScalingList::isLumaScalingList)
            n0 = 2 if sid < 2 else (4 if sid < 8 else 8)
            rec[sid], dc[sid] = [16] * (n0 * n0), 16
            continue
        n = 2 if sid < 2 else (4 if sid < 8 else 8)
        first = sid in (0, 2, 8)
        max_delta = sid if sid < 2 else (sid - 2 if sid < 8 else sid - 8)
        mode = rng.choice(["copy", "pred", "new", "new"])
        delta = 0
        if mode != "new" and not first:
            delta = rng.randrange(0, max_delta + 1)
            if sid > 25 and mode == "pred":
                delta = 0                            # (the uncoded quadrant takes prediction + last sum: stays positive with a flat prediction)
            if c.mono:
                delta = 0                            # (4:0:0: the matrices in between are not sent; this decoder predicts from whatever they hold)
        b.flag(mode == "copy")                       # scaling_list_copy_mode_flag
        if mode != "copy":
            b.flag(mode == "pred")                   # scaling_list_pred_mode_flag
        if mode != "new" and not first:
            b.ue(delta)                              # scaling_list_pred_id_delta
        if mode == "new":
            pred, dc_pred = [8] * (n * n), 8
        elif delta == 0:
            pred, dc_pred = [16] * (n * n), 16
        else:
            ref = sid - delta
            pred, dc_pred = list(rec[ref]), (dc[ref] if ref > 13 else rec[ref][0])
        if mode == "copy":
            rec[sid] = pred
            if sid >= 14:
                dc[sid] = dc_pred
            continue
        nxt = 0
        def wrap(v):
            return ((v + 128) & 255) - 128
        if sid > 13:
            t = rng.randrange(4, 60)
            d = wrap(t - dc_pred)
            b.se(d)                                  # scaling_list_dc_coef
            nxt += d
            dc[sid] = (dc_pred + d) & 255
        cur = list(pred)
        scan8, scann = diag_scan(8), diag_scan(n)
        for i in range(n * n):
            x, y = scan8[i][0], scan8[i][1]          # (the zero-out test looks at the 8x8 scan whatever the size: only 8x8 matrices reach id 26)
            px, py = scann[i]
            pos = py * n + px
            if not (sid > 25 and x >= 4 and y >= 4):
                t = max(1, min(255, 16 + int(rng.gauss(0, 1) * 10) + 2 * (px + py)))
                d = wrap(t - pred[pos] - nxt)
                b.se(d)                              # scaling_list_delta_coef
                nxt += d
            cur[pos] = (pred[pos] + nxt) & 255
            assert cur[pos] > 0
        rec[sid] = cur
    b.flag(0)                                        # aps_extension_flag
    b.trailing()
    return b.bytes()


def write_alf_aps(c, rng, aps_id):
    """adaptation_parameter_set_rbsp with alf_data (parseAlfAps / alfFilterCoeffs): luma filters with a class map, chroma alternatives, CC-ALF filters, clipping indices.
    -> (bytes, number of chroma alternatives, CC-ALF filter counts)"""
    b = Bits()
    b.u(3, 0)                                        # aps_params_type: ALF_APS
    b.u(5, aps_id)                                   # aps_adaptation_parameter_set_id
    b.flag(not c.mono)                               # aps_chroma_present_flag
    b.flag(1)                                        # alf_luma_filter_signal_flag
    if not c.mono:
        b.flag(1)                                    # alf_chroma_filter_signal_flag
        b.flag(c.ccalf)                              # alf_cc_cb_filter_signal_flag
        b.flag(c.ccalf)                              # alf_cc_cr_filter_signal_flag

    def coeffs(n, m):
        for _ in range(n):
            v = rng.choice([0, 0, 1, 1, 2, 3, 5, 8, 13, m])
            b.ue(v)                                  # alf_luma_coeff_abs / alf_chroma_coeff_abs
            if v:
                b.flag(rng.randrange(0, 2))          # ..._sign
    clip = rng.random() < 0.7
    b.flag(clip)                                     # alf_luma_clip_flag
    nf = rng.choice([1, 2, 5, 25])
    b.ue(nf - 1)                                     # alf_luma_num_filters_signalled_minus1
    if nf > 1:
        length = (nf - 1).bit_length()
        for _ in range(25):
            b.u(length, rng.randrange(0, nf))        # alf_luma_coeff_delta_idx
    coeffs(nf * 12, 20)
    if clip:
        for _ in range(nf * 12):
            b.u(2, rng.randrange(0, 4))              # alf_luma_clip_idx
    cclip = rng.random() < 0.7
    nalt = rng.randrange(1, 5)
    if not c.mono:
        b.flag(cclip)                                # alf_chroma_clip_flag
        b.ue(nalt - 1)                               # alf_chroma_num_alt_filters_minus1
    for _ in range(0 if c.mono else nalt):
        coeffs(6, 20)
        if cclip:
            for _ in range(6):
                b.u(2, rng.randrange(0, 4))          # alf_chroma_clip_idx
    ncc = [0, 0]
    if c.ccalf:
        for k in range(2):
            ncc[k] = rng.randrange(1, 5)
            b.ue(ncc[k] - 1)                         # alf_cc_cb / cr_filters_signalled_minus1
            for _ in range(ncc[k] * 7):
                v = rng.choice([0, 0, 1, 2, 3, 4])
                b.u(3, v)                            # alf_cc_cb / cr_mapped_coeff_abs
                if v:
                    b.flag(rng.randrange(0, 2))      # alf_cc_cb / cr_coeff_sign
    b.flag(0)                                        # aps_extension_flag
    b.trailing()
    return b.bytes(), nalt, ncc


def write_rpl(b, cur_poc, ref_pocs, wp=False):
    """ref_pic_list_struct (parseRefPicList): short-term entries only, deltas relative to the previous entry"""
    b.ue(len(ref_pocs))                              # num_ref_entries
    prev = 0
    for i, r in enumerate(ref_pocs):
        d = (cur_poc - r) - prev
        if wp and i > 0:
            b.ue(abs(d))                             # abs_delta_poc_st (with weighted prediction an entry may repeat the picture before it)
            if d:
                b.flag(d > 0)
        else:
            assert d != 0
            b.ue(abs(d) - 1)                         # abs_delta_poc_st (+ 1)
            b.flag(d > 0)                            # strp_entry_sign_flag: 1 = a picture that precedes the current one in output order
        prev = cur_poc - r


def write_slice_header(c, b, pic, sl=None):
    """pic: dict(poc, type 'I' / 'P' / 'B', idr, l0, l1 (POCs)).  sl: None = the picture's only slice, the picture header inside its header; "ph" = the picture header
    alone (a PH NAL unit in front of the slices of a picture that has several); dict(idx, n, type, qp, ...) = one of several slices"""
    idr, st = pic["idr"], pic["type"]
    multi = sl is not None
    if not multi:
        b.flag(1)                                    # sh_picture_header_in_slice_header_flag
    if not multi or sl == "ph":
        write_picture_header(c, b, pic)
        if sl == "ph":
            b.trailing()
            return
    else:
        b.flag(0)                                    # sh_picture_header_in_slice_header_flag
        if c.subpic:
            b.u(max(1, (sl["n"] - 1).bit_length()), sl["idx"])      # sh_subpic_id (the sub-picture's only slice: no address)
        elif sl["n"] > 1:
            b.u((sl["n"] - 1).bit_length(), sl["idx"])      # sh_slice_address: index of the rectangular slice
        st = sl["type"]
    write_slice_header_rest(c, b, pic, st, sl if multi else None)


def write_picture_header(c, b, pic):
    idr, st = pic["idr"], pic["type"]
    b.flag(1 if idr else 0)                          # ph_gdr_or_irap_pic_flag
    b.flag(0)                                        # ph_non_ref_pic_flag
    if idr:
        b.flag(0)                                    # ph_gdr_pic_flag
    inter_allowed = not idr
    b.flag(inter_allowed)                            # ph_inter_slice_allowed_flag
    if inter_allowed:
        b.flag(1)                                    # ph_intra_slice_allowed_flag
    b.ue(pic.get("pps", 0))                          # ph_pic_parameter_set_id
    b.u(8, pic["poc"] & 255)                         # ph_pic_order_cnt_lsb
    if c.lmcs:
        b.flag(1)                                    # ph_lmcs_enabled_flag
        b.u(2, 0)                                    # ph_lmcs_aps_id
        if not c.mono:
            b.flag(pic.get("cscale", 1))             # ph_chroma_residual_scale_flag
    if c.scaling:
        b.flag(pic.get("scaling", 1))                # ph_explicit_scaling_list_enabled_flag
        if pic.get("scaling", 1):
            b.u(3, pic.get("scaling_aps", 0))        # ph_scaling_list_aps_id
    # (no ALF, scaling lists, virtual boundaries, output flag, RPL in the PH, partition overrides, chroma QP offset lists)
    if c.dqp:
        b.ue(0)                                      # ph_cu_qp_delta_subdiv_intra_slice: one quantisation group per CTU
    if inter_allowed:
        if c.dqp:
            b.ue(0)                                  # ph_cu_qp_delta_subdiv_inter_slice
        if c.tmvp:
            b.flag(pic.get("tmvp", 1))               # ph_temporal_mvp_enabled_flag (off when no reference picture has this picture's size: the collocated picture may not be scaled)
        b.flag(0)                                    # ph_mvd_l1_zero_flag
    if c.jccr:
        b.flag(pic.get("jccr_sign", 0))              # ph_joint_cbcr_sign_flag
    # (no QP delta, SAO, deblocking info in the PH)


def write_slice_header_rest(c, b, pic, st, sl):
    idr = pic["idr"]
    inter_allowed = not idr
    # slice header proper: one slice per picture
    if inter_allowed:
        b.ue({"B": 0, "P": 1, "I": 2}[st])           # sh_slice_type
    if idr:
        b.flag(0)                                    # sh_no_output_of_prior_pics_flag
    if c.alf:
        a = sl["alf"] if sl else pic["alf"]
        b.flag(a["on"])                              # sh_alf_enabled_flag
        if a["on"]:
            b.u(3, len(a["luma_aps"]))               # sh_num_alf_aps_ids_luma
            for i in a["luma_aps"]:
                b.u(3, i)                            # sh_alf_aps_id_luma
            if not c.mono:
                b.flag(a["cb"])                      # sh_alf_cb_enabled_flag
                b.flag(a["cr"])                      # sh_alf_cr_enabled_flag
            if (a["cb"] or a["cr"]) and not c.mono:
                b.u(3, a["chroma_aps"])              # sh_alf_aps_id_chroma
            if c.ccalf:
                b.flag(a["cc_cb"] is not None)       # sh_alf_cc_cb_enabled_flag
                if a["cc_cb"] is not None:
                    b.u(3, a["cc_cb"])               # sh_alf_cc_cb_aps_id
                b.flag(a["cc_cr"] is not None)       # sh_alf_cc_cr_enabled_flag
                if a["cc_cr"] is not None:
                    b.u(3, a["cc_cr"])
    if sl:                                           # (with the picture header in a NAL unit of its own the slices say whether they use the picture's LMCS / scaling lists)
        if c.lmcs:
            b.flag(sl["lmcs"])                       # sh_lmcs_used_flag
        if c.scaling and pic.get("scaling", 1):
            b.flag(sl["scaling"])                    # sh_explicit_scaling_list_used_flag
    if not idr:
        write_rpl(b, pic["poc"], pic["l0"], c.wp)    # ref_pic_lists(): both lists, whatever the slice type
        write_rpl(b, pic["poc"], pic["l1"], c.wp)
        n0, n1 = len(pic["l0"]), len(pic["l1"])
        if (st != "I" and n0 > 1) or (st == "B" and n1 > 1):
            b.flag(1)                                # sh_num_ref_idx_active_override_flag
            if n0 > 1:
                b.ue(n0 - 1)                         # sh_num_ref_idx_active_minus1[0]
            if st == "B" and n1 > 1:
                b.ue(n1 - 1)
        if st != "I" and c.tmvp and pic.get("tmvp", 1):
            col_l0 = 1
            if st == "B":
                col_l0 = pic.get("col_l0", 1)
                b.flag(col_l0)                       # sh_collocated_from_l0_flag
            if (col_l0 and n0 > 1) or (not col_l0 and n1 > 1):
                b.ue(pic.get("col_idx", 0) if (st == "B" or pic.get("col_l0", 1)) else 0)      # sh_collocated_ref_idx
        if st != "I" and c.wp:
            # pred_weight_table() (parsePredWeightTable): a denominator, flags per entry, weights and offsets
            wt = pic["wp"]
            b.ue(wt["denom"])                        # luma_log2_weight_denom
            b.se(wt["dchroma"])                      # delta_chroma_log2_weight_denom
            for lst, n in ((0, n0), (1, n1 if st == "B" else 0)):
                ent = wt["l%d" % lst][:n]
                for e in ent:
                    b.flag(e["luma"] is not None)    # luma_weight_lX_flag
                for e in ent:
                    b.flag(e["chroma"] is not None)  # chroma_weight_lX_flag
                for e in ent:
                    if e["luma"] is not None:
                        b.se(e["luma"][0])           # delta_luma_weight_lX
                        b.se(e["luma"][1])           # luma_offset_lX
                    if e["chroma"] is not None:
                        for j in range(2):
                            b.se(e["chroma"][j][0])  # delta_chroma_weight_lX
                            b.se(e["chroma"][j][1])  # delta_chroma_offset_lX
    b.se((sl["qp"] if sl else c.qp) - 26)            # sh_qp_delta
    if c.chroma_qp:
        b.se(pic.get("cb_off", -2))                  # sh_cb_qp_offset
        b.se(pic.get("cr_off", 3))                   # sh_cr_qp_offset
        if c.jccr:
            b.se(-1)                                 # sh_joint_cbcr_qp_offset
    if c.sao:
        b.flag(sl["sao"][0] if sl else 1)            # sh_sao_luma_used_flag
        if not c.mono:
            b.flag(sl["sao"][1] if sl else 1)        # sh_sao_chroma_used_flag
    dq = c.dep_quant and (sl["dq"] if sl else True)
    if c.dep_quant:
        b.flag(dq)                                   # sh_dep_quant_used_flag
    if c.ts and not dq:
        b.flag(c.ts_regular)                         # sh_ts_residual_coding_disabled_flag
    b.trailing()                                     # byte_alignment()


# ---------------------------------------------------------------------------------------------------------------------
# slice data
# ---------------------------------------------------------------------------------------------------------------------
SCAN4 = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0), (0, 3), (1, 2), (2, 1), (3, 0), (1, 3), (2, 2), (3, 1), (2, 3), (3, 2), (3, 3)]      # diagonal scan of a 4x4 coefficient group: scan position -> (x, y)
PREFIX_CTX = [0, 0, 0, 3, 6, 10, 15, 21]


class PictureWriter:
    def __init__(self, c, cab, rng, pic=None):
        self.c, self.cab, self.rng = c, cab, rng
        self.pic = pic or dict(type="I", l0=[], l1=[])
        self.st = self.pic["type"]
        w4, h4 = c.width >> 2, c.height >> 2
        self.cu_w = [[0] * w4 for _ in range(h4)]      # luma width / height of the CU that covers a 4x4 cell (0: not coded yet)
        self.cu_h = [[0] * w4 for _ in range(h4)]
        self.cu_f = [[0] * w4 for _ in range(h4)]      # per cell: 1 skip, 2 intra, 4 affine (context of the flags of later CUs)
        self.cu_q = [[0] * w4 for _ in range(h4)]      # quad-tree depth of the CU
        # dual tree (I slices of a sequence with sps_qtbtt_dual_tree_intra_flag): the chroma tree has neighbours of its own
        self.dual = bool(c.dual_tree) and self.st == "I"
        self.maps = {"single": (self.cu_w, self.cu_h, self.cu_q), "luma": (self.cu_w, self.cu_h, self.cu_q),
                     "chroma": ([[0] * w4 for _ in range(h4)], [[0] * w4 for _ in range(h4)], [[0] * w4 for _ in range(h4)])}
        self.tree = "single"
        self.sl = None                                 # the slice being written (pictures of several slices): dict(idx, type, qp, sao, alf, dq, lmcs)
        self.P = c.partition
        self.stats = dict(cus=0, split=0, cbf=0, coefs=0, skip=0, merge=0, amvp=0, intra=0)
        self.dqp_coded, self.cu_ciip, self.cu = False, False, dict(w=0, h=0, sbt=None, isp=0)
        # intra block copy: the block vector (whole luma samples) of the IBC CU that covers a cell, the history of block vectors of the CTU row
        # (MotionHist::motionLutIbc, emptied at the first CTU of a row: DecCu::TaskDeriveCtuMotionInfo :66-74)
        self.ibc_bv, self.ibc_lut = {}, []
        # dual tree: what CU::checkCCLMAllowed (UnitTools.cpp:3439-3492) asks about - the splits from the tree's 64x64 node down to the CU being written, and per luma cell
        # the depth, quad-tree depth and ISP mode of the luma CU that covers it
        self.path, self.luma_info, self.cur_xy, self.last_isp = [], {}, (0, 0), 0
        assert not (c.ibc and self.P), "this writer's IBC knows one slice, one tile"

    # is the 4x4 cell at luma position (nx, ny) a neighbour the block at (x, y) may look at?  (same slice and tile: CodingStructure::getCURestricted)
    def dq_on(self):
        return self.c.dep_quant and (self.sl is None or self.sl["dq"])

    def avail(self, x, y, nx, ny):
        if nx < 0 or ny < 0:
            return False
        if not self.P:
            return True
        l2 = self.c.log2_ctu
        return self.P["ctu"][(x >> l2, y >> l2)] == self.P["ctu"][(nx >> l2, ny >> l2)]

    def picture(self, ctus=None, cab=None, sl=None):
        S = 1 << self.c.log2_ctu
        if cab is not None:                                                    # one of several slices: its own arithmetic codeword, type and switches
            self.cab, self.sl, self.st = cab, sl, sl["type"]
            self.dual = bool(self.c.dual_tree) and self.st == "I"
        if ctus is None:
            ctus = [(rx, ry) for ry in range(self.c.height // S) for rx in range(self.c.width // S)]
        for (rx, ry) in ctus:
            x, y = rx * S, ry * S
            if rx == 0:
                self.ibc_lut = []
            if True:
                sao_on = self.c.sao and (self.sl is None or any(self.sl["sao"]))
                if sao_on:
                    self.sao(x, y)
                alf = self.sl.get("alf") if self.sl else self.pic.get("alf")
                if self.c.alf and alf["on"]:
                    self.alf(x >> self.c.log2_ctu, y >> self.c.log2_ctu)
                self.dqp_coded = False                                         # (quantisation group = CTU)
                if self.dual:
                    self.dual_ctu(x, y, S, 0, 0)
                else:
                    self.coding_tree(x, y, S, S)
        self.cab.trm(1)                              # end_of_slice_one_bit

    # -- ALF controls of a CTU (CABACReader::readAlf :391-467): on / off per component, filter set, chroma alternative, CC-ALF filter
    def alf(self, rx, ry):
        cab, rng, a = self.cab, self.rng, (self.sl["alf"] if self.sl else self.pic["alf"])
        if not hasattr(self, "alf_ctu"):
            self.alf_ctu = {}
        S = 1 << self.c.log2_ctu
        left = self.alf_ctu.get((rx - 1, ry), [0] * 5) if self.avail(rx * S, ry * S, rx * S - 1, ry * S) else [0] * 5
        above = self.alf_ctu.get((rx, ry - 1), [0] * 5) if self.avail(rx * S, ry * S, rx * S, ry * S - 1) else [0] * 5
        cur = [0] * 5
        for comp in range(1 if self.c.mono else 3):
            if comp and not a["cb" if comp == 1 else "cr"]:
                continue
            on = rng.random() < 0.7
            cab.bin(1 if on else 0, "ctbAlfFlag", comp * 3 + left[comp] + above[comp])      # alf_ctb_flag
            cur[comp] = 1 if on else 0
            if comp == 0 and on:
                n = len(a["luma_aps"])
                use_aps = n > 0 and rng.random() < 0.6
                if n > 0:
                    cab.bin(1 if use_aps else 0, "AlfUseTemporalFilt", 0)      # alf_use_aps_flag
                if use_aps:
                    if n > 1:
                        self.trunc_bin(rng.randrange(0, n), n)                 # alf_luma_prev_filter_idx
                else:
                    self.trunc_bin(rng.randrange(0, 16), 16)                   # alf_luma_fixed_filter_idx
            if comp and on:
                alt = rng.randrange(0, a["nalt"])
                for k in range(a["nalt"] - 1):                                 # alf_ctb_filter_alt_idx: unary, context coded
                    cab.bin(1 if alt > k else 0, "ctbAlfAlternative", comp - 1)
                    if alt <= k:
                        break
        for comp in (1, 2):
            key = "cc_cb" if comp == 1 else "cc_cr"
            if self.c.ccalf and a[key] is not None:
                cnt = a["ncc"][comp - 1]
                idc = rng.randrange(0, cnt + 1)
                ctx = (1 if left[2 + comp] else 0) + (1 if above[2 + comp] else 0) + (3 if comp == 2 else 0)
                cab.bin(1 if idc else 0, "CcAlfFilterControlFlag", ctx)        # alf_ctb_cc_cb_idc / cr_idc
                if idc:
                    v = 1
                    while v != cnt:
                        cab.ep(1 if idc > v else 0)
                        if idc <= v:
                            break
                        v += 1
                cur[2 + comp] = idc
        self.alf_ctu[(rx, ry)] = cur

    # -- sao( rx, ry ) (CABACReader::sao): merge left / above, else type, four offsets, band position or edge class per component
    def sao(self, x, y):
        cab, rng = self.cab, self.rng
        if self.avail(x, y, x - 1, y):
            m = rng.random() < 0.25
            cab.bin(1 if m else 0, "SaoMergeFlag", 0)                          # sao_merge_left_flag
            if m:
                return
        if self.avail(x, y, x, y - 1):
            m = rng.random() < 0.25
            cab.bin(1 if m else 0, "SaoMergeFlag", 0)                          # sao_merge_up_flag
            if m:
                return
        mx = (1 << (min(self.c.bit_depth, 10) - 5)) - 1
        mode_cb = 0
        for comp in range(1 if self.c.mono else 3):
            if self.sl is not None and not self.sl["sao"][1 if comp else 0]:
                continue                                                       # (the slice uses SAO for the other channel type only)
            if comp != 2:
                mode = rng.choice([0, 1, 2])                                   # off, band offset, edge offset
                cab.bin(1 if mode else 0, "SaoTypeIdx", 0)                     # sao_type_idx_luma / chroma
                if mode:
                    cab.ep(1 if mode == 2 else 0)
                if comp == 1:
                    mode_cb = mode
            else:
                mode = mode_cb
            if not mode:
                continue
            offs = [rng.randrange(0, min(mx, 6) + 1) for _ in range(4)]
            for o in offs:
                self.unary_eq(o, mx)                                           # sao_offset_abs
            if mode == 1:
                for o in offs:
                    if o:
                        cab.ep(rng.randrange(0, 2))                            # sao_offset_sign_flag
                cab.eps(rng.randrange(0, 32), 5)                               # sao_band_position
            elif comp != 2:
                cab.eps(rng.randrange(0, 4), 2)                                # sao_eo_class_luma / chroma

    # -- which splits a block may take (Partitioner::canSplit, UnitPartitioner.cpp:281-385, for a single tree inside the picture; the smallest coding block is
    # 8x8, so none of the mode-type conditions - local dual tree - ever holds)
    def can_split(self, w, h, mt_depth, last, idx):
        c = self.c
        chroma = self.tree == "chroma"
        min_qt, min_bt, max_btt = 1 << (c.log2_min_qt_c if chroma else c.log2_min_qt), 1 << c.log2_min_cb, 1 << c.log2_max_btt
        can_qt = last in ("ctu", "qt") and w > min_qt and not (chroma and (w >> 1) <= 4)
        bh = bv = th = tv = False
        if mt_depth < c.mtt_depth and (w > min_bt or h > min_bt) and w <= max_btt and h <= max_btt:
            bh = bv = True
            if last in ("th", "tv") and idx == 1:                              # the middle part of a ternary split is not halved in the same direction
                if last == "th":
                    bh = False
                else:
                    bv = False
            bh = bh and h > min_bt and (w <= 64 or h > 64)
            bv = bv and w > min_bt and (w > 64 or h <= 64)
            if w <= 64 and h <= 64:
                th, tv = h > 2 * min_bt, w > 2 * min_bt
            if chroma:                                                         # (sizes of the chroma block: no chroma block below 16 samples, none 2 wide)
                cw, ch = w >> 1, h >> 1
                bh = bh and cw * ch > 16
                th = th and cw * ch > 32
                bv = bv and cw * ch > 16 and cw > 4
                tv = tv and cw * ch > 32 and cw > 8
        return can_qt, bh, bv, th, tv

    # -- a CTU of a dual-tree I slice (CABACReader::dt_implicit_qt_split :172): blocks above 64x64 are quartered without any syntax, then a luma tree and a chroma
    # tree per 64x64 (or per CTU)
    def dual_ctu(self, x, y, size, qt_depth, idx):
        if size > 64:
            hs = size >> 1
            for i, (dx, dy) in enumerate(((0, 0), (hs, 0), (0, hs), (hs, hs))):
                self.dual_ctu(x + dx, y + dy, hs, qt_depth + 1, i)
            return
        last = "qt" if qt_depth else "ctu"
        for tree in ("luma", "chroma"):
            self.tree = tree
            self.path = []
            self.coding_tree(x, y, size, size, qt_depth, 0, last, idx)
        self.tree = "single"

    # -- coding_tree: split_cu_flag, split_qt_flag, mtt_split_cu_vertical_flag, mtt_split_cu_binary_flag where more than one choice exists
    # (CABACReader::split_cu_mode :679-830)
    def coding_tree(self, x, y, w, h, qt_depth=0, mt_depth=0, last="ctu", idx=0):
        cab, rng = self.cab, self.rng
        can_qt, bh, bv, th, tv = self.can_split(w, h, mt_depth, last, idx)
        num_hor, num_ver = bh + th, bv + tv
        num_split = 2 * can_qt + num_hor + num_ver
        mode = None
        m_w, m_h, m_q = self.maps[self.tree]
        if num_split:
            left_h = m_h[y >> 2][(x >> 2) - 1] if self.avail(x, y, x - 1, y) else 0
            above_w = m_w[(y >> 2) - 1][x >> 2] if self.avail(x, y, x, y - 1) else 0
            p = self.c.p_split if can_qt else self.c.p_mtt
            split = rng.random() < p
            ctx = (1 if (left_h and left_h < h) else 0) + (1 if (above_w and above_w < w) else 0) + (0, 0, 0, 3, 3, 6, 6)[num_split]
            cab.bin(1 if split else 0, "SplitFlag", ctx)                       # split_cu_flag
            if split:
                can_btt = num_hor or num_ver
                is_qt = can_qt
                if can_qt and can_btt:
                    is_qt = rng.random() < 0.5
                    lq = m_q[y >> 2][(x >> 2) - 1] if self.avail(x, y, x - 1, y) else -1
                    aq = m_q[(y >> 2) - 1][x >> 2] if self.avail(x, y, x, y - 1) else -1
                    cab.bin(1 if is_qt else 0, "SplitQtFlag", (1 if lq > qt_depth else 0) + (1 if aq > qt_depth else 0) + (0 if qt_depth < 2 else 3))      # split_qt_flag
                if is_qt:
                    mode = "qt"
                else:
                    ver = bool(num_ver)
                    if num_ver and num_hor:
                        ver = rng.random() < 0.5
                        chv = 0
                        if num_ver == num_hor:
                            if left_h and above_w:
                                dep_a, dep_l = w >> (above_w.bit_length() - 1), h >> (left_h.bit_length() - 1)
                                chv = 0 if dep_a == dep_l else (1 if dep_a < dep_l else 2)
                        else:
                            chv = 3 if num_ver < num_hor else 4
                        cab.bin(1 if ver else 0, "SplitHvFlag", chv)           # mtt_split_cu_vertical_flag
                    can14, is12 = (tv, bv) if ver else (th, bh)
                    if is12 and can14:
                        is12 = rng.random() < 0.5
                        cab.bin(1 if is12 else 0, "Split12Flag", (1 if mt_depth <= 1 else 0) + (2 if ver else 0))      # mtt_split_cu_binary_flag
                    mode = ("bv" if is12 else "tv") if ver else ("bh" if is12 else "th")
        if mode:
            self.stats["split"] += 1
            if mode == "qt":
                parts = [(x, y, w >> 1, h >> 1), (x + (w >> 1), y, w >> 1, h >> 1), (x, y + (h >> 1), w >> 1, h >> 1), (x + (w >> 1), y + (h >> 1), w >> 1, h >> 1)]
            elif mode == "bh":
                parts = [(x, y, w, h >> 1), (x, y + (h >> 1), w, h >> 1)]
            elif mode == "bv":
                parts = [(x, y, w >> 1, h), (x + (w >> 1), y, w >> 1, h)]
            elif mode == "th":
                parts = [(x, y, w, h >> 2), (x, y + (h >> 2), w, h >> 1), (x, y + 3 * (h >> 2), w, h >> 2)]
            else:
                parts = [(x, y, w >> 2, h), (x + (w >> 2), y, w >> 1, h), (x + 3 * (w >> 2), y, w >> 2, h)]
            self.path.append(mode)
            for i, (px, py, pw, ph) in enumerate(parts):
                if mode == "qt":
                    self.coding_tree(px, py, pw, ph, qt_depth + 1, 0, "qt", i)
                else:
                    self.coding_tree(px, py, pw, ph, qt_depth, mt_depth + 1, mode, i)
            self.path.pop()
            return
        self.cur_xy, self.last_isp = (x, y), 0
        f = self.coding_unit(x, y, w, h)
        if self.tree != "chroma":
            for yy in range(y >> 2, (y + h) >> 2):
                for xx in range(x >> 2, (x + w) >> 2):
                    self.luma_info[(xx, yy)] = (qt_depth + mt_depth, qt_depth, self.last_isp)
        for yy in range(y >> 2, (y + h) >> 2):
            for xx in range(x >> 2, (x + w) >> 2):
                m_w[yy][xx] = w
                m_h[yy][xx] = h
                m_q[yy][xx] = qt_depth
                if self.tree != "chroma":
                    self.cu_f[yy][xx] = f

    # -- coding_unit of an I slice, single tree: intra luma mode, intra chroma mode, transform tree
    def coding_unit(self, x, y, w, h):
        cab, rng = self.cab, self.rng
        self.stats["cus"] += 1
        if self.st != "I":
            return self.coding_unit_inter(x, y, w, h)
        if self.c.ibc and self.tree != "chroma":
            return self.coding_unit_ibc(x, y, w, h)
        return self.intra_cu(x, y, w, h)

    # ---- intra block copy (I slices; sps_ibc_enabled_flag with ONE merge candidate).  Unlike every other syntax element of this writer the block vector cannot be
    # random: the reference block has to be reconstructed already, so the writer follows the decoder's derivation of candidate 0 - the IBC CU left of the bottom-left
    # sample, else the one above the top-right sample, else the newest entry of the row's history, else zero (PU::getIBCMergeCandidates, UnitTools.cpp:728-830;
    # history: DecCu.cpp:884-900, MotionInfo.h:242) - and codes the difference to a vector it has checked: reference block inside the current CTU or the one / two
    # CTUs left of it, every 4x4 cell of it coded before this CU, even components (chroma blocks of 4:2:0 then start on a chroma sample)
    def ibc_candidate(self, x, y, w, h):
        for (nx, ny) in ((x - 1, y + h - 1), (x + w - 1, y - 1)):
            if nx >= 0 and ny >= 0 and self.avail(x, y, nx, ny) and self.cu_w[ny >> 2][nx >> 2] and (self.cu_f[ny >> 2][nx >> 2] & 16):
                return self.ibc_bv[(nx >> 2, ny >> 2)]
        return self.ibc_lut[-1] if self.ibc_lut else None

    def ibc_valid(self, x, y, w, h, bv):
        S = 1 << self.c.log2_ctu
        rx, ry = x + bv[0], y + bv[1]
        # the CTU row of the CU; the CU's CTU and the one (CTU 128) or two CTUs left of it: all of that is still in the decoder's buffer of reconstructed samples
        # (256 x 128 luma samples for CTU 128, CodingStructure.cpp:543 - twice what the standard's virtual buffer holds, so the left CTU is there whole)
        left = self.ibc_left_ctus()
        if (bv[0] | bv[1]) & 1 or rx < max(0, (x & -S) - left * S) or ry < (y & -S) or rx + w > (x & -S) + S or ry + h > (y & -S) + S:
            return False
        return all(self.cu_w[cy][cx] for cy in range(ry >> 2, (ry + h + 3) >> 2) for cx in range(rx >> 2, (rx + w + 3) >> 2))

    def ibc_left_ctus(self):
        return 1 if self.c.log2_ctu == 7 else 2

    def ibc_pick(self, x, y, w, h):
        S, rng = 1 << self.c.log2_ctu, self.rng
        x0, y0 = x & -S, y & -S
        xl = max(0, x0 - self.ibc_left_ctus() * S)
        for _ in range(24):
            rx, ry = xl + 2 * rng.randrange(0, (x0 + S - w - xl) // 2 + 1), y0 + 2 * rng.randrange(0, (S - h) // 2 + 1)
            if self.ibc_valid(x, y, w, h, (rx - x, ry - y)):
                return (rx - x, ry - y)
        return None

    def mvd_write(self, hv, vv):                                               # mvd_coding with given components (CABACReader::mvd_coding)
        cab = self.cab
        cab.bin(1 if hv else 0, "Mvd", 0)
        cab.bin(1 if vv else 0, "Mvd", 0)
        if hv:
            cab.bin(1 if abs(hv) > 1 else 0, "Mvd", 1)
        if vv:
            cab.bin(1 if abs(vv) > 1 else 0, "Mvd", 1)
        for a in (hv, vv):
            if a:
                if abs(a) > 1:
                    self.rem_abs_ep(abs(a) - 2, 1, 0)
                cab.ep(1 if a < 0 else 0)

    def ibc_decide(self, x, y, w, h, p):
        rng = self.rng
        mode, bv, pred = None, None, None
        if w <= 64 and h <= 64 and rng.random() < p:
            pred = self.ibc_candidate(x, y, w, h)
            pred_ok = pred is not None and self.ibc_valid(x, y, w, h, pred)
            r = rng.random()
            if pred_ok and r < 0.45:
                mode, bv = ("skip" if r < 0.2 else "merge"), pred
            else:
                bv = self.ibc_pick(x, y, w, h)
                mode = "amvp" if bv else None
        return mode, bv, pred

    # everything of an IBC CU behind pred_mode_ibc_flag (general_merge_flag .. the transform tree), then its vector into the cell map and the history
    def ibc_rest(self, x, y, w, h, mode, bv, pred):
        cab, rng, c = self.cab, self.rng, self.c
        if mode != "skip":
            cab.bin(1 if mode == "merge" else 0, "MergeFlag", 0)               # general_merge_flag (merge_idx: one candidate)
            root = True
            if mode == "amvp":
                p = pred if pred is not None else (0, 0)
                dh, dv = bv[0] - p[0], bv[1] - p[1]
                self.mvd_write(dh, dv)                                         # (mvp_l0_flag: one candidate)
                if c.amvr and (dh or dv):
                    cab.bin(0, "ImvFlag", 1)                                   # amvr_precision_idx: whole samples (CABACReader::amvr_mode :1001-1025)
                root = rng.random() < 0.7
                cab.bin(1 if root else 0, "QtRootCbf", 0)                      # cu_coded_flag
            if root:
                self.cu = dict(intra=False, ibc=True, w=w, h=h, isp=0, mip=False, viol=False, lfnst_last=False, mts_last=False, sbt=None)
                self.transform_tree(w, h, intra=False, root=True)
                self.lfnst_and_mts()
        self.stats["ibc"] = self.stats.get("ibc", 0) + 1
        for cy in range(y >> 2, (y + h) >> 2):
            for cx in range(x >> 2, (x + w) >> 2):
                self.ibc_bv[(cx, cy)] = bv
        if w * h > 16:
            if bv in self.ibc_lut:
                self.ibc_lut.remove(bv)
            elif len(self.ibc_lut) == 5:
                self.ibc_lut.pop(0)
            self.ibc_lut.append(bv)
        return 16 | (1 if mode == "skip" else 0)

    def coding_unit_ibc(self, x, y, w, h):
        cab, c = self.cab, self.c
        left, above = self.neigh(x, y)
        mode, bv, pred = self.ibc_decide(x, y, w, h, c.p_ibc)
        cab.bin(1 if mode == "skip" else 0, "SkipFlag", (left & 1) + (above & 1))      # cu_skip_flag (every luma CU of an I slice once IBC is enabled; a skipped one is an IBC merge CU)
        if mode != "skip":
            if w <= 64 and h <= 64:
                cab.bin(1 if mode else 0, "IBCFlag", (1 if left & 16 else 0) + (1 if above & 16 else 0))      # pred_mode_ibc_flag
            if not mode:
                return self.intra_cu(x, y, w, h)
        return self.ibc_rest(x, y, w, h, mode, bv, pred)

    # -- an intra CU: modes, transform tree, lfnst_idx, mts_idx (CABACReader::cu_pred_data, cu_residual :1404-1456)
    def intra_cu(self, x, y, w, h):
        if self.tree == "chroma":                                              # the chroma CU of a dual tree: a chroma mode, chroma transform units, an LFNST index of its own
            self.chroma_mode(w, h)
            self.cu = dict(intra=True, w=w, h=h, isp=0, mip=False, viol=False, lfnst_last=False, mts_last=False, sbt=None, bdpcm=0, bdpcm_c=self.bdpcm_c)
            self.transform_tree(w, h, intra=True, root=True)
            self.lfnst_and_mts()
            return 2
        self.bdpcm_c = 0
        info = self.intra_modes(x, y, w, h)
        self.last_isp = info["isp"]
        self.cu = dict(intra=True, w=w, h=h, isp=info["isp"], mip=info["mip"], viol=False, lfnst_last=False, mts_last=False, sbt=None, bdpcm=info["bdpcm"], bdpcm_c=self.bdpcm_c)
        self.transform_tree(w, h, intra=True, root=True)
        self.lfnst_and_mts()
        return 2 | (8 if info["mip"] else 0)

    def lfnst_and_mts(self):
        cab, rng, c, cu = self.cab, self.rng, self.c, self.cu
        mx = 1 << c.log2_max_tb
        lfnst = 0
        if c.lfnst and cu["intra"] and not (cu["mip"] and not (cu["w"] >= 16 and cu["h"] >= 16)) and cu["w"] <= mx and cu["h"] <= mx \
                and not cu["viol"] and (cu["lfnst_last"] or cu["isp"]) and not cu.get("ts_any"):
            lfnst = rng.choice([0, 1, 2])
            cab.bin(1 if lfnst else 0, "LFNSTIdx", 0 if self.tree == "single" else 1)      # lfnst_idx (context: single or separate tree)
            if lfnst:
                cab.bin(lfnst - 1, "LFNSTIdx", 2)
        if c.mts and not cu.get("ibc") and self.tree != "chroma" and cu["w"] <= 32 and cu["h"] <= 32 and not cu["isp"] and not cu["sbt"] and not cu.get("ts_first") and not cu.get("bdpcm") and cu["mts_last"] and lfnst == 0 and not cu.get("mts_viol"):
            m = rng.choice([0, 0, 1, 2, 3, 4])
            cab.bin(1 if m else 0, "MTSIndex", 0)                                # mts_idx
            for k in range(1, 4):
                if m < k:
                    break
                cab.bin(1 if m > k else 0, "MTSIndex", k)

    def neigh(self, x, y):
        left = self.cu_f[y >> 2][(x >> 2) - 1] if self.avail(x, y, x - 1, y) else 0
        above = self.cu_f[(y >> 2) - 1][x >> 2] if self.avail(x, y, x, y - 1) else 0
        return left, above

    # -- coding_unit of a P / B slice (CABACReader::coding_unit, prediction_unit, merge_data)
    def coding_unit_inter(self, x, y, w, h):
        cab, rng, c = self.cab, self.rng, self.c
        left, above = self.neigh(x, y)
        self.cu_ciip = False
        ibc_ctx = (1 if left & 16 else 0) + (1 if above & 16 else 0)
        ibc_asked = c.ibc and w <= 64 and h <= 64                              # (P / B slices of a sequence with IBC: the flag follows a skip flag of 1 and a pred_mode_flag of 0)
        imode, ibv, ipred = self.ibc_decide(x, y, w, h, 0.5 * c.p_ibc) if ibc_asked else (None, None, None)
        skip = imode == "skip" or (imode is None and rng.random() < c.p_skip)
        cab.bin(1 if skip else 0, "SkipFlag", (left & 1) + (above & 1))        # cu_skip_flag
        if skip:
            if ibc_asked:
                cab.bin(1 if imode else 0, "IBCFlag", ibc_ctx)                 # pred_mode_ibc_flag of a skipped CU (CABACReader::cu_skip_flag :945-975)
            if imode:
                return self.ibc_rest(x, y, w, h, imode, ibv, ipred)
            self.stats["skip"] += 1
            aff = self.merge_data(x, y, w, h, True, left, above)
            return 1 | (4 if aff else 0)
        intra = imode is None and rng.random() < c.p_intra
        cab.bin(1 if intra else 0, "PredMode", 1 if ((left & 2) or (above & 2)) else 0)      # pred_mode_flag
        if intra:
            self.stats["intra"] += 1
            return self.intra_cu(x, y, w, h)
        if ibc_asked:
            cab.bin(1 if imode else 0, "IBCFlag", ibc_ctx)                     # pred_mode_ibc_flag (CABACReader::pred_mode :1082-1093)
        if imode:
            return self.ibc_rest(x, y, w, h, imode, ibv, ipred)
        merge = rng.random() < c.p_merge
        cab.bin(1 if merge else 0, "MergeFlag", 0)                             # general_merge_flag
        aff = False
        if merge:
            self.stats["merge"] += 1
            aff = self.merge_data(x, y, w, h, False, left, above)
            root = True                                                        # (a merge CU that is not skipped has a residual)
        else:
            self.stats["amvp"] += 1
            aff = self.amvp(w, h, left, above)
            root = rng.random() < 0.7
            cab.bin(1 if root else 0, "QtRootCbf", 0)                          # cu_coded_flag
        if root:
            self.cu = dict(intra=False, w=w, h=h, isp=0, mip=False, viol=False, lfnst_last=False, mts_last=False, sbt=None)
            mx = 1 << c.log2_max_tb
            if c.sbt and not self.cu_ciip and w <= mx and h <= mx:
                # cu_sbt_flag, cu_sbt_quad_flag, cu_sbt_horizontal_flag, cu_sbt_pos_flag (CABACReader::sbt_mode :1476)
                vh, hh, vq, hq = w >= 8, h >= 8, w >= 16, h >= 16
                if vh or hh:
                    # (this writer's residual coding knows 4x4 coefficient groups: only splits whose residual part stays 8 luma samples wide and high, chroma 4)
                    opts = [(q, hz) for q in (False, True) for hz in (False, True) if ((hq if hz else vq) if q else (hh if hz else vh))
                            and (w if hz else w // (4 if q else 2)) >= 8 and (h // (4 if q else 2) if hz else h) >= 8]
                    use = bool(opts) and rng.random() < 0.35
                    cab.bin(1 if use else 0, "SbtFlag", 1 if w * h <= 256 else 0)
                    if use:
                        quad, hor = rng.choice(opts)
                        if (vh or hh) and (vq or hq):
                            cab.bin(1 if quad else 0, "SbtQuadFlag", 0)
                        av, ah = (vq, hq) if quad else (vh, hh)
                        if av and ah:
                            cab.bin(1 if hor else 0, "SbtHorFlag", 0 if w == h else (1 if w < h else 2))
                        pos1 = rng.random() < 0.5
                        cab.bin(1 if pos1 else 0, "SbtPosFlag", 0)
                        self.cu["sbt"] = (quad, hor, pos1)
            self.transform_tree(w, h, intra=False, root=True)
            self.lfnst_and_mts()
        return 4 if aff else 0

    def merge_idx(self, name, num_minus1):
        idx = self.rng.randrange(0, num_minus1 + 1)
        if num_minus1 > 0:
            self.cab.bin(1 if idx else 0, name, 0)
            if idx:
                for k in range(1, idx):
                    self.cab.ep(1)
                if idx < num_minus1:
                    self.cab.ep(0)

    def unary_eq(self, v, mx):                                                 # unary_max_eqprob
        for _ in range(v):
            self.cab.ep(1)
        if v < mx:
            self.cab.ep(0)

    def merge_data(self, x, y, w, h, skip, left, above):
        cab, rng, c = self.cab, self.rng, self.c
        if c.max_aff_merge > 0 and w >= 8 and h >= 8:
            aff = rng.random() < 0.25
            cab.bin(1 if aff else 0, "SubblockMergeFlag", (1 if left & 4 else 0) + (1 if above & 4 else 0))      # merge_subblock_flag
            if aff:
                self.merge_idx("AffMergeIdx", c.max_aff_merge - 1)             # merge_subblock_idx
                return True
        ciip_av = c.ciip and not skip and w < 128 and h < 128 and w * h >= 64
        geo_av = c.gpm and self.st == "B" and 8 <= w <= 64 and 8 <= h <= 64 and w < 8 * h and h < 8 * w
        regular = True
        if geo_av or ciip_av:
            regular = rng.random() < 0.6
            cab.bin(1 if regular else 0, "RegularMergeFlag", 0 if skip else 1)      # regular_merge_flag
        if regular:
            mmvd = False
            if c.mmvd:
                mmvd = rng.random() < 0.3
                cab.bin(1 if mmvd else 0, "MmvdFlag", 0)                       # mmvd_merge_flag
            if mmvd:
                cab.bin(rng.randrange(0, 2), "MmvdMergeIdx", 0)                # mmvd_cand_flag
                step = rng.randrange(0, 8)
                cab.bin(1 if step else 0, "MmvdStepMvpIdx", 0)                 # mmvd_distance_idx
                if step:
                    for k in range(1, step):
                        cab.ep(1)
                    if step < 7:
                        cab.ep(0)
                cab.eps(rng.randrange(0, 4), 2)                                # mmvd_direction_idx
            else:
                self.merge_idx("MergeIdx", 5)                                  # merge_idx (MaxNumMergeCand 6)
            return False
        ciip = False
        if geo_av and ciip_av:
            ciip = rng.random() < 0.5
            cab.bin(1 if ciip else 0, "CiipFlag", 0)                           # ciip_flag
        elif ciip_av:
            ciip = True
        if ciip:
            self.cu_ciip = True
            self.merge_idx("MergeIdx", 5)
            return False
        # geometric partitioning: split direction, two different candidates out of MaxNumGpmMergeCand = 6
        self.trunc_bin(rng.randrange(0, 64), 64)                               # merge_gpm_partition_idx
        c0 = rng.randrange(0, 6)
        cab.bin(1 if c0 else 0, "MergeIdx", 0)                                 # merge_gpm_idx0
        if c0:
            self.unary_eq(c0 - 1, 4)
        c1 = rng.randrange(0, 5)
        cab.bin(1 if c1 else 0, "MergeIdx", 0)                                 # merge_gpm_idx1
        if c1:
            self.unary_eq(c1 - 1, 3)
        return False

    def mvd(self):
        cab, rng, m = self.cab, self.rng, self.c.max_mvd
        h, v = rng.randrange(-m, m + 1), rng.randrange(-m, m + 1)
        if rng.random() < 0.3:
            h = 0
        if rng.random() < 0.3:
            v = 0
        cab.bin(1 if h else 0, "Mvd", 0)                                       # abs_mvd_greater0_flag[0 / 1]
        cab.bin(1 if v else 0, "Mvd", 0)
        if h:
            cab.bin(1 if abs(h) > 1 else 0, "Mvd", 1)                          # abs_mvd_greater1_flag
        if v:
            cab.bin(1 if abs(v) > 1 else 0, "Mvd", 1)
        for a in (h, v):
            if a:
                if abs(a) > 1:
                    self.rem_abs_ep(abs(a) - 2, 1, 0)                          # abs_mvd_minus2: EG1
                cab.ep(1 if a < 0 else 0)                                      # mvd_sign_flag
        return bool(h or v)

    def rem_abs_ep(self, v, rice, cutoff):
        """inverse of BinDecoder::decodeRemAbsEP for values far below the escape length"""
        p = 0
        while True:
            off = (p << rice) if p < cutoff else ((((1 << (p - cutoff)) + cutoff - 1)) << rice)
            length = rice if p < cutoff else rice + (p - cutoff)
            if v < off + (1 << length):
                break
            p += 1
        assert p < 16
        for _ in range(p):
            self.cab.ep(1)
        self.cab.ep(0)
        self.cab.eps(v - off, length)

    def ref_idx(self, n):
        r = self.rng.randrange(0, n)
        if n <= 1:
            return
        self.cab.bin(1 if r else 0, "RefPic", 0)
        if not r:
            return
        if n > 2:
            self.cab.bin(1 if r > 1 else 0, "RefPic", 1)
            if r > 1:
                for idx in range(3, n + 1):
                    if idx == n:
                        break
                    b = 1 if r >= idx else 0
                    self.cab.ep(b)
                    if not b:
                        break

    def amvp(self, w, h, left, above):
        cab, rng, c = self.cab, self.rng, self.c
        n0, n1 = len(self.pic["l0"]), len(self.pic["l1"])
        dirn = 1
        if self.st == "B":
            dirn = rng.choice([1, 2, 3, 3])
            cab.bin(1 if dirn == 3 else 0, "InterDir", 7 - ((w.bit_length() + h.bit_length() - 2 + 1) >> 1))      # inter_pred_idc (no 4x8 / 8x4 blocks here)
            if dirn != 3:
                cab.bin(1 if dirn == 2 else 0, "InterDir", 5)
        aff, six = False, False
        if c.affine and w >= 16 and h >= 16:
            aff = rng.random() < 0.3
            cab.bin(1 if aff else 0, "AffineFlag", (1 if left & 4 else 0) + (1 if above & 4 else 0))      # inter_affine_flag
            if aff:
                six = rng.random() < 0.5
                cab.bin(1 if six else 0, "AffineType", 0)                      # cu_affine_type_flag
        smvd = False
        if c.smvd and dirn == 3 and not aff and self.pic.get("bidir"):
            smvd = rng.random() < 0.4
            cab.bin(1 if smvd else 0, "SmvdFlag", 0)                           # sym_mvd_flag
        nonzero = False
        for lst, n in ((1, n0), (2, n1)):
            if dirn & lst:
                if not (smvd and lst == 2):
                    if not smvd:
                        self.ref_idx(n)                                        # ref_idx_l0 / l1 (symmetric MVD: the nearest pictures on either side)
                    for _ in range(1 + (aff and 1) + (six and 1)):
                        nonzero = self.mvd() or nonzero
                cab.bin(rng.randrange(0, 2), "MVPIdx", 0)                      # mvp_l0_flag / mvp_l1_flag
        # amvr_flag / amvr_precision_idx (CABACReader::amvr_mode :991, affine_amvr_mode :1031): only with a non-zero MVD
        if c.amvr and nonzero:
            if aff:
                v = rng.choice([0, 0, 1, 2])
                cab.bin(1 if v else 0, "ImvFlag", 2)
                if v:
                    cab.bin(v - 1, "ImvFlag", 3)
            else:
                v = rng.choice([0, 0, 1, 2, 3])                                # quarter, (1) integer, (2) four samples, (3) half sample
                cab.bin(1 if v else 0, "ImvFlag", 0)
                if v:
                    cab.bin(0 if v == 3 else 1, "ImvFlag", 4)
                    if v != 3:
                        cab.bin(v - 1, "ImvFlag", 1)
        # bcw_idx (CABACReader::cu_bcw_flag :1180): bi-prediction of at least 256 samples, no explicit weights on the two pictures (this writer: none in the slice)
        if c.bcw and self.st == "B" and dirn == 3 and w * h >= 256 and not c.wp:
            num = 5 if self.pic.get("ldc") else 3
            idx = rng.randrange(0, num)
            cab.bin(1 if idx else 0, "BcwIdx", 0)
            if idx:
                for k in range(1, num - 1):
                    cab.ep(1 if idx > k else 0)
                    if idx <= k:
                        break
        return aff

    # -- intra_luma_pred_mode / intra_chroma_pred_mode (CABACReader :1242-1400, :2541-2576, :3125-3149)
    def intra_modes(self, x, y, w, h):
        cab, rng, c = self.cab, self.rng, self.c
        info = dict(isp=0, mip=False, bdpcm=0)
        done = False
        if c.ts and c.bdpcm and w <= 32 and h <= 32:
            bd = rng.choice([0, 0, 0, 1, 2])
            cab.bin(1 if bd else 0, "BDPCMMode", 0)                            # intra_bdpcm_luma_flag
            if bd:
                cab.bin(bd - 1, "BDPCMMode", 1)                                # intra_bdpcm_luma_dir_flag
                info["bdpcm"] = bd
                if self.tree == "single":
                    self.chroma_mode(w, h)
                return info
        if c.mip:
            left, above = self.neigh(x, y)
            mip = rng.random() < 0.25
            ctx = 3 if (w > 2 * h or h > 2 * w) else (1 if left & 8 else 0) + (1 if above & 8 else 0)
            cab.bin(1 if mip else 0, "MipFlag", ctx)                           # intra_mip_flag
            if mip:
                cab.ep(rng.randrange(0, 2))                                    # intra_mip_transposed_flag
                n = 16 if (w == 4 and h == 4) else (8 if (w == 4 or h == 4 or (w == 8 and h == 8)) else 6)
                self.trunc_bin(rng.randrange(0, n), n)                         # intra_mip_mode
                info["mip"] = True
                done = True
        if not done:
            mrl = 0
            if c.mrl and (y & ((1 << c.log2_ctu) - 1)):
                mrl = rng.choice([0, 0, 0, 1, 2])
                cab.bin(1 if mrl else 0, "MultiRefLineIdx", 0)                 # intra_luma_ref_idx
                if mrl:
                    cab.bin(mrl - 1, "MultiRefLineIdx", 1)
            mx = 1 << c.log2_max_tb
            if c.isp and not mrl and w <= mx and h <= mx and w * h > 16:
                # (only where the partitions stay at least 4 wide / high: the residual writer codes 4x4 coefficient groups)
                cands = [d for d in (1, 2) if (h if d == 1 else w) >= 16]
                isp = rng.choice(cands) if (cands and rng.random() < 0.3) else 0
                cab.bin(1 if isp else 0, "ISPMode", 0)                         # intra_subpartitions_mode_flag
                if isp:
                    cab.bin(isp - 1, "ISPMode", 1)                             # intra_subpartitions_split_flag: 0 horizontal, 1 vertical
                info["isp"] = isp
            mpm = True
            if not mrl:
                mpm = rng.random() < 0.6
                cab.bin(1 if mpm else 0, "IPredMode", 0, sub=0)               # intra_luma_mpm_flag
            if mpm:
                not_planar = True
                if not mrl:
                    not_planar = rng.random() < 0.7
                    cab.bin(1 if not_planar else 0, "IntraLumaPlanarFlag", 0 if info["isp"] else 1)      # intra_luma_not_planar_flag
                if not_planar:
                    idx = rng.randrange(0, 5)                                  # intra_luma_mpm_idx: truncated unary, bypass, cMax 4
                    for k in range(idx):
                        cab.ep(1)
                    if idx < 4:
                        cab.ep(0)
            else:
                self.trunc_bin(rng.randrange(0, 61), 61)                       # intra_luma_mpm_remainder
        if self.tree == "single":
            self.chroma_mode(w, h)
        return info

    def chroma_mode(self, w, h):
        cab, rng, c = self.cab, self.rng, self.c
        self.bdpcm_c = 0
        if c.mono:
            return
        if c.ts and c.bdpcm and (w >> 1) <= 32 and (h >> 1) <= 32:
            bd = rng.choice([0, 0, 0, 1, 2])
            cab.bin(1 if bd else 0, "BDPCMMode", 2)                            # intra_bdpcm_chroma_flag
            if bd:
                cab.bin(bd - 1, "BDPCMMode", 3)                                # intra_bdpcm_chroma_dir_flag
                self.bdpcm_c = bd
                return
        cclm_ok = self.tree == "single"
        if c.cclm and self.tree == "chroma":
            # CU::checkCCLMAllowed: CTUs of 32 always; else the chroma tree's 64x64 node must be split by a quad split, not at all, or horizontally in two with the
            # halves split vertically in two or not at all - and the luma tree's 64x64 node must not start with a binary / ternary split nor be one CU with ISP
            if c.log2_ctu <= 5:
                cclm_ok = True
            else:
                s1 = self.path[0] if self.path else None
                s2 = self.path[1] if len(self.path) > 1 else None
                cclm_ok = s1 == "qt" or s1 is None or (s1 == "bh" and s2 in ("bv", None))
                if cclm_ok:
                    d64 = 1 if c.log2_ctu == 7 else 0
                    d, q, isp = self.luma_info[(self.cur_xy[0] >> 2, self.cur_xy[1] >> 2)]
                    if (d > d64 and q == d64) or (d == d64 and isp):
                        cclm_ok = False
        if c.cclm and cclm_ok:
            lm = rng.random() < 0.3
            cab.bin(1 if lm else 0, "CclmModeFlag", 0)                         # cclm_mode_flag
            if lm:
                k = rng.randrange(0, 3)
                cab.bin(1 if k else 0, "CclmModeIdx", 0)                       # cclm_mode_idx: LM, MDLM_L, MDLM_T
                if k:
                    cab.ep(k - 1)
                return
        if rng.random() < 0.5:
            cab.bin(0, "IPredMode", 0, sub=1)                                  # intra_chroma_pred_mode: derived mode
        else:
            cab.bin(1, "IPredMode", 0, sub=1)
            cab.eps(rng.randrange(0, 4), 2)

    def trunc_bin(self, v, n):                                                 # xReadTruncBinCode
        thresh = n.bit_length() - 1
        val = 1 << thresh
        b = n - val
        if v < val - b:
            self.cab.eps(v, thresh)
        else:
            self.cab.eps(v + val - b, thresh + 1)

    def transform_tree(self, w, h, intra, root):
        mx = 1 << self.c.log2_max_tb
        if w > mx or h > mx:                                                   # TU_MAX_TR_SPLIT: halved in every direction that is too long, z order
            nw, nh = (w >> 1 if w > mx else w), (h >> 1 if h > mx else h)
            for _ in range((w // nw) * (h // nh)):
                self.transform_tree(nw, nh, intra, False)
            return
        sbt = self.cu.get("sbt") if (root and not intra) else None
        if sbt:
            # sub-block transform: the CU in two transform units, a half or a quarter of it carries the residual, the other one nothing at all
            quad, hor, pos1 = sbt
            frac = 4 if quad else 2
            rw, rh = (w, h // frac) if hor else (w // frac, h)
            for i in range(2):
                if i == (1 if pos1 else 0):
                    self.transform_unit(rw, rh, False, False, sbt=True)
            return
        isp = self.cu["isp"] if (intra and root) else 0
        if isp:
            # intra sub-partitions: four transform units of a quarter of the height (1) / width (2) - CUs whose partitions would be thinner than 4 do not take ISP
            # here -, luma coded flags with the previous partition's flag as context, the last one inferred when none before it is set; chroma with the last
            n, pw, ph = 4, (w if isp == 1 else w >> 2), (h >> 2 if isp == 1 else h)
            prev, any_cbf = False, False
            for i in range(n):
                last = i == n - 1
                prev = self.transform_unit(pw, ph, True, False, isp=(i, last, prev, any_cbf), cw=w >> 1, ch=h >> 1)
                any_cbf = any_cbf or prev
            return
        self.transform_unit(w, h, intra, root)

    def transform_unit(self, w, h, intra, depth0, isp=None, cw=None, ch=None, sbt=False):
        cab, rng, c = self.cab, self.rng, self.c
        chroma = (isp is None or isp[1]) and self.tree != "luma" and not c.mono               # (ISP: the unsplit chroma blocks come with the last partition; dual tree: none in the luma tree)
        cb = cr = False
        if chroma:
            cb = rng.random() < c.p_cbf_chroma
            cr = rng.random() < c.p_cbf_chroma
            bdc = self.cu.get("bdpcm_c")
            cab.bin(1 if cb else 0, "QtCbf", 1 if bdc else 0, sub=1)           # tu_cb_coded_flag (BDPCM blocks: contexts of their own, CABACReader::cbf_comp :2066)
            cab.bin(1 if cr else 0, "QtCbf", 2 if bdc else (1 if cb else 0), sub=2)      # tu_cr_coded_flag
        yy = rng.random() < c.p_cbf
        if self.tree == "chroma":
            yy = False                                                         # (no luma in the chroma tree)
        elif isp is not None:
            i, last, prev, any_cbf = isp
            if last and not any_cbf:
                yy = True                                                      # (inferred)
            else:
                cab.bin(1 if yy else 0, "QtCbf", 2 + (1 if prev else 0), sub=0)
        elif (sbt or (not intra and depth0)) and not (cb or cr):
            yy = True                                                          # (inferred: the CU has a residual and chroma has none)
        else:
            cab.bin(1 if yy else 0, "QtCbf", 1 if self.cu.get("bdpcm") else 0, sub=0)      # tu_y_coded_flag
        if c.dqp and not self.dqp_coded and self.tree != "chroma" and (self.cu["w"] > 64 or self.cu["h"] > 64 or yy or cb or cr):
            # cu_qp_delta_abs / cu_qp_delta_sign_flag (CABACReader::cu_qp_delta :2293): once per quantisation group - here a CTU
            dq = rng.choice([0, 0, 1, -1, 2, -2, 3, -4, 5, -6, 7])
            a = abs(dq)
            for k in range(min(a, 5)):
                cab.bin(1, "DeltaQP", 0 if k == 0 else 1)
            if a < 5:
                cab.bin(0, "DeltaQP", 0 if a == 0 else 1)
            else:
                v, k = a - 5, 0                                                # exp_golomb_eqprob, order 0
                while v >= (1 << k):
                    cab.ep(1)
                    v -= 1 << k
                    k += 1
                cab.ep(0)
                if k:
                    cab.eps(v, k)
            if a:
                cab.ep(1 if dq < 0 else 0)
            self.dqp_coded = True
        joint = False
        if chroma and self.c.jccr and ((intra and (cb or cr)) or (cb and cr)):
            joint = rng.random() < 0.4
            cab.bin(1 if joint else 0, "JointCbCrFlag", 2 * cb + cr - 1)       # tu_joint_cbcr_residual_flag
        if cw is None:
            cw, ch = w >> 1, h >> 1
        if yy:
            self.residual(w, h, 0)
        if cb:
            self.residual(cw, ch, 1)
        if cr and not (joint and cb):
            self.residual(cw, ch, 1)
        return yy

    # -- residual_coding: coefficients of the first 4x4 coefficient group only, levels 1..3, at most three of them (well inside the budget of
    # context-coded bins, CoeffCodingContext::m_regBinLimit)
    def residual(self, w, h, ch):
        c, cu = self.c, self.cu
        if c.ts:
            # transform_skip_flag (CABACReader::ts_flag :2493): inferred for BDPCM blocks; then the transform-skip residual coding unless the slice switches it off
            bd = cu.get("bdpcm_c") if ch else cu.get("bdpcm")
            ts = bool(bd)
            if not bd and not (cu.get("isp") and ch == 0) and w <= 32 and h <= 32 and not cu.get("sbt"):
                ts = self.rng.random() < 0.3
                self.cab.bin(1 if ts else 0, "MTSIndex", 4 if ch == 0 else 5)
            if ts:
                cu["ts_any"] = True
                if ch == 0 and not cu.get("luma_tus"):
                    cu["ts_first"] = True
                if ch == 0:
                    cu["luma_tus"] = cu.get("luma_tus", 0) + 1
                if not c.ts_regular:
                    return self.residual_ts(w, h, ch, bool(bd))
                saved = {k: cu.get(k) for k in ("viol", "lfnst_last", "mts_last", "mts_viol")}      # (regular residual coding of a transform-skip block: nothing of it counts for lfnst_idx / mts_idx)
                r = self.residual_full(w, h, ch) if c.big_resi else self.residual_small(w, h, ch)
                cu.update(saved)
                return r
            if ch == 0:
                cu["luma_tus"] = cu.get("luma_tus", 0) + 1
        if self.c.big_resi:
            return self.residual_full(w, h, ch)
        return self.residual_small(w, h, ch)

    def residual_small(self, w, h, ch):
        cab, rng = self.cab, self.rng
        self.stats["cbf"] += 1
        last = rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 7, 9, 12, 15])
        levels = {last: rng.choice([1, 1, 1, 2, 3])}
        for p in rng.sample(range(last), min(last, rng.randrange(0, 3))):
            levels[p] = rng.choice([1, 1, 2, 3])
        self.stats["coefs"] += len(levels)
        log2w, log2h = w.bit_length() - 1, h.bit_length() - 1
        cu = getattr(self, "cu", None)
        if cu is not None:                                                     # what lfnst_idx / mts_idx depend on (CABACReader::residual_coding :2385-2399)
            cu["viol"] = cu["viol"] or last > (7 if ((w == 4 and h == 4) or (w == 8 and h == 8)) else 15)
            cu["lfnst_last"] = cu["lfnst_last"] or last >= 1
            if ch == 0:
                cu["mts_last"] = cu["mts_last"] or last >= 1
        lx, ly = SCAN4[last]
        # last_sig_coeff_x_prefix / y_prefix (positions 0..3: no suffix)
        for (pos, log2s, size, name) in ((lx, log2w, w, "LastX"), (ly, log2h, h, "LastY")):
            off = PREFIX_CTX[log2s] if ch == 0 else 0
            shift = ((log2s + 1) >> 2) if ch == 0 else min(2, max(0, size >> 3))
            group_idx_max = [0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9][min(32, size) - 1]
            for k in range(pos):
                cab.bin(1, name, off + (k >> shift), sub=ch)
            if pos < group_idx_max:
                cab.bin(0, name, off + (pos >> shift), sub=ch)
        # levels, from the last position down to 0
        tpl = {}                                                               # template sums per block position: (sum of first-pass levels, count)
        absl = {}
        first = True
        tmpl_diag, tmpl_sum1 = -1, -1
        signs = []
        state, trans = 0, (32040 if self.dq_on() else 0)                   # dependent quantisation: the quantiser state picks the context set of sig_coeff_flag
        for sp in range(last, -1, -1):
            x, y = SCAN4[sp]
            lv = levels.get(sp, 0)
            if not first:
                s, n = tpl.get((x, y), (0, 0))
                diag = x + y
                ofs = min((s + 1) >> 1, 3) + (4 if diag < 2 else 0)
                if ch == 0:
                    ofs += 4 if diag < 5 else 0
                tmpl_diag, tmpl_sum1 = diag, s - n
                cab.bin(1 if lv else 0, "SigFlag", ofs, sub=ch + 2 * max(0, state - 1))      # sig_coeff_flag')
            if lv:
                off = 0
                if tmpl_diag != -1:
                    off = min(tmpl_sum1, 4) + 1
                    off += (15 if ch == 0 else 5) if tmpl_diag == 0 else ((10 if tmpl_diag < 3 else 5 if tmpl_diag < 10 else 0) if ch == 0 else 0)
                gt1 = lv >= 2
                cab.bin(1 if gt1 else 0, "GtxFlag", off, sub=ch + 2)           # abs_level_gtx_flag[0]
                if gt1:
                    cab.bin((lv - 2) & 1, "ParFlag", off, sub=ch)              # par_level_flag
                    cab.bin(0, "GtxFlag", off, sub=ch)                         # abs_level_gtx_flag[1]: levels up to 3
                signs.append(rng.randrange(0, 2))
                # absVal1stPass: the positions whose template holds this one
                for (dx, dy) in ((0, 2), (1, 1), (0, 1), (2, 0), (1, 0)):
                    px, py = x - dx, y - dy
                    if px >= 0 and py >= 0:
                        s, n = tpl.get((px, py), (0, 0))
                        tpl[(px, py)] = (s + lv, n + 1)
            state = (trans >> ((state << 2) + ((lv & 1) << 1))) & 3
            first = False
        for s in signs:
            cab.ep(s)                                                          # coeff_sign_flag, in coding order


    # -- residual_ts_coding (CABACReader::residual_codingTS :2863, residual_coding_subblockTS :2890): coefficient groups first to last, three passes per group; the
    # contexts look at the left and upper neighbour's value as the decoder holds it at that moment, so this writer RUNS the decoder's procedure with bins of its choice
    def residual_ts(self, w, h, ch, bdpcm):
        cab, rng = self.cab, self.rng
        self.stats["cbf"] += 1
        wg, hg = w >> 2, h >> 2
        cgs = []
        for d in range(wg + hg - 1):
            for y in range(min(d, hg - 1), -1, -1):
                if d - y < wg:
                    cgs.append((d - y, y))
        coeff, flagged = {}, set()
        bins = (w * h * 7) >> 2
        dense = rng.random() < 0.3
        for g, (cx, cy) in enumerate(cgs):
            if g == len(cgs) - 1 and not flagged:
                sig = True
            else:
                sig = rng.random() < (0.7 if (dense or g == 0) else 0.25)
                cab.bin(1 if sig else 0, "TsSigCoeffGroup", (1 if (cx - 1, cy) in flagged else 0) + (1 if (cx, cy - 1) in flagged else 0))
            if not sig:
                continue
            flagged.add((cx, cy))
            pos = [(cx * 4 + SCAN4[i][0], cy * 4 + SCAN4[i][1]) for i in range(16)]
            nz, last1, last2 = [], -1, -1
            i = 0
            while i < 16 and bins >= 4:                                        # pass 1: sig, sign, gt1, parity
                x, y = pos[i]
                l, a = coeff.get((x - 1, y), 0), coeff.get((x, y - 1), 0)
                if not nz and i == 15:
                    s1 = 1
                else:
                    s1 = 1 if rng.random() < (0.6 if dense else 0.3) else 0
                    cab.bin(s1, "TsSigFlag", (1 if l else 0) + (1 if a else 0))
                    bins -= 1
                if s1:
                    sc = 0 if ((l == 0 and a == 0) or l * a < 0) else (1 if (l >= 0 and a >= 0) else 2)
                    sign = rng.randrange(0, 2)
                    cab.bin(sign, "TsResidualSign", sc + (3 if bdpcm else 0))
                    gt1 = rng.randrange(0, 2)
                    cab.bin(gt1, "TsLrg1Flag", 3 if bdpcm else (1 if l else 0) + (1 if a else 0))
                    bins -= 2
                    par = 0
                    if gt1:
                        par = rng.randrange(0, 2)
                        cab.bin(par, "TsParFlag", 0)
                        bins -= 1
                    coeff[(x, y)] = (-1 if sign else 1) * (1 + par + gt1)
                    nz.append(((x, y), sign))
                last1 = i
                i += 1
            j = 0
            while j < 16 and bins >= 4:                                        # pass 2: up to four greater-than flags
                x, y = pos[j]
                t = abs(coeff.get((x, y), 0))
                cutoff = 2
                for _ in range(4):
                    if t >= cutoff:
                        g2 = 1 if rng.random() < 0.4 else 0
                        cab.bin(g2, "TsGtxFlag", cutoff >> 1)
                        bins -= 1
                        t += g2 << 1
                    cutoff += 2
                coeff[(x, y)] = t
                last2 = j
                j += 1
            for k in range(16):                                                # pass 3: remainders, and whole values for what the budget did not reach
                x, y = pos[k]
                t = abs(coeff.get((x, y), 0))
                cutoff = 10 if k <= last2 else (2 if k <= last1 else 0)
                if t >= cutoff:
                    rem = rng.choice([0, 0, 0, 1, 1, 2, 3, 5, 9])
                    self.rem_abs_ep(rem, 1, 5)
                    t += (rem << 1) if k <= last1 else rem
                    if t and k > last1:
                        sign = rng.randrange(0, 2)
                        cab.ep(sign)
                        nz.append(((x, y), sign))
                if not bdpcm and cutoff and t > 0:
                    pred = max(abs(coeff.get((x - 1, y), 0)), abs(coeff.get((x, y - 1), 0)))
                    t = pred if (t == 1 and pred > 0) else t - (1 if t <= pred else 0)
                coeff[(x, y)] = t
            for (xy, sign) in nz:
                coeff[xy] = -coeff[xy] if sign else coeff[xy]
            self.stats["coefs"] += len(nz)

    # -- residual_coding in full (CABACReader::residual_coding :2361-2460, residual_coding_subblock :2704-2861, CoeffCodingContext): any last position inside the
    # 32x32 zero-out region, coded_sub_block_flag per 4x4 coefficient group, the context-coded pass with its bin budget, Golomb-Rice remainders with the
    # template-derived parameter, the bypass pass for what the budget no longer covers, levels up to a few dozen, dependent quantisation states
    def residual_full(self, w, h, ch):
        cab, rng = self.cab, self.rng
        self.stats["cbf"] += 1
        wz, hz = min(32, w), min(32, h)
        wg, hg = wz >> 2, hz >> 2
        cgs = []                                                               # coefficient groups in scan order (ScanGenerator, Rom.cpp:130-172)
        for d in range(wg + hg - 1):
            for y in range(min(d, hg - 1), -1, -1):
                if d - y < wg:
                    cgs.append((d - y, y))
        def pos_of(sp):
            cx, cy = cgs[sp >> 4]
            ix, iy = SCAN4[sp & 15]
            return cx * 4 + ix, cy * 4 + iy
        # -- what is coded: the last group (mostly the first few), which groups hold anything, the levels
        # a luma block of an SBT CU in a sequence with MTS keeps coefficients in its first 16 columns / rows only (implicit DST-7 / DCT-8): shorter last-position
        # prefix, coefficient groups beyond are not even flagged (CABACReader::last_sig_coeff :2641, residual_coding :2416-2424)
        cu0 = getattr(self, "cu", None) or {}
        zo = ch == 0 and self.c.mts and bool(cu0.get("sbt")) and w <= 32 and h <= 32
        lim_x, lim_y = (4 if (zo and w == 32) else wg), (4 if (zo and h == 32) else hg)
        ok_cg = [i for i, (cx, cy) in enumerate(cgs) if cx < lim_x and cy < lim_y]
        ncg = len(ok_cg)
        last_cg = ok_cg[0 if rng.random() < 0.45 else min(ncg - 1, int(rng.expovariate(0.5)))]
        last = (last_cg << 4) + rng.choice([0, 0, 1, 2, 3, 5, 8, 11, 15])
        dense = rng.random() < 0.15                                            # (some blocks full enough to run out of context-coded bins)
        def level():
            r = rng.random()
            return 1 if r < 0.5 else 2 if r < 0.7 else 3 if r < 0.8 else rng.randrange(4, 8) if r < 0.93 else rng.randrange(8, 60)
        levels = {last: level()}
        sig_cg = {last_cg: True, 0: True}
        for g in range(last_cg - 1, -1, -1):
            if g not in ok_cg:
                continue
            if g == 0 or rng.random() < (0.8 if dense else 0.4):
                sig_cg[g] = True
                for sp in range((g << 4) + 15, (g << 4) - 1, -1):
                    if rng.random() < (0.7 if dense else 0.2):
                        levels[sp] = level()
        for sp in range(last - 1, (last_cg << 4) - 1, -1):
            if rng.random() < (0.7 if dense else 0.25):
                levels[sp] = level()
        self.stats["coefs"] += len(levels)
        cu = getattr(self, "cu", None)
        if cu is not None:                                                     # what lfnst_idx / mts_idx depend on (CABACReader::residual_coding :2385-2399, :2437-2440)
            cu["viol"] = cu["viol"] or last > (7 if ((w == 4 and h == 4) or (w == 8 and h == 8)) else 15)
            cu["lfnst_last"] = cu["lfnst_last"] or last >= 1
            if ch == 0:
                cu["mts_last"] = cu["mts_last"] or last >= 1
        # -- last_sig_coeff_{x,y}_prefix, then the suffixes (CABACReader::last_sig_coeff :2636-2700)
        log2w, log2h = w.bit_length() - 1, h.bit_length() - 1
        lx, ly = pos_of(last)
        GROUP_IDX = [0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9]
        MIN_IN_GROUP = [0, 1, 2, 3, 4, 6, 8, 12, 16, 24]
        grp = []
        for (pos, log2s, size, name) in ((lx, log2w, w, "LastX"), (ly, log2h, h, "LastY")):
            off = PREFIX_CTX[log2s] if ch == 0 else 0
            shift = ((log2s + 1) >> 2) if ch == 0 else min(2, max(0, size >> 3))
            g, gmax = GROUP_IDX[pos], GROUP_IDX[(16 if (zo and size == 32) else min(32, size)) - 1]
            for k in range(g):
                cab.bin(1, name, off + (k >> shift), sub=ch)
            if g < gmax:
                cab.bin(0, name, off + (g >> shift), sub=ch)
            grp.append((pos, g))
        for pos, g in grp:
            if g > 3:
                cab.eps(pos - MIN_IN_GROUP[g], (g - 2) >> 1)
        # -- the groups from the last one down
        tpl, coeff = {}, {}                                                    # per position: (sum of first-pass values, number) of its template; current absolute values
        tmpl_diag, tmpl_sum1 = -1, -1
        state, trans = 0, (32040 if self.dq_on() else 0)
        area = (16 if (zo and w == 32) else wz) * (16 if (zo and h == 32) else hz)
        rem_bins = (area * 28) >> 4
        flagged = set()
        RICE = [0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3]
        def tsum(x, y, base):
            sm = 0
            if x + 2 < w:
                sm += coeff.get((x + 1, y), 0) + coeff.get((x + 2, y), 0)
                if y + 1 < h:
                    sm += coeff.get((x + 1, y + 1), 0)
            elif x + 1 < w:
                sm += coeff.get((x + 1, y), 0)
                if y + 1 < h:
                    sm += coeff.get((x + 1, y + 1), 0)
            if y + 2 < h:
                sm += coeff.get((x, y + 1), 0) + coeff.get((x, y + 2), 0)
            elif y + 1 < h:
                sm += coeff.get((x, y + 1), 0)
            return max(min(sm - 5 * base, 31), 0)
        for g in range(last_cg, -1, -1):
            if g not in ok_cg:
                continue
            cx, cy = cgs[g]
            min_sub = g << 4
            is_last = g == last_cg
            sig = bool(sig_cg.get(g))
            if not (is_last or g == 0):
                right = (cx + 1, cy) in flagged if cx != wg - 1 else False
                lower = (cx, cy + 1) in flagged if cy != hg - 1 else False
                cab.bin(1 if sig else 0, "SigCoeffGroup", 1 if (right or lower) else 0, sub=ch)      # coded_sub_block_flag
            if not sig:
                continue
            flagged.add((cx, cy))
            if ch == 0 and (cx > 3 or cy > 3) and cu is not None:
                cu["mts_viol"] = True
            nxt = last if is_last else min_sub + 15
            infer = nxt if is_last else (min_sub if g else -1)
            if not is_last and g and not any(levels.get(sp, 0) for sp in range(min_sub + 1, min_sub + 16)):
                levels.setdefault(min_sub, level())                            # (a flagged group holds something: its first position when nothing else)
                if not levels[min_sub]:
                    levels[min_sub] = 1
            num_nz, gt2pos, nsigns = 0, [], 0
            while nxt >= min_sub and rem_bins >= 4:
                x, y = pos_of(nxt)
                lv = levels.get(nxt, 0)
                inferred = (num_nz == 0 and nxt == infer)
                if inferred:
                    assert lv
                else:
                    sm, n = tpl.get((x, y), (0, 0))
                    diag = x + y
                    ofs = min((sm + 1) >> 1, 3) + (4 if diag < 2 else 0)
                    if ch == 0:
                        ofs += 4 if diag < 5 else 0
                    tmpl_diag, tmpl_sum1 = diag, sm - n
                    cab.bin(1 if lv else 0, "SigFlag", ofs, sub=ch + 2 * max(0, state - 1))      # sig_coeff_flag
                    rem_bins -= 1
                if lv:
                    off = 0
                    if tmpl_diag != -1:
                        off = min(tmpl_sum1, 4) + 1
                        off += (15 if ch == 0 else 5) if tmpl_diag == 0 else ((10 if tmpl_diag < 3 else 5 if tmpl_diag < 10 else 0) if ch == 0 else 0)
                    num_nz += 1
                    gt1 = lv >= 2
                    cab.bin(1 if gt1 else 0, "GtxFlag", off, sub=ch + 2)       # abs_level_gtx_flag[0]
                    rem_bins -= 1
                    v1 = 1
                    if gt1:
                        par, gt2 = lv & 1, lv >= 4
                        cab.bin(par, "ParFlag", off, sub=ch)                   # par_level_flag
                        cab.bin(1 if gt2 else 0, "GtxFlag", off, sub=ch)       # abs_level_gtx_flag[1]
                        rem_bins -= 2
                        v1 = 2 + par + (2 if gt2 else 0)
                        if gt2:
                            gt2pos.append((x, y, (lv - v1) >> 1))
                    coeff[(x, y)] = v1
                    for (dx, dy) in ((0, 2), (1, 1), (0, 1), (2, 0), (1, 0)):  # absVal1stPass: the positions whose template holds this one
                        px, py = x - dx, y - dy
                        if px >= 0 and py >= 0:
                            sm, n = tpl.get((px, py), (0, 0))
                            tpl[(px, py)] = (sm + v1, n + 1)
                state = (trans >> ((state << 2) + ((lv & 1) << 1))) & 3
                nxt -= 1
            for (x, y, rem) in gt2pos:                                         # abs_remainder
                self.rem_abs_ep(rem, RICE[tsum(x, y, 4)], 5)
                coeff[(x, y)] += rem << 1
            while nxt >= min_sub:                                              # dec_abs_level: what the budget of context-coded bins no longer covers
                x, y = pos_of(nxt)
                lv = levels.get(nxt, 0)
                rice = RICE[tsum(x, y, 0)]
                pos0 = (1 if state < 2 else 2) << rice
                self.rem_abs_ep(pos0 if lv == 0 else (lv - 1 if lv <= pos0 else lv), rice, 5)
                state = (trans >> ((state << 2) + ((lv & 1) << 1))) & 3
                if lv:
                    coeff[(x, y)] = lv
                    num_nz += 1
                nxt -= 1
            for _ in range(num_nz):
                cab.ep(rng.randrange(0, 2))                                    # coeff_sign_flag


NAL_TRAIL = 0


def gop_plan(num_pictures, inter):
    """pictures in decoding order.  Intra streams: IDR pictures only.  Inter streams: an IDR picture, then groups of four - a key picture (P, or B with
    both lists in the past), a B picture half way between two decoded pictures (reference pictures at equal distance on either side: what DMVR and
    BDOF ask for), its two B neighbours"""
    if not inter:
        return [dict(poc=0, type="I", idr=True, l0=[], l1=[]) for _ in range(num_pictures)]
    pics = [dict(poc=0, type="I", idr=True, l0=[], l1=[])]
    base = 0
    while len(pics) < num_pictures:
        k = base + 4
        prev = [base] + ([base - 4] if base >= 4 else [])
        pics.append(dict(poc=k, type="P" if (k // 4) % 2 else "B", idr=False, l0=prev, l1=(prev if (k // 4) % 2 == 0 else [])))
        pics.append(dict(poc=base + 2, type="B", idr=False, l0=[base, k], l1=[k, base]))
        pics.append(dict(poc=base + 1, type="B", idr=False, l0=[base, base + 2], l1=[base + 2, k], col_l0=0))
        pics.append(dict(poc=base + 3, type="B", idr=False, l0=[base + 2, base], l1=[k]))
        base = k
    return pics[:num_pictures]


def write_hash_sei(md5s):
    """sei_rbsp with one decoded picture hash message (payload type 132): MD5 of every colour component of the decoded picture"""
    b = Bits()
    b.u(8, 132)                                      # payload_type
    b.u(8, 2 + 16 * len(md5s))                       # payload_size
    b.u(8, 0)                                        # dph_sei_hash_type: MD5
    b.flag(len(md5s) == 1)                           # dph_sei_single_component_flag
    b.u(7, 0)                                        # dph_sei_reserved_zero_7bits
    for m in md5s:
        for byte in m:
            b.u(8, byte)                             # dph_sei_picture_md5
    b.trailing()
    return b.bytes()


def write_stream(c, num_pictures, seed, tables, renorm, hashes=None):
    """hashes: [per picture in decoding order: MD5 digest per component] -> a decoded-picture-hash SEI behind every picture (what the decoder checks with verifyPictureHash / -dph)"""
    rng = random.Random(seed)
    out = bytearray()
    out += nal(NAL_SPS, write_sps(c), long_start=True)
    import copy
    cfgs = []
    for k, (ww, hh, win) in enumerate(c.sizes):
        ck = copy.copy(c)
        ck.width, ck.height = ww, hh
        ck.partition = partition_of(ck)
        cfgs.append(ck)
        out += nal(NAL_PPS, write_pps(ck, k, win), long_start=True)
    size_of_poc = {}
    if c.lmcs:
        out += nal(NAL_PREFIX_APS, write_lmcs_aps(c, rng, 0), long_start=True)
    if c.scaling:
        for i in range(2):
            out += nal(NAL_PREFIX_APS, write_scaling_aps(c, rng, i), long_start=True)
    alf_aps = []
    if c.alf:
        for i in range(c.alf_aps):
            data, nalt, ncc = write_alf_aps(c, rng, i)
            out += nal(NAL_PREFIX_APS, data, long_start=True)
            alf_aps.append((nalt, ncc))
    stats = []
    c_seq = c
    for pic_idx, pic in enumerate(gop_plan(num_pictures, c.inter)):
        # the picture's size (reference picture resampling: the sizes in turn); temporal MV prediction only from a picture of the same size and window
        pic["pps"] = (pic_idx // 2) % len(c_seq.sizes) if len(c_seq.sizes) > 1 else 0
        c = cfgs[pic["pps"]]
        size_of_poc[pic["poc"]] = pic["pps"]
        if len(c_seq.sizes) > 1 and pic["type"] != "I":
            same0 = [i for i, r in enumerate(pic["l0"]) if size_of_poc[r] == pic["pps"]]
            same1 = [i for i, r in enumerate(pic["l1"]) if size_of_poc[r] == pic["pps"]]
            pic["tmvp"] = 1 if (same0 or (same1 and pic["type"] == "B")) else 0
            if pic["tmvp"]:
                want_l0 = pic.get("col_l0", 1)
                if (want_l0 and same0) or not same1 or pic["type"] != "B":
                    pic["col_l0"], pic["col_idx"] = 1, same0[0] if same0 else 0
                    if not same0:
                        pic["tmvp"] = 0
                else:
                    pic["col_l0"], pic["col_idx"] = 0, same1[0]
            if c.sbtmvp and not c.affine:
                c = copy.copy(c)
                c.max_aff_merge = 1 if pic["tmvp"] else 0
        if c.scaling:
            pic["scaling"], pic["scaling_aps"] = (1 if rng.random() < 0.85 else 0), rng.randrange(0, 2)
        if pic["type"] != "I":
            cur, l0, l1 = pic["poc"], pic["l0"], pic["l1"]
            pic["ldc"] = all(r < cur for r in l0 + l1)                         # Slice::getCheckLDC: no reference picture follows in output order
            # symmetric MVD: the nearest picture before in one list and after in the other (DecLibParser.cpp:850-925)
            pic["bidir"] = (not pic["ldc"]) and ((any(r < cur for r in l0) and any(r > cur for r in l1)) or (any(r > cur for r in l0) and any(r < cur for r in l1)))
            if c.wp:
                def entry():
                    return dict(luma=(rng.randrange(-20, 21), rng.randrange(-10, 11)) if rng.random() < 0.6 else None,
                                chroma=[(rng.randrange(-20, 21), rng.randrange(-20, 21)) for _ in range(2)] if rng.random() < 0.5 else None)
                pic["wp"] = dict(denom=rng.randrange(2, 7), dchroma=rng.randrange(-1, 2), l0=[entry() for _ in l0], l1=[entry() for _ in l1])
        def alf_choice():
            # the slice's ALF choice: which APSs the luma filter sets come from, the APS of the chroma filters (its alternatives), the APSs of the CC-ALF filters
            ids = list(range(c.alf_aps))
            rng.shuffle(ids)
            ch = rng.randrange(0, c.alf_aps)
            cc = [rng.randrange(0, c.alf_aps) if (c.ccalf and rng.random() < 0.8) else None for _ in range(2)]
            return dict(on=rng.random() < 0.9, luma_aps=ids[:rng.randrange(0, c.alf_aps + 1)], cb=rng.random() < 0.8, cr=rng.random() < 0.8, chroma_aps=ch,
                        nalt=alf_aps[ch][0], cc_cb=cc[0], cc_cr=cc[1], ncc=[alf_aps[cc[0]][1][0] if cc[0] is not None else 0, alf_aps[cc[1]][1][1] if cc[1] is not None else 0])
        if c.alf:
            pic["alf"] = alf_choice()
        if c.partition:
            # a picture header NAL unit, then the slices: each with a type, a QP, SAO / ALF / dependent quantisation / LMCS switches of its own
            b = Bits()
            write_slice_header(c, b, pic, "ph")
            out += nal(NAL_PH, b.bytes(), long_start=True)
            pw = PictureWriter(c, None, rng, pic)
            nsl = len(c.partition["slices"])
            for k, ctus in enumerate(c.partition["slices"]):
                st = pic["type"]
                if st != "I" and rng.random() < 0.3:
                    st = rng.choice(["I", "P"])
                sl = dict(idx=k, n=nsl, type=st, qp=max(12, min(45, c.qp + rng.randrange(-6, 7))), sao=(rng.random() < 0.8, rng.random() < 0.8), dq=rng.random() < 0.7,
                          lmcs=1 if rng.random() < 0.75 else 0, scaling=1 if rng.random() < 0.75 else 0)
                if c.alf:
                    sl["alf"] = alf_choice()
                b = Bits()
                write_slice_header(c, b, pic, sl)
                cab = Cabac(tables, renorm, {"B": 0, "P": 1, "I": 2}[st], sl["qp"])
                pw.picture(ctus, cab, sl)
                b.b += cab.finish()
                b.trailing()
                out += nal(NAL_IDR_N_LP if pic["idr"] else NAL_TRAIL, b.bytes(), long_start=(k == 0))
        else:
            b = Bits()
            write_slice_header(c, b, pic)
            cab = Cabac(tables, renorm, {"B": 0, "P": 1, "I": 2}[pic["type"]], c.qp)
            pw = PictureWriter(c, cab, rng, pic)
            pw.picture()
            b.b += cab.finish()
            b.trailing()
            out += nal(NAL_IDR_N_LP if pic["idr"] else NAL_TRAIL, b.bytes(), long_start=True)
        if hashes is not None:
            out += nal(NAL_SUFFIX_SEI, write_hash_sei(hashes[pic_idx]))
        stats.append(pw.stats)
        c = c_seq
    return bytes(out), stats


def reference_md5(path, frames_expected=None, keep=False, extra=()):
    """decode with the reference's own decoder and application -> MD5 over the output frames (16-bit little endian for more than 8 bits, planar)"""
    yuv = path + ".ref.yuv"
    r = subprocess.run([APP_REF, "-b", path, "-o", yuv, "-t", "2", "-v", "3"] + list(extra), capture_output=True, text=True, timeout=300)
    if r.returncode != 0 or not os.path.exists(yuv):
        raise RuntimeError("the reference decoder refused %s:\n%s" % (path, (r.stdout + r.stderr)[-2000:]))
    data = open(yuv, "rb").read()
    os.remove(yuv)
    if keep:
        return hashlib.md5(data).hexdigest(), data, r.stdout + r.stderr
    return hashlib.md5(data).hexdigest(), len(data), r.stdout + r.stderr


def picture_hashes(c, num_pictures, yuv):
    """[per picture in decoding order: MD5 digests of Y, Cb, Cr] from the reference decoder's output file (pictures in output order = ascending POC; samples of more than 8 bits as two bytes,
    little endian - the byte order the hash of the standard is defined over)"""
    bps = 2 if c.bit_depth > 8 else 1
    plan = gop_plan(num_pictures, c.inter)
    dims = [c.sizes[(i // 2) % len(c.sizes) if len(c.sizes) > 1 else 0][:2] for i in range(len(plan))]
    # (an intra stream is a sequence of IDR pictures, all of POC 0, put out in decoding order; an inter stream is one coded video sequence put out by POC)
    order = sorted(range(len(plan)), key=lambda i: (plan[i]["poc"], i)) if c.inter else list(range(len(plan)))
    out, off = [None] * len(plan), 0
    for i in order:
        planes = []
        ysz, csz = dims[i][0] * dims[i][1] * bps, (dims[i][0] // 2) * (dims[i][1] // 2) * bps
        if c.mono:                                                             # (the application writes the luma plane of a 4:0:0 picture only)
            planes.append(hashlib.md5(yuv[off:off + ysz]).digest())
            off += ysz
            out[i] = planes
            continue
        for sz in (ysz, csz, csz):
            planes.append(hashlib.md5(yuv[off:off + sz]).digest())
            off += sz
        out[i] = planes
    assert off == len(yuv)
    return out


FIXTURES = [
    # name, config, pictures, seed
    ("mini_ctu32_64x64", dict(width=64, height=64, log2_ctu=5, qp=30), 2, 1),
    ("mini_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=32), 3, 2),
    ("mini_ctu128_256x256", dict(width=256, height=256, log2_ctu=7, qp=27, p_split=0.7), 2, 3),
    ("mini_ctu64_tb32_192x128_8bit", dict(width=192, height=128, log2_ctu=6, qp=35, bit_depth=8, max_tb64=False, p_cbf=0.8, p_cbf_chroma=0.6), 2, 4),
    ("mini_ctu128_nodeblock_384x256", dict(width=384, height=256, log2_ctu=7, qp=24, deblock=False, p_split=0.8), 2, 5),
    # inter pictures: skip / merge / AMVP CUs, temporal MV prediction, BDOF and DMVR (decoder-side tools: no syntax of their own)
    ("mini_inter_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, inter=True), 9, 11),
    ("mini_inter_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, qp=28, inter=True, p_split=0.7), 9, 12),
    ("mini_inter_tools_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, qp=32, inter=True, sbtmvp=True, mmvd=True, affine=True, ciip=True, gpm=True, p_split=0.65), 9, 13),
    ("mini_inter_tools_ctu64_8bit_320x192", dict(width=320, height=192, log2_ctu=6, qp=34, bit_depth=8, inter=True, sbtmvp=True, mmvd=True, affine=True, ciip=True, gpm=True, p_skip=0.2), 13, 14),
    # SAO, LMCS (luma mapping + chroma residual scaling), joint Cb-Cr residuals, dependent quantisation - intra pictures, then with every inter tool
    ("mini_filters_ctu64_256x192", dict(width=256, height=192, log2_ctu=6, qp=30, sao=True, lmcs=True, jccr=True, dep_quant=True, p_cbf=0.7, p_cbf_chroma=0.5), 3, 21),
    ("mini_filters_inter_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, qp=29, inter=True, sbtmvp=True, mmvd=True, affine=True, ciip=True, gpm=True,
                                               sao=True, lmcs=True, jccr=True, dep_quant=True, p_cbf=0.6, p_cbf_chroma=0.4), 9, 22),
    # binary and ternary splits (rectangular CUs from 8x16 to 64x32: wide-angle intra modes, rectangular transforms, GPM / CIIP / affine size rules)
    ("mini_mtt_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, log2_min_qt=4, qp=30, mtt_depth=2), 2, 31),
    ("mini_mtt_inter_tools_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, log2_min_qt=5, qp=30, mtt_depth=3, inter=True, sbtmvp=True, mmvd=True, affine=True, ciip=True,
                                                 gpm=True, sao=True, lmcs=True, jccr=True, dep_quant=True, p_split=0.7), 9, 34),
    # the intra tools with syntax of their own: multiple reference lines, intra sub-partitions, matrix-based prediction, cross-component linear model; LFNST and explicit MTS
    ("mini_intra_tools_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, mrl=True, isp=True, mip=True, cclm=True, lfnst=True, mts=True, p_cbf=0.7), 2, 41),
    ("mini_intra_tools_mtt_ctu128_256x256", dict(width=256, height=256, log2_ctu=7, log2_min_qt=4, qp=28, mtt_depth=2, mrl=True, mip=True, cclm=True, isp=True, lfnst=True, mts=True,
                                                 sao=True, lmcs=True, jccr=True, dep_quant=True), 3, 47),
    # ALF and CC-ALF: several APSs (luma filter sets with class maps, chroma alternatives, cross-component filters, clipping), per-slice choice, per-CTU controls
    ("mini_alf_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, alf=True), 3, 51),
    ("mini_alf_ccalf_sao_ctu128_256x256", dict(width=256, height=256, log2_ctu=7, qp=28, alf=True, ccalf=True, sao=True), 3, 52),
    # everything at once, random access
    ("mini_all_tools_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, log2_min_qt=4, qp=30, mtt_depth=2, inter=True, sbtmvp=True, mmvd=True, affine=True, ciip=True, gpm=True,
                                           mrl=True, mip=True, cclm=True, isp=True, lfnst=True, mts=True, sao=True, lmcs=True, jccr=True, dep_quant=True, alf=True, ccalf=True,
                                           alf_aps=3, p_intra=0.25), 9, 53),
    # residual coding in full: last positions anywhere in the 32x32 region, many coefficient groups, levels up to 59 with Golomb-Rice remainders, blocks dense
    # enough to exhaust the context-coded bins (bypass pass); what the extractor's packed coefficient corners and the transforms see from a real parser
    ("mini_resi_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, big_resi=True, p_cbf=0.8, p_cbf_chroma=0.6), 2, 61),
    ("mini_resi_tb32_8bit_ctu64_192x128", dict(width=192, height=128, log2_ctu=6, qp=36, bit_depth=8, max_tb64=False, big_resi=True, p_cbf=0.8, p_cbf_chroma=0.6, lfnst=True, mts=True), 2, 64),
    ("mini_resi_dq_jccr_ctu128_256x256", dict(width=256, height=256, log2_ctu=7, qp=34, big_resi=True, dep_quant=True, jccr=True, p_cbf=0.8, p_cbf_chroma=0.6, p_split=0.5), 2, 62),
    ("mini_resi_all_tools_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, log2_min_qt=4, qp=32, mtt_depth=2, inter=True, sbtmvp=True, mmvd=True, affine=True, ciip=True,
                                                gpm=True, mrl=True, mip=True, cclm=True, isp=True, lfnst=True, mts=True, sao=True, lmcs=True, jccr=True, dep_quant=True, alf=True,
                                                ccalf=True, big_resi=True, p_intra=0.25), 9, 63),
    # more inter tools with syntax of their own: adaptive MV resolution (also affine), BCW weights, symmetric MVD, sub-block transforms (with the implicit
    # DST-7 / DCT-8 zero-out when MTS is on)
    ("mini_amvr_bcw_smvd_sbt_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, log2_min_qt=4, qp=31, mtt_depth=2, inter=True, sbtmvp=True, mmvd=True, affine=True, ciip=True,
                                                   gpm=True, amvr=True, bcw=True, smvd=True, sbt=True, mts=True, big_resi=True, p_intra=0.1, p_skip=0.15, p_merge=0.35), 13, 75),
    # QP that changes from CTU to CTU (cu_qp_delta), chroma QP offsets in the PPS and the slice header, deblocking offsets per component, luma-adaptive deblocking
    ("mini_dqp_chroma_qp_ladf_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, log2_min_qt=4, qp=30, mtt_depth=2, inter=True, sbtmvp=True, affine=True, dqp=True,
                                                    chroma_qp=True, db_offsets=True, ladf=True, jccr=True, sao=True, big_resi=True, p_intra=0.2), 9, 76),
    # horizontal reference wrap-around with motion vectors that leave the picture by up to 75 samples
    ("mini_wraparound_ctu64_384x256", dict(width=384, height=256, log2_ctu=6, log2_min_qt=4, qp=30, mtt_depth=1, inter=True, sbtmvp=True, affine=True, mmvd=True, gpm=True, ciip=True,
                                           wrap=True, max_mvd=300, p_intra=0.1), 9, 79),
    # virtual boundaries of the in-loop filters (sequence level): deblocking, SAO, ALF and CC-ALF stop there (the back-end's unfused filter passes)
    ("mini_virtual_boundaries_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, log2_min_qt=4, qp=30, mtt_depth=2, inter=True, vb=True, sao=True, alf=True, ccalf=True,
                                                    big_resi=True, p_intra=0.2), 5, 80),
    # explicit weighted prediction: weights and offsets per reference entry, uni- and bi-prediction
    ("mini_weighted_pred_ctu64_8bit_320x192", dict(width=320, height=192, log2_ctu=6, log2_min_qt=4, qp=32, bit_depth=8, mtt_depth=2, inter=True, sbtmvp=True, affine=True, mmvd=True,
                                                   gpm=True, ciip=True, wp=True, p_intra=0.1), 9, 81),
    # dual tree: I slices with a luma tree and a chroma tree per 64x64 (blocks above quartered without syntax), chroma CUs with modes, transform units and LFNST of
    # their own; single-tree inter pictures behind a dual-tree IRAP
    ("mini_dual_tree_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, dual_tree=True), 2, 91),
    ("mini_dual_tree_mtt_ctu128_256x256", dict(width=256, height=256, log2_ctu=7, log2_min_qt=4, qp=28, mtt_depth=2, dual_tree=True, big_resi=True, jccr=True), 2, 92),
    ("mini_dual_tree_ctu32_128x64", dict(width=128, height=64, log2_ctu=5, qp=30, dual_tree=True, log2_min_qt_c=3, big_resi=True), 2, 93),
    ("mini_dual_tree_tools_ctu128_256x256", dict(width=256, height=256, log2_ctu=7, log2_min_qt=4, qp=30, mtt_depth=2, dual_tree=True, mrl=True, isp=True, mip=True, lfnst=True, mts=True,
                                                 big_resi=True, jccr=True, dep_quant=True, sao=True, lmcs=True, alf=True, ccalf=True, dqp=True), 3, 94),
    ("mini_dual_tree_inter_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, log2_min_qt=4, qp=30, mtt_depth=2, dual_tree=True, inter=True, sbtmvp=True, affine=True, mip=True,
                                                 lfnst=True, big_resi=True, lmcs=True), 5, 95),
    # explicit scaling lists: two APSs of 28 matrices (flat, copied, predicted, coded; DC entries; the 64x64 zero-out), chosen per picture; with and without LFNST blocks exempt
    ("mini_scaling_lists_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, scaling=True, big_resi=True), 3, 101),
    ("mini_scaling_lists_tools_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, log2_min_qt=4, qp=30, mtt_depth=2, inter=True, sbtmvp=True, affine=True, scaling=True, lfnst=True,
                                                     mts=True, isp=True, mip=True, jccr=True, dep_quant=True, big_resi=True, sbt=True, p_intra=0.25), 9, 102),
    ("mini_scaling_lists_dual_tree_ctu128_256x256", dict(width=256, height=256, log2_ctu=7, log2_min_qt=4, qp=27, mtt_depth=2, dual_tree=True, scaling=True, lfnst=True, big_resi=True), 2, 103),
    # transform skip and BDPCM (luma and chroma): the transform-skip residual coding with its neighbour-dependent contexts and level mapping; the regular residual coding
    # for transform-skip blocks (sh_ts_residual_coding_disabled_flag); with dependent quantisation; in a dual tree with scaling lists
    ("mini_ts_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, ts=True, big_resi=True, p_cbf=0.8, p_cbf_chroma=0.6), 2, 111),
    ("mini_ts_bdpcm_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, ts=True, bdpcm=True, big_resi=True, p_cbf=0.8, p_cbf_chroma=0.6), 2, 112),
    ("mini_ts_regular_resi_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, ts=True, bdpcm=True, ts_regular=True, big_resi=True, p_cbf=0.8, p_cbf_chroma=0.6), 2, 113),
    ("mini_ts_bdpcm_tools_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, log2_min_qt=4, qp=30, mtt_depth=2, inter=True, sbtmvp=True, affine=True, ts=True, bdpcm=True, lfnst=True,
                                                mts=True, isp=True, mip=True, mrl=True, jccr=True, big_resi=True, sbt=True, sao=True, lmcs=True, dqp=True, p_intra=0.3), 9, 114),
    ("mini_ts_bdpcm_dual_tree_scaling_ctu128_256x256", dict(width=256, height=256, log2_ctu=7, log2_min_qt=4, qp=28, mtt_depth=2, dual_tree=True, ts=True, bdpcm=True, lfnst=True,
                                                            big_resi=True, scaling=True), 2, 115),
    ("mini_ts_bdpcm_dep_quant_8bit_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, bit_depth=8, ts=True, bdpcm=True, dep_quant=True, big_resi=True), 2, 116),
    # several slices per picture (a picture header NAL unit, then slices with a type, QP, SAO / ALF / LMCS / dependent-quantisation switches of their own: I and P slices
    # inside B pictures): bands of CTU rows inside one tile; a grid of tiles with one slice each, with and without in-loop filtering across the boundaries
    ("mini_slices_rows_ctu64_256x256", dict(width=256, height=256, log2_ctu=6, qp=30, part=("rows", [1, 2]), sao=True, big_resi=True), 2, 121),
    ("mini_slices_tiles_ctu64_384x256", dict(width=384, height=256, log2_ctu=6, qp=30, part=("tiles", 3, 2), sao=True, alf=True, ccalf=True, big_resi=True), 2, 122),
    ("mini_slices_rows_inter_ctu64_384x256", dict(width=384, height=256, log2_ctu=6, log2_min_qt=4, qp=30, mtt_depth=2, part=("rows", [1]), inter=True, sbtmvp=True, affine=True, mmvd=True,
                                                  gpm=True, ciip=True, sao=True, alf=True, ccalf=True, lmcs=True, dep_quant=True, jccr=True, big_resi=True, dqp=True, p_intra=0.2), 9, 123),
    ("mini_tiles_no_filter_across_ctu64_384x256", dict(width=384, height=256, log2_ctu=6, log2_min_qt=4, qp=31, mtt_depth=2, part=("tiles", 2, 2), lf_across=False, inter=True, sbtmvp=True,
                                                       affine=True, sao=True, alf=True, ccalf=True, lmcs=True, scaling=True, dual_tree=True, big_resi=True, mip=True, isp=True, mrl=True,
                                                       p_intra=0.25), 9, 124),
    # reference picture resampling: several PPSs of different picture size (and scaling windows with offsets) in one coded video sequence, the size changing every
    # two pictures - prediction from pictures 2x, 1.5x, 1.2x, 0.5x .. the size, per direction; temporal MV prediction only from a picture of the same size
    ("mini_rpr_half_ctu64_384x256", dict(width=384, height=256, log2_ctu=6, log2_min_qt=4, qp=30, mtt_depth=1, inter=True, rpr=[(192, 128, None)], p_intra=0.1), 9, 131),
    ("mini_rpr_four_sizes_ctu64_384x256", dict(width=384, height=256, log2_ctu=6, log2_min_qt=4, qp=30, mtt_depth=2, inter=True, sbtmvp=True, affine=True, mmvd=True, gpm=True, ciip=True,
                                               rpr=[(256, 192, (2, 2, 0, 4)), (320, 256, (-4, 0, 2, 2)), (192, 128, None)], sao=True, alf=True, lmcs=True, big_resi=True, p_intra=0.1), 13, 132),
    ("mini_rpr_8bit_ctu128_256x256", dict(width=256, height=256, log2_ctu=7, log2_min_qt=4, qp=32, bit_depth=8, mtt_depth=1, inter=True, sbtmvp=True, affine=True, amvr=True, bcw=True,
                                          rpr=[(128, 128, None)], dep_quant=True, p_intra=0.1), 9, 133),
    # 4:0:0: no chroma anywhere - parameter sets, APSs without chroma parts, slice data without chroma modes and coded flags, a single-component picture hash
    ("mini_400_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, mono=True, big_resi=True, sao=True, alf=True, lmcs=True), 3, 141),
    ("mini_400_tools_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, log2_min_qt=4, qp=30, mtt_depth=2, mono=True, inter=True, sbtmvp=True, affine=True, mmvd=True, gpm=True,
                                           ciip=True, mrl=True, isp=True, mip=True, lfnst=True, mts=True, ts=True, bdpcm=True, sbt=True, sao=True, alf=True, lmcs=True, scaling=True,
                                           dqp=True, big_resi=True, p_intra=0.25), 9, 142),
    # sub-pictures (one slice each): bands of CTU rows / tiles; treated as pictures or not (motion vectors clipped and references clamped at the sub-picture's
    # rectangle), in-loop filtering across their boundaries on or off - per sub-picture; motion vectors up to 50 samples
    ("mini_subpics_rows_ctu64_384x256", dict(width=384, height=256, log2_ctu=6, log2_min_qt=4, qp=30, mtt_depth=1, part=("rows", [2]), subpic=[(1, 0), (1, 1)], inter=True, sbtmvp=True,
                                             affine=True, mmvd=True, sao=True, alf=True, max_mvd=200, p_intra=0.1), 9, 151),
    ("mini_subpics_tiles_ctu64_384x256", dict(width=384, height=256, log2_ctu=6, log2_min_qt=4, qp=30, mtt_depth=2, part=("tiles", 3, 2), subpic=[(1, 0), (0, 1), (1, 1), (0, 0)],
                                              inter=True, sbtmvp=True, affine=True, mmvd=True, gpm=True, ciip=True, sao=True, alf=True, ccalf=True, lmcs=True, big_resi=True,
                                              max_mvd=200, p_intra=0.1), 9, 152),
    ("mini_all_tools_ctu64_8bit_320x192", dict(width=320, height=192, log2_ctu=6, log2_min_qt=4, qp=33, bit_depth=8, mtt_depth=3, inter=True, sbtmvp=True, mmvd=True, affine=True,
                                               ciip=True, gpm=True, mrl=True, mip=True, cclm=True, isp=True, lfnst=True, mts=True, sao=True, lmcs=True, jccr=True, dep_quant=True,
                                               alf=True, ccalf=True, p_intra=0.2, p_skip=0.2), 13, 54),
    # intra block copy through the real parser (round 5): block vectors the writer has checked against its model of the decoder's candidate 0 (see coding_unit_ibc) - I pictures,
    # with the intra tools and the filters, in the luma tree of dual-tree pictures, and in the I, P and B pictures of a random-access stream with every inter tool
    ("mini_ibc_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, ibc=True, p_ibc=0.4), 2, 201),
    ("mini_ibc_tools_ctu128_256x256", dict(width=256, height=256, log2_ctu=7, log2_min_qt=4, qp=28, mtt_depth=2, mrl=True, isp=True, mip=True, cclm=True, lfnst=True, mts=True, sao=True, lmcs=True,
                                           jccr=True, dep_quant=True, alf=True, ccalf=True, ts=True, bdpcm=True, ibc=True, p_ibc=0.4), 2, 202),
    ("mini_ibc_dual_tree_ctu128_256x256", dict(width=256, height=256, log2_ctu=7, log2_min_qt=4, qp=30, mtt_depth=2, dual_tree=True, mrl=True, isp=True, mip=True, lfnst=True, mts=True,
                                               ibc=True, p_ibc=0.5), 2, 203),
    # IBC CUs of 64x64 / 64x32 in a sequence whose largest transform is 32: four / two transform units per IBC CU (round 5, finding 13 of DESIGN.md section 3: the randomised
    # GPU leg found such CUs refused by the back-end's record checks)
    # CIIP coding units of several transform units: a largest transform size of 32 splits a 64-wide / 64-high CIIP CU into two or four units - the CU is predicted and
    # blended as a whole, the residuals are added unit by unit (DecCu.cpp:449-470); refused by the back-end until round 6
    ("mini_ciip_tb32_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, log2_min_qt=4, qp=32, max_tb64=False, mtt_depth=1, inter=True, mmvd=True, ciip=True, lmcs=True, jccr=True,
                                           p_split=0.35, p_merge=0.6, p_cbf=0.8, p_cbf_chroma=0.6, p_intra=0.15), 9, 301),
    ("mini_ibc_tb32_8bit_ctu64_192x128", dict(width=192, height=128, log2_ctu=6, qp=35, bit_depth=8, max_tb64=False, p_cbf=0.8, p_cbf_chroma=0.6, p_split=0.3, ibc=True, p_ibc=0.6), 3, 207),
    # CCLM in the chroma tree of dual-tree pictures (round 5: the writer follows CU::checkCCLMAllowed), with IBC in the luma tree and the filters
    ("mini_dual_tree_cclm_ibc_ctu128_256x256", dict(width=256, height=256, log2_ctu=7, log2_min_qt=4, qp=28, mtt_depth=2, dual_tree=True, cclm=True, mrl=True, isp=True, mip=True, lfnst=True,
                                                    mts=True, sao=True, lmcs=True, jccr=True, dep_quant=True, alf=True, ccalf=True, ibc=True, p_ibc=0.3), 2, 205),
    ("mini_dual_tree_cclm_ctu64_256x128", dict(width=256, height=128, log2_ctu=6, qp=30, dual_tree=True, cclm=True, isp=True, mtt_depth=1), 2, 206),
    ("mini_ibc_inter_tools_ctu128_384x256", dict(width=384, height=256, log2_ctu=7, log2_min_qt=4, qp=30, mtt_depth=2, inter=True, sbtmvp=True, mmvd=True, affine=True, ciip=True, gpm=True, amvr=True,
                                                 bcw=True, smvd=True, sbt=True, sao=True, lmcs=True, jccr=True, dep_quant=True, alf=True, p_intra=0.2, ibc=True, p_ibc=0.4), 9, 204),
]


# A stream at a BASELINE size (round 5): 3840x2176 - whole CTUs of 128, the writer makes no implicit splits at the picture boundary -, one GOP of 16 behind the
# I picture, the tool mix of mini_all_tools_ctu128_384x256.  Kept out of FIXTURES: the random sweeps (tools/fuzz_*.py, tests/test_gpu_fuzz.py) draw from that
# list, and 8 000 CUs per picture take the writer a second per picture.  `--big` (or --only mini_4k) writes it.
BIG_FIXTURES = [("mini_4k_all_tools_ctu128_3840x2176", dict(dict([f for f in FIXTURES if f[0] == "mini_all_tools_ctu128_384x256"][0][1]), width=3840, height=2176), 17, 4242)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "bitstreams"))
    ap.add_argument("--only", default=None)
    ap.add_argument("--big", action="store_true", help="also the stream(s) at a BASELINE picture size (BIG_FIXTURES)")
    a = ap.parse_args()
    tables, renorm = load_context_tables()
    for name, kw, n, seed in FIXTURES + (BIG_FIXTURES if a.big or (a.only and any(a.only in f[0] for f in BIG_FIXTURES)) else []):
        if a.only and a.only not in name:
            continue
        c = Cfg(**kw)
        data, stats = write_stream(c, n, seed, tables, renorm)
        d = os.path.join(a.out, name)
        os.makedirs(d, exist_ok=True)
        bit = os.path.join(d, name + ".bit")
        open(bit, "wb").write(data)
        md5, yuv, log = reference_md5(bit, keep=True)
        total = sum(c.sizes[(i // 2) % len(c.sizes) if len(c.sizes) > 1 else 0][0] * c.sizes[(i // 2) % len(c.sizes) if len(c.sizes) > 1 else 0][1] * (2 if c.mono else 3) // 2 * (2 if c.bit_depth > 8 else 1) for i in range(n))
        assert len(yuv) == total, "the reference decoder put out %d bytes, %d pictures of %d bytes in all expected\n%s" % (len(yuv), n, total, log[-1500:])
        # second pass: the same stream with a decoded-picture-hash SEI behind every picture (the hashes are the reference decoder's), checked by the reference decoder itself
        data, stats = write_stream(c, n, seed, tables, renorm, hashes=picture_hashes(c, n, yuv))
        open(bit, "wb").write(data)
        md5b, _, log = reference_md5(bit, extra=["-dph"])
        assert md5b == md5 and not re.search(r"mismatch|\*\*\*ERROR", log), log[-1500:]
        open(os.path.join(d, name + ".yuv.md5"), "w").write("%s  %s.yuv\n" % (md5, name))
        print("%-40s %6d bytes, %d pictures, %s  CUs %s" % (name, len(data), n, md5, [s["cus"] for s in stats]))
        if c.inter:
            print("     " + ", ".join("%s %d" % (k, sum(s[k] for s in stats)) for k in ("skip", "merge", "amvp", "intra", "cbf", "coefs")))


if __name__ == "__main__":
    main()
