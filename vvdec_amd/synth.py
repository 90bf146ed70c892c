"""Synthetic pre-parsed VVC picture stream (measurement/test infrastructure; wraps tools/synth.cpp).

`generate(params)` returns a PictureDesc; `ra_gop(...)` yields the pictures of a hierarchical-B random-access stream
 with DPB slot assignment, in decode order.
"""
import ctypes as C
import os
import subprocess
import numpy as np
from . import abi, desc

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SRC = os.path.join(_ROOT, "tools", "synth.cpp")
_LIB = os.path.join(_ROOT, "tools", "libvvrsynth.so")


def build(force=False):
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "vvr.h")       # (the generator writes vvr.h records and the ABI version)
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(_SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-Wall", _SRC, "-o", _LIB])
    return _LIB


class Params(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("width", C.c_uint16), ("height", C.c_uint16),
                ("bit_depth", C.c_uint8), ("log2_ctu", C.c_uint8), ("chroma_format", C.c_uint8), ("slice_type", C.c_uint8),
                ("tool_flags", C.c_uint32), ("num_ref", C.c_int8 * 2), ("poc", C.c_int32),
                ("ref_poc", C.c_int32 * abi.VVR_MAX_REFS * 2), ("ref_slot", C.c_int16 * abi.VVR_MAX_REFS * 2), ("out_slot", C.c_int16),
                ("base_qp", C.c_int8), ("min_cu_log2", C.c_uint8),
                ("p_intra", C.c_float), ("p_bi", C.c_float), ("p_coded", C.c_float), ("p_coded_chroma", C.c_float),
                ("p_small_corner", C.c_float), ("p_mts", C.c_float), ("p_ts", C.c_float), ("p_lfnst", C.c_float),
                ("p_split_scale", C.c_float), ("mv_sigma", C.c_float),
                ("p_sao", C.c_float), ("p_alf_luma", C.c_float), ("p_alf_chroma", C.c_float), ("p_ccalf", C.c_float),
                ("p_imv_hpel", C.c_float), ("p_jccr", C.c_float), ("p_mrl", C.c_float), ("p_bdpcm", C.c_float),
                ("p_affine", C.c_float), ("p_geo", C.c_float), ("p_ciip", C.c_float), ("p_sbtmvp", C.c_float), ("p_bcw", C.c_float), ("p_cclm", C.c_float), ("p_mip", C.c_float), ("p_sbt", C.c_float), ("p_isp", C.c_float), ("dual_tree", C.c_float), ("p_ibc", C.c_float),
                ("num_slices", C.c_uint8), ("tile_cols", C.c_uint8), ("tile_rows", C.c_uint8), ("wrap_offset", C.c_uint16), ("subpics", C.c_uint8), ("intra_slices", C.c_uint8), ("virtual_boundaries", C.c_uint8), ("scaled_refs", C.c_uint16 * 2), ("mv_window", C.c_uint16)]


class Buffers(C.Structure):
    _fields_ = [("cu", C.c_void_p), ("max_cu", C.c_uint32), ("tu", C.c_void_p), ("max_tu", C.c_uint32),
                ("coef", C.c_void_p), ("max_coef", C.c_uint64), ("ctu_first_cu", C.c_void_p),
                ("motion", C.c_void_p), ("lfp", C.c_void_p * 2), ("sao", C.c_void_p), ("alf", C.c_void_p), ("alf_params", C.c_void_p), ("lmcs", C.c_void_p), ("wp", C.c_void_p), ("scaling", C.c_void_p), ("ctu_slice", C.c_void_p), ("ctu_tile", C.c_void_p), ("subpics", C.c_void_p),
                ("num_cu", C.c_uint32), ("num_tu", C.c_uint32), ("num_coef", C.c_uint64), ("num_dmvr", C.c_uint32), ("num_subpics", C.c_uint32),
                ("hdr", abi.PicHeader)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.vvs_generate.restype = C.c_int
    return _lib


def default_params(**kw):
    p = Params()
    lib().vvs_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def set_refs(p, l0, l1=()):
    for l, lst in enumerate((l0, l1)):
        p.num_ref[l] = len(lst)
        for i, (slot, poc) in enumerate(lst):
            p.ref_slot[l][i] = slot
            p.ref_poc[l][i] = poc


def _compact(a, n, alloc=None):
    """the first n records as an array of their own, copied byte by byte (a field-wise copy would leave the padding bytes of the
    records uninitialised, and descriptions should be reproducible down to the byte)"""
    out = alloc(n, a.dtype) if alloc else np.zeros(n, a.dtype)
    out.view(np.uint8)[:] = a[:n].view(np.uint8)
    return out


def generate(p, alloc=None):
    """Run the generator; returns a desc.PictureDesc owning compact copies of all arrays (alloc: see desc.PictureDesc)."""
    L = lib()
    mcu, mtu, mcoef = C.c_uint32(), C.c_uint32(), C.c_uint64()
    L.vvs_bounds(C.byref(p), C.byref(mcu), C.byref(mtu), C.byref(mcoef))
    d = desc.PictureDesc(p.width, p.height, p.bit_depth, p.log2_ctu, p.chroma_format, p.slice_type, p.poc, p.out_slot, p.tool_flags, alloc=alloc)
    cu = np.zeros(mcu.value, desc.CU_DT)
    tu = np.zeros(mtu.value, desc.TU_DT)
    coef = np.zeros(mcoef.value, np.int16)
    d.motion = np.zeros(d.w4 * d.h4, desc.MOTION_DT)
    d.sao = np.zeros(d.num_ctu, desc.SAO_DT)
    d.alf = np.zeros(d.num_ctu, desc.ALF_DT)
    d.alf_params = abi.AlfParams()
    b = Buffers()
    b.cu, b.max_cu, b.tu, b.max_tu = cu.ctypes.data, mcu.value, tu.ctypes.data, mtu.value
    b.coef, b.max_coef = coef.ctypes.data, mcoef.value
    b.ctu_first_cu = d.ctu_first_cu.ctypes.data
    b.motion = d.motion.ctypes.data
    b.lfp[0], b.lfp[1] = d.lfp[0].ctypes.data, d.lfp[1].ctypes.data
    b.sao, b.alf = d.sao.ctypes.data, d.alf.ctypes.data
    b.alf_params = C.addressof(d.alf_params)
    if p.tool_flags & abi.TOOL_LMCS:
        d.lmcs = abi.LmcsParams()
        b.lmcs = C.addressof(d.lmcs)
    if (p.tool_flags & abi.TOOL_WP) and p.slice_type != abi.SLICE_I:
        d.wp = abi.WpParams()
        b.wp = C.addressof(d.wp)
    if p.tool_flags & abi.TOOL_SCALING_LIST:
        d.scaling = abi.ScalingList()
        b.scaling = C.addressof(d.scaling)
    if p.num_slices > 1:
        d.ctu_slice = np.zeros(d.num_ctu, np.uint16)
        b.ctu_slice = d.ctu_slice.ctypes.data
    if p.tile_cols > 1 or p.tile_rows > 1:
        d.ctu_tile = np.zeros(d.num_ctu, np.uint16)
        b.ctu_tile = d.ctu_tile.ctypes.data
    subpics = None
    if p.subpics & 1:
        subpics = np.zeros(255, np.dtype(abi.Subpic))
        b.subpics = subpics.ctypes.data
        if d.ctu_slice is None:                      # one slice per sub-picture
            d.ctu_slice = np.zeros(d.num_ctu, np.uint16)
            b.ctu_slice = d.ctu_slice.ctypes.data
    rc = L.vvs_generate(C.byref(p), C.byref(b))
    if rc != 0:
        raise RuntimeError("vvs_generate failed (%d)" % rc)
    d.hdr = abi.PicHeader.from_buffer_copy(b.hdr)
    d.cu = _compact(cu, b.num_cu, alloc)
    d.tu = _compact(tu, b.num_tu, alloc)
    d.coef = _compact(coef, max(1, b.num_coef), alloc)
    d.num_dmvr = b.num_dmvr
    if subpics is not None and b.num_subpics:
        d.subpics = subpics[:b.num_subpics].copy()
    return d


def vary_slices(d, seed, alf_sets=2, wp_sets=2, intra_slices=0):
    """Give the slices of a generated multi-slice description headers of their own (vvr_slice_header): each slice draws whether it uses dependent
    quantisation, LMCS, the explicit scaling lists (of the tools the picture has), its deblocking offsets, and which ALF / weight tables it
    refers to.  The extra tables are rearrangements of the generated one (classes, alternatives and filters rolled; weights of the entries that
    are `present` changed) so every value stays in the range the syntax allows and the CUs' mc_mode stays valid.  intra_slices: the mask the
    picture was generated with (Params.intra_slices) - those slices are I slices."""
    assert d.ctu_slice is not None, "a description with more than one slice"
    rng = np.random.default_rng(seed)
    n = int(d.ctu_slice.max()) + 1
    f = int(d.hdr.tool_flags)
    sl = np.zeros(n, np.dtype(abi.SliceHeader))
    for i in range(n):
        t = f & abi.SLICE_TOOL_MASK
        for bit in (abi.TOOL_DEP_QUANT, abi.TOOL_LMCS, abi.TOOL_SCALING_LIST):
            have = bool(f & bit) or bit == abi.TOOL_DEP_QUANT
            on = have and (i == 0 or (i > 1 and rng.random() < 0.5))      # slice 0: everything the picture has, slice 1: nothing
            t = (t | bit) if on else (t & ~bit)
        if not (t & abi.TOOL_LMCS):
            t &= ~abi.TOOL_LMCS_CSCALE
        sl["tool_flags"][i] = t
        sl["deblock_beta_offset_div2"][i] = rng.integers(-6, 7, 3)
        sl["deblock_tc_offset_div2"][i] = rng.integers(-6, 7, 3)
        sl["slice_type"][i] = abi.SLICE_I if (intra_slices >> (i & 7)) & 1 else d.hdr.slice_type
        if sl["slice_type"][i] == abi.SLICE_I:
            sl["tool_flags"][i] = int(sl["tool_flags"][i]) & ~abi.TOOL_WP
        sl["alf_set"][i] = i % alf_sets if d.alf_params is not None else 0
        sl["wp_set"][i] = i % wp_sets if d.wp is not None else 0
    d.slices = sl
    if d.alf_params is not None and alf_sets > 1:
        d.alf_sets = [d.alf_params]
        for k in range(1, alf_sets):
            a = abi.AlfParams.from_buffer_copy(d.alf_params)
            for name, axes in (("luma_coeff", (0, 1)), ("luma_clip", (0, 1)), ("chroma_coeff", (0,)), ("chroma_clip", (0,)), ("ccalf_coeff", (1,))):
                v = np.ctypeslib.as_array(getattr(a, name))
                v[...] = np.roll(v, k, axis=axes)
            d.alf_sets.append(a)
    if d.wp is not None and wp_sets > 1:
        d.wp_sets = [d.wp]
        for k in range(1, wp_sets):
            w = abi.WpParams.from_buffer_copy(d.wp)
            for l in range(2):
                for i in range(abi.VVR_MAX_REFS):
                    for c in range(3):
                        e = w.e[l][i][c]
                        if e.present:
                            e.weight = int(np.clip(e.weight + (k if (i + c) & 1 else -k) * 3, (1 << w.log2_denom[1 if c else 0]) - 128, (1 << w.log2_denom[1 if c else 0]) + 127))
                            e.offset = int(np.clip(-e.offset + k, -128, 127))
            d.wp_sets.append(w)
    return d


def attach_rpr(d, refs, win=(0, 0), colloc=(1, 1)):
    """Reference picture resampling: give a generated description its vvr_rpr_params.  refs: {(list, idx): dict(ratio=(rx, ry), size=(w, h), win=(left, top))}
    names the scaled reference pictures (the description must have been generated with Params.scaled_refs naming the same ones); every other reference
    picture of the lists has the current picture's size and window.  win: the current picture's scaling window offsets (luma samples);
    colloc: sps_chroma_horizontal / vertical_collocated_flag of the reference pictures' SPS."""
    r = abi.RprParams()
    r.win_left, r.win_top = win
    for l in range(2):
        for i in range(d.hdr.num_ref[l]):
            e = r.ref[l][i]
            spec = refs.get((l, i))
            e.ratio[0], e.ratio[1] = spec["ratio"] if spec else (1 << 14, 1 << 14)
            e.width, e.height = spec.get("size", (d.hdr.width, d.hdr.height)) if spec else (d.hdr.width, d.hdr.height)
            e.win_left, e.win_top = spec.get("win", win) if spec else win
            e.scaled = 1 if spec else 0
            e.hor_collocated_chroma, e.ver_collocated_chroma = colloc
    d.rpr = r
    return d


def picture_for_plan(plan, width, height, seed=1234, tool_flags=0, alloc=None, **kw):
    """PictureDesc of one stream.PicPlan (generator seed = base seed + POC)."""
    p = default_params(width=width, height=height, seed=seed + plan.poc, tool_flags=tool_flags, slice_type=plan.slice_type, **kw)
    p.poc = plan.poc
    p.out_slot = plan.slot
    if plan.slice_type != abi.SLICE_I:
        set_refs(p, plan.ref_slots[0], plan.ref_slots[1])
    return generate(p, alloc)


def natural_picture(width, height, seed, bit_depth=10):
    """A smooth-plus-texture test picture (stands in for an IRAP picture until intra reconstruction is enabled)."""
    rng = np.random.default_rng(seed)
    mx = (1 << bit_depth) - 1
    gy, gx = np.mgrid[0:height, 0:width]
    base = (np.sin(gx / 37.0 + seed) + np.cos(gy / 23.0 - seed) + np.sin((gx + gy) / 61.0)) * (mx / 8.0) + mx / 2.0
    blocks = rng.integers(-mx // 10, mx // 10, (height // 16 + 1, width // 16 + 1)).repeat(16, 0).repeat(16, 1)[:height, :width]
    y = np.clip(base + blocks + rng.integers(-12, 13, (height, width)), 0, mx).astype(np.uint16)
    cb = np.clip(y[::2, ::2].astype(np.int32) // 2 + mx // 4 + rng.integers(-6, 7, (height // 2, width // 2)), 0, mx).astype(np.uint16)
    cr = np.clip(mx - y[::2, ::2].astype(np.int32) // 2 - mx // 4 + rng.integers(-6, 7, (height // 2, width // 2)), 0, mx).astype(np.uint16)
    return [y, cb, cr]
