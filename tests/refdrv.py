"""ctypes wrapper of oracle/_ref/libvvref.so (the unmodified reference classes driven by oracle/ref_harness.cpp).

TEST INFRASTRUCTURE ONLY.  Available only where oracle/_ref has been built (needs /root/reference at build time);
tests that use it skip otherwise and rely on the committed golden fixtures it produced (tests/golden/)."""
import ctypes as C
import os
import numpy as np
from vvdec_amd import abi

_LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libvvref.so")
SIMD, DERIVE_LFP, STOP_AFTER_RECO, STOP_AFTER_DBK, STOP_AFTER_SAO, SPAN_AFFINE = 1, 2, 4, 8, 16, 32
ROTATE_REF_LISTS = 64      # VVREF_ROTATE_REF_LISTS: every slice holds the description's reference lists rotated by its index (the extractor merges them again)
_lib = None


def available():
    return os.path.exists(_LIB)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_LIB)
        _lib.vvref_reconstruct.restype = C.c_int
        _lib.vvref_last_error.restype = C.c_char_p
    return _lib


def reconstruct(desc, refs=None, flags=0, want_lfp=False, want_dmvr=0):
    """desc: PictureDesc; refs: {slot: [Y, Cb, Cr] uint16 arrays}. Returns dict(planes=[...], ms=..., lfp=..., dmvr=...)."""
    L = lib()
    p = desc.c(keep_lfp=True)
    nslots = (max(refs.keys()) + 1) if refs else 0
    ref_ptrs = (C.POINTER(C.c_uint16) * max(1, nslots * 3))()
    keep = []
    for slot, planes in (refs or {}).items():
        for c, pl in enumerate(planes):
            a = np.ascontiguousarray(pl, dtype=np.uint16)
            keep.append(a)
            ref_ptrs[slot * 3 + c] = a.ctypes.data_as(C.POINTER(C.c_uint16))
    ncomp = 3 if desc.hdr.chroma_format else 1
    outs = [np.zeros(desc.plane_shape(c), np.uint16) for c in range(ncomp)]
    out_ptrs = (C.POINTER(C.c_uint16) * 3)()
    for c in range(ncomp):
        out_ptrs[c] = outs[c].ctypes.data_as(C.POINTER(C.c_uint16))
    lfp_ptrs = (C.POINTER(abi.Lfp) * 2)()
    lfps = None
    if want_lfp:
        lfps = [np.zeros(desc.w4 * desc.h4, np.dtype(abi.Lfp)) for _ in range(2)]
        for d in range(2):
            lfp_ptrs[d] = lfps[d].ctypes.data_as(C.POINTER(abi.Lfp))
    dmvr = np.zeros((max(1, want_dmvr), 2), np.int32)
    ms = (C.c_double * 8)()
    rc = L.vvref_reconstruct(C.byref(p), ref_ptrs, out_ptrs, lfp_ptrs if want_lfp else None,
                             dmvr.ctypes.data_as(C.POINTER(C.c_int32)) if want_dmvr else None, flags, ms)
    if rc != 0:
        raise RuntimeError("vvref_reconstruct failed: " + L.vvref_last_error().decode())
    return dict(planes=outs, ms=list(ms), lfp=lfps, dmvr=dmvr)


def reconstruct_threaded(desc, refs=None, threads=4):
    """the picture through the reference's OWN scheduler: DecLibRecon's per-picture set-up and its 15-state CTU task (DecLibRecon.cpp:429-1110) on its
    ThreadPool with `threads` threads (0: the calling thread), the tasks starting at LF_INIT (the harness hands over final motion).
    -> dict(planes, ms = wall clock of decompressPicture + waitForPrevDecompressedPic)"""
    L = lib()
    p = desc.c(keep_lfp=True)
    ref_ptrs, keep, _ = _ref_ptrs(refs or {})
    ncomp = 3 if desc.hdr.chroma_format else 1
    outs = [np.zeros(desc.plane_shape(c), np.uint16) for c in range(ncomp)]
    out_ptrs = (C.POINTER(C.c_uint16) * 3)()
    for c in range(ncomp):
        out_ptrs[c] = outs[c].ctypes.data_as(C.POINTER(C.c_uint16))
    ms = C.c_double()
    L.vvref_reconstruct_threaded.restype = C.c_int
    rc = L.vvref_reconstruct_threaded(C.byref(p), ref_ptrs, out_ptrs, int(threads), C.byref(ms))
    if rc != 0:
        raise RuntimeError("vvref_reconstruct_threaded failed: " + L.vvref_last_error().decode())
    return dict(planes=outs, ms=ms.value)


def _ref_ptrs(refs):
    nslots = (max(refs.keys()) + 1) if refs else 0
    ref_ptrs = (C.POINTER(C.c_uint16) * max(1, nslots * 3))()
    keep = []
    for slot, planes in (refs or {}).items():
        for c, pl in enumerate(planes):
            a = np.ascontiguousarray(pl, dtype=np.uint16)
            keep.append(a)
            ref_ptrs[slot * 3 + c] = a.ctypes.data_as(C.POINTER(C.c_uint16))
    return ref_ptrs, keep, nslots


def reconstruct_with_motion(desc, refs=None, flags=0):
    """the reference's own stages -> (planes, motion field after DecCu::TaskFinishMotionInfo as an array of abi.Motion, picture raster 4x4 grid)"""
    L = lib()
    p = desc.c(keep_lfp=True)
    ref_ptrs, keep, _ = _ref_ptrs(refs)
    ncomp = 3 if desc.hdr.chroma_format else 1
    outs = [np.zeros(desc.plane_shape(c), np.uint16) for c in range(ncomp)]
    out_ptrs = (C.POINTER(C.c_uint16) * 3)()
    for c in range(ncomp):
        out_ptrs[c] = outs[c].ctypes.data_as(C.POINTER(C.c_uint16))
    motion = np.zeros(desc.w4 * desc.h4, np.dtype(abi.Motion))
    L.vvref_reconstruct_with_motion.restype = C.c_int
    rc = L.vvref_reconstruct_with_motion(C.byref(p), ref_ptrs, out_ptrs, motion.ctypes.data_as(C.c_void_p), flags)
    if rc != 0:
        raise RuntimeError("vvref_reconstruct_with_motion failed: " + L.vvref_last_error().decode())
    return outs, motion


_BIND = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libvvrefbind.so")
_bind = None


def binding_available():
    return os.path.exists(_BIND)


def run_binding(desc, refs, backend_path, num_slots=8, flags=0):
    """the picture through integration/DecLibReconAmd.h (the DecLibRecon replacement: extractor -> vvr_submit -> vvr_wait -> TaskFinishMotionInfo),
    executed on the back-end library `backend_path` (libvvdec_amd.so, or the stand-in build of the CPU tests).  -> (planes, motion field)"""
    global _bind
    if _bind is None:
        C.CDLL(backend_path, mode=C.RTLD_GLOBAL)        # the harness's vvr_* calls bind to this library
        _bind = C.CDLL(_BIND)
        _bind.vvref_run_binding.restype = C.c_int
        _bind.vvref_last_error.restype = C.c_char_p
    p = desc.c(keep_lfp=True)
    ref_ptrs, keep, nslots = _ref_ptrs(refs)
    ncomp = 3 if desc.hdr.chroma_format else 1
    outs = [np.zeros(desc.plane_shape(c), np.uint16) for c in range(ncomp)]
    out_ptrs = (C.POINTER(C.c_uint16) * 3)()
    for c in range(ncomp):
        out_ptrs[c] = outs[c].ctypes.data_as(C.POINTER(C.c_uint16))
    motion = np.zeros(desc.w4 * desc.h4, np.dtype(abi.Motion))
    _bind.vvref_set_extra_flags(flags)
    try:
        rc = _bind.vvref_run_binding(C.byref(p), ref_ptrs, out_ptrs, motion.ctypes.data_as(C.c_void_p), max(num_slots, nslots + 2))
    finally:
        _bind.vvref_set_extra_flags(0)
    if rc != 0:
        raise RuntimeError("vvref_run_binding failed: " + _bind.vvref_last_error().decode())
    return outs, motion


_DROPIN_HARNESS = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libvvrefdropin.so")
DROPIN_LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libvvdec.so")
_dropin = None


def dropin_available():
    return os.path.exists(_DROPIN_HARNESS) and os.path.exists(DROPIN_LIB)


def run_dropin(desc, refs, backend_path, threads=2, flags=0):
    """the picture through the DROP-IN: the reference's class vvdec::DecLibRecon with the member functions of integration/DecLibReconDropIn.cpp
    (create( ThreadPool*, id, upscale ) / decompressPicture / waitForPrevDecompressedPic / destroy on a thread pool of `threads` threads), on the
    back-end library `backend_path`.  -> (planes as they sit in the Picture's own buffers, motion field)"""
    global _dropin
    if _dropin is None:
        C.CDLL(backend_path, mode=C.RTLD_GLOBAL)        # the drop-in's vvr_* calls bind to this library
        _dropin = C.CDLL(_DROPIN_HARNESS)
        _dropin.vvref_run_dropin.restype = C.c_int
        _dropin.vvref_last_error.restype = C.c_char_p
    p = desc.c(keep_lfp=True)
    ref_ptrs, keep, nslots = _ref_ptrs(refs)
    ncomp = 3 if desc.hdr.chroma_format else 1
    outs = [np.zeros(desc.plane_shape(c), np.uint16) for c in range(ncomp)]
    out_ptrs = (C.POINTER(C.c_uint16) * 3)()
    for c in range(ncomp):
        out_ptrs[c] = outs[c].ctypes.data_as(C.POINTER(C.c_uint16))
    motion = np.zeros(desc.w4 * desc.h4, np.dtype(abi.Motion))
    _dropin.vvref_set_extra_flags(flags)              # (e.g. ROTATE_REF_LISTS: what the harness builds, not what the drop-in does)
    try:
        rc = _dropin.vvref_run_dropin(C.byref(p), ref_ptrs, out_ptrs, motion.ctypes.data_as(C.c_void_p), threads)
    finally:
        _dropin.vvref_set_extra_flags(0)
    if rc != 0:
        raise RuntimeError("vvref_run_dropin failed: " + _dropin.vvref_last_error().decode())
    return outs, motion


def extract(desc, refs=None, flags=0):
    """description -> the reference decoder's own objects -> description, through the reference-side glue integration/vvr_extract.h.
    Returns a dict of numpy copies of every array of the extracted vvr_picture (and its header)."""
    L = lib()
    L.vvref_extract.restype = C.POINTER(abi.Picture)
    p = desc.c(keep_lfp=True)
    nslots = (max(refs.keys()) + 1) if refs else 0
    ref_ptrs = (C.POINTER(C.c_uint16) * max(1, nslots * 3))()
    keep = []
    for slot, planes in (refs or {}).items():
        for c, pl in enumerate(planes):
            a = np.ascontiguousarray(pl, dtype=np.uint16)
            keep.append(a)
            ref_ptrs[slot * 3 + c] = a.ctypes.data_as(C.POINTER(C.c_uint16))
    nd = C.c_uint32()
    r = L.vvref_extract(C.byref(p), ref_ptrs, C.byref(nd), flags)      # flags: DERIVE_LFP = the reference derives the edge parameters itself
    if not r:
        raise RuntimeError("vvref_extract failed: " + L.vvref_last_error().decode())
    q = r.contents
    h = abi.PicHeader.from_buffer_copy(q.hdr)
    n4 = ((h.width + 3) >> 2) * ((h.height + 3) >> 2)
    ctu = 1 << h.log2_ctu
    nctu = ((h.width + ctu - 1) // ctu) * ((h.height + ctu - 1) // ctu)

    def arr(ptr, n, ctype):
        if not ptr or not n:
            return None
        return np.frombuffer((C.c_char * (C.sizeof(ctype) * n)).from_address(C.addressof(ptr.contents)), np.dtype(ctype)).copy()

    def one(ptr, ctype):
        return ctype.from_buffer_copy(ptr.contents) if ptr else None

    def many(ptr, n, ctype):
        return [ctype.from_buffer_copy(ptr[k]) for k in range(n)] if ptr and n else None

    return dict(rpr=one(q.rpr, abi.RprParams), slices=arr(q.slices, q.num_slices, abi.SliceHeader), alf_sets=many(q.alf_params, q.num_alf_sets, abi.AlfParams), wp_sets=many(q.wp, q.num_wp_sets, abi.WpParams),
                hdr=h, num_dmvr=nd.value, cu=arr(q.cu, q.num_cu, abi.Cu), tu=arr(q.tu, q.num_tu, abi.Tu), coef=arr(q.coef, q.num_coef, abi.i16),
                ctu_first_cu=arr(q.ctu_first_cu, nctu + 1, abi.u32), motion=arr(q.motion, n4, abi.Motion),
                lfp=[arr(q.lfp[0], n4, abi.Lfp), arr(q.lfp[1], n4, abi.Lfp)], sao=arr(q.sao, nctu, abi.SaoCtu), alf=arr(q.alf, nctu, abi.AlfCtu),
                alf_params=one(q.alf_params, abi.AlfParams), lmcs=one(q.lmcs, abi.LmcsParams), wp=one(q.wp, abi.WpParams), scaling=one(q.scaling, abi.ScalingList),
                ctu_slice=arr(q.ctu_slice, nctu, abi.u16), ctu_tile=arr(q.ctu_tile, nctu, abi.u16),
                subpics=(np.frombuffer((C.c_char * (C.sizeof(abi.Subpic) * q.num_subpics)).from_address(q.subpics), np.dtype(abi.Subpic)).copy() if q.subpics and q.num_subpics else None))


def desc_from_extract(e):
    """the dict extract() returns as a PictureDesc again (to run the oracle or the back-end on what the extractor wrote)"""
    from vvdec_amd import desc as _desc
    h = e["hdr"]
    x = _desc.PictureDesc(h.width, h.height, h.bit_depth, h.log2_ctu, h.chroma_format)
    x.hdr = h
    for k in ("coef", "ctu_first_cu", "sao", "alf", "ctu_slice", "ctu_tile", "slices", "alf_sets", "wp_sets", "lmcs", "scaling", "subpics", "alf_params", "wp", "rpr"):
        setattr(x, k, e[k])
    x.cu, x.tu, x.motion = e["cu"].view(_desc.CU_DT), e["tu"].view(_desc.TU_DT), e["motion"].view(_desc.MOTION_DT)
    x.lfp = [e["lfp"][0].view(_desc.LFP_DT), e["lfp"][1].view(_desc.LFP_DT)]
    x.sao = None if e["sao"] is None else e["sao"].view(_desc.SAO_DT)
    x.alf = None if e["alf"] is None else e["alf"].view(_desc.ALF_DT)
    x.num_dmvr = e["num_dmvr"]
    return x


# ---------------------------------------------------------------------------------------------------------------------
# the plain-C restatement (oracle/libvvoracle.so) through the same calling convention
# ---------------------------------------------------------------------------------------------------------------------
_OLIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "libvvoracle.so")
_olib = None


def oracle_lib():
    global _olib
    if _olib is None:
        if not os.path.exists(_OLIB):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.dirname(_OLIB), "oracle"], stdout=subprocess.DEVNULL)
        _olib = C.CDLL(_OLIB)
        _olib.vvo_reconstruct.restype = C.c_int
        _olib.vvo_last_error.restype = C.c_char_p
    return _olib


def oracle_reconstruct(desc, refs=None, flags=0):
    L = oracle_lib()
    p = desc.c(keep_lfp=True)
    nslots = (max(refs.keys()) + 1) if refs else 0
    ref_ptrs = (C.POINTER(C.c_uint16) * max(1, nslots * 3))()
    keep = []
    for slot, planes in (refs or {}).items():
        for c, pl in enumerate(planes):
            a = np.ascontiguousarray(pl, dtype=np.uint16)
            keep.append(a)
            ref_ptrs[slot * 3 + c] = a.ctypes.data_as(C.POINTER(C.c_uint16))
    ncomp = 3 if desc.hdr.chroma_format else 1
    outs = [np.zeros(desc.plane_shape(c), np.uint16) for c in range(ncomp)]
    out_ptrs = (C.POINTER(C.c_uint16) * 3)()
    for c in range(ncomp):
        out_ptrs[c] = outs[c].ctypes.data_as(C.POINTER(C.c_uint16))
    rc = L.vvo_reconstruct(C.byref(p), ref_ptrs, out_ptrs, flags)
    if rc != 0:
        raise RuntimeError("vvo_reconstruct failed: " + L.vvo_last_error().decode())
    return outs


def oracle_dmvr(n):
    """delta MVs (n x 2, 1/16 sample) of the last oracle_reconstruct call"""
    L = oracle_lib()
    L.vvo_get_dmvr.restype = C.c_uint32
    a = np.zeros((max(1, n), 2), np.int32)
    L.vvo_get_dmvr(a.ctypes.data_as(C.POINTER(C.c_int32)), n)
    return a[:n]


# ---- decoded picture hash, restated (test infrastructure; pinned against the reference's functions in tests/test_oracle_vs_ref.py) ---------------
_CRC_T = None


def hash_crc(plane, bit_depth):
    """compCRC (CommonLib/PicYuvMD5.cpp:99-138): CRC-16 / 0x1021, start 0xffff, message bits shifted in at the bottom - low byte, then (bit depth
    > 8) high byte of every sample in raster order - and 16 zero bits at the end; 2 digest bytes, high byte first"""
    global _CRC_T
    if _CRC_T is None:
        _CRC_T = []
        for t in range(256):
            v = t << 8
            for _ in range(8):
                v = ((v << 1) & 0xffff) ^ (0x1021 if v & 0x8000 else 0)
            _CRC_T.append(v)
    a = np.ascontiguousarray(plane, dtype=np.uint16)
    data = (a.astype("<u2").tobytes() if bit_depth > 8 else a.astype(np.uint8).tobytes()) + b"\0\0"
    crc = 0xffff
    for b in data:
        crc = ((((crc << 8) & 0xffff) | b) ^ _CRC_T[crc >> 8])
    return bytes([(crc >> 8) & 0xff, crc & 0xff])


def hash_checksum(plane, bit_depth):
    """compChecksum (CommonLib/PicYuvMD5.cpp:153-179): sum over the samples of (low byte ^ m) [+ (high byte ^ m) when the bit depth exceeds 8],
    m = (x & 255) ^ (y & 255) ^ (x >> 8) ^ (y >> 8) as uint8, modulo 2^32; 4 digest bytes, most significant first"""
    a = np.ascontiguousarray(plane, dtype=np.uint16).astype(np.uint64)
    h, w = a.shape
    y, x = np.mgrid[0:h, 0:w]
    m = ((x & 0xff) ^ (y & 0xff) ^ (x >> 8) ^ (y >> 8)).astype(np.uint64) & 0xff
    s = int(((a & 0xff) ^ m).sum())
    if bit_depth > 8:
        s += int(((a >> 8) ^ m).sum())
    s &= 0xffffffff
    return bytes([(s >> 24) & 0xff, (s >> 16) & 0xff, (s >> 8) & 0xff, s & 0xff])


def hash_md5(plane, bit_depth):
    """calcMD5 (CommonLib/PicYuvMD5.cpp:197-221): MD5 over the samples in raster order, one byte per sample up to 8 bits, else two (little endian)"""
    import hashlib
    a = np.ascontiguousarray(plane, dtype=np.uint16)
    return hashlib.md5(a.astype(np.uint8).tobytes() if bit_depth <= 8 else a.astype("<u2").tobytes()).digest()


def picture_hash(planes, bit_depth, method):
    """per-component digests of a picture: method 0 MD5, 1 CRC, 2 checksum (the hash types of the decoded-picture-hash SEI)"""
    f = (hash_md5, hash_crc, hash_checksum)[method]
    return [f(p, bit_depth) for p in planes]
