"""Host-side container for one pre-parsed picture (numpy arrays <-> the C `vvr_picture`).

Mirrors what the reference hands to DecLibRecon::decompressPicture (a parsed `Picture` with its CodingStructure,
DecLibRecon.cpp:429): CU/TU records, packed levels, motion field, deblocking edge parameters, SAO/ALF controls.
"""
import ctypes as C
import numpy as np
from . import abi

CU_DT = np.dtype(abi.Cu)
TU_DT = np.dtype(abi.Tu)
MOTION_DT = np.dtype(abi.Motion)
LFP_DT = np.dtype(abi.Lfp)
SAO_DT = np.dtype(abi.SaoCtu)
ALF_DT = np.dtype(abi.AlfCtu)


class PictureDesc:
    """Owns the arrays of one picture description; `.c()` returns a ctypes Picture that points into them."""

    def __init__(self, width, height, bit_depth=10, log2_ctu=7, chroma_format=1, slice_type=abi.SLICE_I, poc=0, out_slot=0, tool_flags=0, alloc=None):
        h = abi.PicHeader()
        h.abi_version = abi.VVR_ABI_VERSION
        h.width, h.height, h.bit_depth, h.log2_ctu, h.chroma_format = width, height, bit_depth, log2_ctu, chroma_format
        h.slice_type, h.poc, h.out_slot, h.tool_flags = slice_type, poc, out_slot, tool_flags
        h.min_qp_ts = 4
        self.hdr = h
        self.w4, self.h4 = (width + 3) // 4, (height + 3) // 4
        ctu = 1 << log2_ctu
        self.ctus_x, self.ctus_y = (width + ctu - 1) // ctu, (height + ctu - 1) // ctu
        self.num_ctu = self.ctus_x * self.ctus_y
        self.cu = np.zeros(0, CU_DT)
        self.tu = np.zeros(0, TU_DT)
        self.ctu_first_cu = np.zeros(self.num_ctu + 1, np.uint32)
        self.coef = np.zeros(0, np.int16)
        self.motion = None
        # alloc(n, dtype): where the large arrays live (default: ordinary numpy memory; Reconstructor.host_array: memory the device reads directly)
        self.alloc = alloc or (lambda n, dt: np.zeros(n, dt))
        self.lfp = [self.alloc(self.w4 * self.h4, LFP_DT), self.alloc(self.w4 * self.h4, LFP_DT)]
        for a in self.lfp:
            a.view(np.uint8)[:] = 0
        self.sao = None
        self.alf = None
        self.alf_params = None
        self.lmcs = None
        self.wp = None
        self.scaling = None
        self.ctu_slice = None          # uint16 [num_ctu] or None (one slice)
        self.ctu_tile = None           # uint16 [num_ctu] or None (one tile)
        self.subpics = None            # array of abi.Subpic records or None (the picture is its only sub-picture)
        self.slices = None             # array of abi.SliceHeader records (indexed by ctu_slice) or None: every slice takes the picture header's values
        self.alf_sets = None           # list of abi.AlfParams selected by SliceHeader.alf_set (None: the single table alf_params)
        self.wp_sets = None            # list of abi.WpParams selected by SliceHeader.wp_set (None: the single table wp)
        self.rpr = None                # abi.RprParams or None: no reference picture is scaled

    def set_refs(self, l0, l1=()):
        """l0/l1: lists of (slot, poc)."""
        for l, lst in enumerate((l0, l1)):
            self.hdr.num_ref[l] = len(lst)
            for i, (slot, poc) in enumerate(lst):
                self.hdr.ref_slot[l][i] = slot
                self.hdr.ref_poc[l][i] = poc

    def c(self, keep_lfp=False):
        """the C view.  keep_lfp: hand the edge-parameter tables over even when the header says the back-end derives them itself (VVR_TOOL_LFP_ON_DEVICE):
        the checkers - oracle, reference classes - always take the tables"""
        p = abi.Picture()
        p.hdr = self.hdr
        p.num_cu, p.num_tu = len(self.cu), len(self.tu)
        self.cu = np.ascontiguousarray(self.cu)
        self.tu = np.ascontiguousarray(self.tu)
        self.coef = np.ascontiguousarray(self.coef, dtype=np.int16)
        if len(self.coef) == 0:
            self.coef = np.zeros(1, np.int16)
        p.cu = self.cu.ctypes.data_as(C.POINTER(abi.Cu))
        p.tu = self.tu.ctypes.data_as(C.POINTER(abi.Tu))
        p.ctu_first_cu = self.ctu_first_cu.ctypes.data_as(C.POINTER(abi.u32))
        p.coef = self.coef.ctypes.data_as(C.POINTER(abi.i16))
        p.num_coef = len(self.coef)
        if self.motion is not None:
            p.motion = self.motion.ctypes.data_as(C.POINTER(abi.Motion))
        if keep_lfp or not (self.hdr.tool_flags & abi.TOOL_LFP_ON_DEVICE):       # (with the flag the back-end derives the edge parameters itself: the tables stay at home)
            for d in range(2):
                p.lfp[d] = self.lfp[d].ctypes.data_as(C.POINTER(abi.Lfp))
        if self.sao is not None:
            p.sao = self.sao.ctypes.data_as(C.POINTER(abi.SaoCtu))
        if self.alf is not None:
            p.alf = self.alf.ctypes.data_as(C.POINTER(abi.AlfCtu))
        if self.alf_sets:
            self._alf_arr = (abi.AlfParams * len(self.alf_sets))(*self.alf_sets)
            p.alf_params = C.cast(self._alf_arr, C.POINTER(abi.AlfParams))
            p.num_alf_sets = len(self.alf_sets)
        elif self.alf_params is not None:
            p.alf_params = C.pointer(self.alf_params)
            p.num_alf_sets = 1
        if self.lmcs is not None:
            p.lmcs = C.pointer(self.lmcs)
        if self.wp_sets:
            self._wp_arr = (abi.WpParams * len(self.wp_sets))(*self.wp_sets)
            p.wp = C.cast(self._wp_arr, C.POINTER(abi.WpParams))
            p.num_wp_sets = len(self.wp_sets)
        elif self.wp is not None:
            p.wp = C.pointer(self.wp)
            p.num_wp_sets = 1
        if self.scaling is not None:
            p.scaling = C.pointer(self.scaling)
        if self.ctu_slice is not None:
            self.ctu_slice = np.ascontiguousarray(self.ctu_slice, dtype=np.uint16)
            p.ctu_slice = self.ctu_slice.ctypes.data_as(C.POINTER(abi.u16))
        if self.ctu_tile is not None:
            self.ctu_tile = np.ascontiguousarray(self.ctu_tile, dtype=np.uint16)
            p.ctu_tile = self.ctu_tile.ctypes.data_as(C.POINTER(abi.u16))
        if self.subpics is not None and len(self.subpics):
            self.subpics = np.ascontiguousarray(self.subpics, dtype=np.dtype(abi.Subpic))
            p.subpics = self.subpics.ctypes.data
            p.num_subpics = len(self.subpics)
        if self.slices is not None and len(self.slices):
            self.slices = np.ascontiguousarray(self.slices, dtype=np.dtype(abi.SliceHeader))
            p.slices = C.cast(self.slices.ctypes.data, C.POINTER(abi.SliceHeader))
            p.num_slices = len(self.slices)
        if self.rpr is not None:
            p.rpr = C.pointer(self.rpr)
        p.resident = 0
        self._keep = p
        return p

    def plane_shape(self, comp):
        return (self.hdr.height >> (1 if comp else 0), self.hdr.width >> (1 if comp else 0))
