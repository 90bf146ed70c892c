"""ctypes mirror of include/vvr.h (the C ABI of the reconstruction back-end).

Field names, order and widths follow include/vvr.h one to one; `tests/test_abi.py` checks sizeof/offsetof against the
compiled library so the two cannot drift apart silently.
"""
import ctypes as C

VVR_ABI_VERSION = 5
VVR_MAX_REFS = 16
VVR_MAX_ALF_APS = 8
VVR_ALF_CLASSES = 25
VVR_ALF_LUMA_TAPS = 13
VVR_ALF_CHR_TAPS = 7
VVR_ALF_MAX_CHR_ALT = 8
VVR_CCALF_FILTERS = 4
VVR_CCALF_TAPS = 7

# status codes
VVR_OK, VVR_ERR_UNSPECIFIED, VVR_ERR_PARAMETER, VVR_ERR_UNSUPPORTED, VVR_ERR_DEVICE, VVR_ERR_NO_DEVICE, VVR_ERR_BUSY = 0, -1, -2, -3, -4, -5, -6
VVR_NOT_READY = 1          # non-blocking stream-order queries: ask again

# tool flags
TOOL_SAO_LUMA, TOOL_SAO_CHROMA, TOOL_ALF, TOOL_CCALF, TOOL_LMCS, TOOL_LMCS_CSCALE, TOOL_DEBLOCK_OFF, TOOL_DEP_QUANT, \
    TOOL_BDOF, TOOL_DMVR, TOOL_PROF, TOOL_JCCR_SIGN, TOOL_STILL_REF, TOOL_LFNST, TOOL_MTS, TOOL_CCLM_COLLOC, \
    TOOL_WP, TOOL_SCALING_LIST, TOOL_SCALING_LIST_NO_LFNST, TOOL_IMPLICIT_MTS, TOOL_IBC, TOOL_LADF, TOOL_NO_LF_ACROSS_SLICES, TOOL_NO_LF_ACROSS_TILES, TOOL_AFFINE_MV_ON_DEVICE, TOOL_COL_MOTION, TOOL_LFP_ON_DEVICE = [1 << i for i in range(27)]
SLICE_TOOL_MASK = TOOL_DEP_QUANT | TOOL_LMCS | TOOL_LMCS_CSCALE | TOOL_SCALING_LIST | TOOL_WP       # VVR_SLICE_TOOL_MASK: the switches a vvr_slice_header carries

PRED_INTER, PRED_INTRA, PRED_IBC = 0, 1, 2
TREE_JOINT, TREE_LUMA, TREE_CHROMA = 0, 1, 2
CU_ROOT_CBF, CU_SKIP, CU_MERGE, CU_AFFINE, CU_AFFINE_6P, CU_CIIP, CU_GEO, CU_SBTMVP, CU_MIP, CU_MIP_TRANSP, CU_SMVD, CU_MMVD = [1 << i for i in range(12)]
MC_NONE, MC_UNI, MC_BI, MC_BDOF, MC_DMVR, MC_DMVR_BDOF, MC_AFFINE, MC_SBTMVP, MC_GEO = range(9)
MTS_DCT2, MTS_SKIP, MTS_DST7_DST7, MTS_DCT8_DST7, MTS_DST7_DCT8, MTS_DCT8_DCT8 = range(6)
TR_DCT2, TR_DCT8, TR_DST7 = 0, 1, 2
SLICE_B, SLICE_P, SLICE_I = 0, 1, 2
STOP_NONE, STOP_RECO, STOP_DEBLOCK, STOP_SAO = 0, 1, 2, 3

u8, i8, u16, i16, u32, i32, u64 = C.c_uint8, C.c_int8, C.c_uint16, C.c_int16, C.c_uint32, C.c_int32, C.c_uint64


class AlfParams(C.Structure):
    _fields_ = [("luma_coeff", i16 * VVR_ALF_LUMA_TAPS * VVR_ALF_CLASSES * VVR_MAX_ALF_APS),
                ("luma_clip", i16 * VVR_ALF_LUMA_TAPS * VVR_ALF_CLASSES * VVR_MAX_ALF_APS),
                ("chroma_coeff", i16 * VVR_ALF_CHR_TAPS * VVR_ALF_MAX_CHR_ALT),
                ("chroma_clip", i16 * VVR_ALF_CHR_TAPS * VVR_ALF_MAX_CHR_ALT),
                ("ccalf_coeff", i16 * (VVR_CCALF_TAPS + 1) * VVR_CCALF_FILTERS * 2),
                ("num_luma_aps", u8), ("pad", u8 * 7)]


class LmcsParams(C.Structure):
    _fields_ = [("fwd_lut", i16 * 4096), ("inv_lut", i16 * 4096), ("chroma_scale", i16 * 16), ("pivot", i16 * 17), ("min_bin", i16), ("max_bin", i16),
                ("model_delta_cw", i16 * 16), ("model_delta_crs", i16), ("pad", i16 * 4)]


class PicHeader(C.Structure):
    _fields_ = [("abi_version", u32), ("tool_flags", u32), ("width", u16), ("height", u16),
                ("chroma_format", u8), ("bit_depth", u8), ("log2_ctu", u8), ("slice_type", u8),
                ("poc", i32), ("out_slot", i16), ("num_ref", i8 * 2),
                ("ref_slot", i16 * VVR_MAX_REFS * 2), ("ref_poc", i32 * VVR_MAX_REFS * 2),
                ("deblock_beta_offset_div2", i8 * 3), ("deblock_tc_offset_div2", i8 * 3),
                ("log2_sao_offset_scale", u8 * 2), ("min_qp_ts", i8),
                ("ladf_num_intervals", u8), ("ladf_qp_offset", i8 * 5), ("pad", u8), ("ladf_lower_bound", i16 * 5),
                ("num_ver_vb", u8), ("num_hor_vb", u8), ("wrap_offset", u16), ("pad2", u8 * 2), ("vb_pos_x", u16 * 3), ("vb_pos_y", u16 * 3), ("pad3", u8 * 4)]


class Cu(C.Structure):
    _fields_ = [("x", u16), ("y", u16), ("w", u8), ("h", u8), ("tree", u8), ("pred_mode", u8),
                ("flags", u16), ("qp", i8), ("mc_mode", u8),
                ("intra_dir", u8 * 2), ("multi_ref_idx", u8), ("isp_mode", u8), ("bdpcm", u8 * 2), ("lfnst_idx", u8), ("sbt_info", u8),
                ("inter_dir", u8), ("ref_idx", i8 * 2), ("bcw_idx", u8), ("imv", u8), ("geo_split_dir", u8), ("geo_dir_ref", u8 * 2),
                ("ciip_neigh_intra", u8), ("lfnst_intra_mode", u8), ("pad0", u8 * 2),
                ("mv", i32 * 2 * 3 * 2), ("geo_mv", i32 * 2 * 2),
                ("first_tu", u32), ("num_tu", u32), ("dmvr_off", u32), ("pad1", u32)]


class Tu(C.Structure):
    _fields_ = [("x", u16), ("y", u16), ("w", u8), ("h", u8), ("comp_mask", u8), ("cbf", u8), ("joint_cbcr", u8),
                ("mts_idx", u8 * 3), ("max_scan_x", u8 * 3), ("max_scan_y", u8 * 3), ("qp", i8 * 3), ("tr_type", u8 * 3), ("pad0", u8),
                ("coef_off", u32 * 3), ("cu", u32)]


class Motion(C.Structure):
    _fields_ = [("mv", i32 * 2 * 2), ("ref_idx", i8 * 2), ("pad", u8 * 2)]


class Lfp(C.Structure):
    _fields_ = [("qp", i8 * 3), ("bs", u8), ("side_max_filt_length", u8), ("flags", u8), ("pad", u8 * 2)]


class SaoCtu(C.Structure):
    _fields_ = [("mode", u8 * 3), ("type", u8 * 3), ("band_pos", u8 * 3), ("offset", i8 * 4 * 3), ("pad", u8 * 3)]


class AlfCtu(C.Structure):
    _fields_ = [("cc_idc", u8 * 2), ("enable", u8 * 3), ("alt", u8 * 2), ("pad", u8), ("luma_filter_idx", i16), ("pad2", u8 * 2)]


class WpEntry(C.Structure):
    _fields_ = [("weight", i16), ("offset", i16), ("present", u8), ("pad", u8 * 3)]


class WpParams(C.Structure):
    _fields_ = [("log2_denom", u8 * 2), ("pad", u8 * 6), ("e", WpEntry * 3 * VVR_MAX_REFS * 2)]


class ScalingList(C.Structure):
    _fields_ = [("coef", u8 * 64 * 28), ("dc", u8 * 28), ("pad", u8 * 4)]


class Subpic(C.Structure):
    _fields_ = [("x0", u16), ("y0", u16), ("x1", u16), ("y1", u16), ("treated_as_pic", u8), ("lf_across", u8), ("pad", u8 * 2)]


class SliceHeader(C.Structure):      # vvr_slice_header: what a slice header sets for its slice only
    _fields_ = [("tool_flags", u32), ("deblock_beta_offset_div2", i8 * 3), ("deblock_tc_offset_div2", i8 * 3), ("slice_type", u8), ("alf_set", u8), ("wp_set", u8), ("pad", u8 * 3)]


class RprRef(C.Structure):           # vvr_rpr_ref: one reference picture as the current picture sees it (reference picture resampling)
    _fields_ = [("ratio", i32 * 2), ("win_left", i32), ("win_top", i32), ("width", u16), ("height", u16), ("scaled", u8), ("hor_collocated_chroma", u8), ("ver_collocated_chroma", u8), ("pad", u8)]


class RprParams(C.Structure):
    _fields_ = [("win_left", i32), ("win_top", i32), ("ref", RprRef * VVR_MAX_REFS * 2)]


class Picture(C.Structure):
    _fields_ = [("hdr", PicHeader), ("num_cu", u32), ("num_tu", u32),
                ("cu", C.POINTER(Cu)), ("tu", C.POINTER(Tu)), ("ctu_first_cu", C.POINTER(u32)),
                ("coef", C.POINTER(i16)), ("num_coef", u64),
                ("motion", C.POINTER(Motion)), ("lfp", C.POINTER(Lfp) * 2),
                ("sao", C.POINTER(SaoCtu)), ("alf", C.POINTER(AlfCtu)),
                ("alf_params", C.POINTER(AlfParams)), ("lmcs", C.POINTER(LmcsParams)),
                ("wp", C.POINTER(WpParams)), ("scaling", C.POINTER(ScalingList)),
                ("ctu_slice", C.POINTER(u16)), ("ctu_tile", C.POINTER(u16)), ("subpics", C.c_void_p), ("num_subpics", u32),
                ("slices", C.POINTER(SliceHeader)), ("rpr", C.POINTER(RprParams)), ("num_slices", u32), ("num_alf_sets", u32), ("num_wp_sets", u32), ("resident", C.c_int)]


class Config(C.Structure):
    _fields_ = [("abi_version", u32), ("device", i32), ("max_width", u16), ("max_height", u16),
                ("chroma_format", u8), ("bit_depth", u8), ("log2_ctu", u8), ("num_slots", u8), ("num_streams", u8), ("host_threads", u8), ("stop_after", u8), ("ring_entries", u8),
                ("read_buffers", u8), ("pad", u8 * 7), ("ext_planes", C.c_void_p)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", u64), ("total_ms", C.c_double), ("algo_bytes", C.c_double)]
