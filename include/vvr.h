/*
 * vvr.h — C ABI of the MI355X-native VVC reconstruction back-end ("vvr" = VVC reconstruction).
 *
 * This is the drop-in boundary for the reconstruction stage of VVdeC.  It replaces the inner seam
 *     DecLibRecon::create / destroy / decompressPicture / waitForPrevDecompressedPic / getCurrPic
 *     (reference: source/Lib/DecoderLib/DecLibRecon.h:143-200, called only from DecLib::reconPicture and
 *      DecLib::blockAndFinishPictures, source/Lib/DecoderLib/DecLib.cpp:612-655)
 * with plain-C entry points: plain pointers and sizes, status codes instead of exceptions, no C++/torch types.
 *
 * Contract on entry of vvr_submit (mirrors the contract on entry of decompressPicture):
 *   - the host parser has produced, for one picture, the per-CTU / per-CU / per-TU mode records, the quantised
 *     coefficient levels (reference: written by CABACReader.cpp:2457-2478 into the reco plane; here: a packed
 *     int16 stream, only the [0..maxScanPosX] x [0..maxScanPosY] corner of every coded transform block),
 *     the final motion field after MV derivation ("MIDER", DecCu.cpp:62/720 — stays on the host), and the
 *     deblocking edge parameters (LoopFilter::calcFilterStrengthsCTU, LoopFilter.cpp:360/495 — stays on the host);
 *   - reference pictures are identified by DPB slot numbers owned by this context.
 * Contract on exit of vvr_wait: the three planes of the output slot hold the final (post deblock/SAO/ALF) samples,
 *   identical to the reference decoder's output for the same records (VVC is an integer specification).
 *
 * All structs are little-endian PODs with fixed layout; arrays are struct-of-arrays per picture.
 * Coordinates are in LUMA samples unless a field says otherwise.  Only 4:2:0 and 4:0:0 are accepted in this version.
 */
#ifndef VVR_H
#define VVR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define VVR_API __attribute__((visibility("default")))
#else
#define VVR_API
#endif

#define VVR_ABI_VERSION 5

/* ------------------------------------------------------------------------------------------------------------------
 * status codes (negative = error; mirrors the style of vvdecErrorCodes, include/vvdec/vvdec.h.in:91-105)
 * ---------------------------------------------------------------------------------------------------------------- */
enum {
  VVR_OK               = 0,
  VVR_ERR_UNSPECIFIED  = -1,
  VVR_ERR_PARAMETER    = -2,   /* inconsistent picture description                                  */
  VVR_ERR_UNSUPPORTED  = -3,   /* a coding tool / format this build does not reconstruct            */
  VVR_ERR_DEVICE       = -4,   /* HIP runtime error (message via vvr_last_error)                    */
  VVR_ERR_NO_DEVICE    = -5,   /* no usable gfx950 device: the back-end never falls back to the CPU */
  VVR_NOT_READY        = 1,    /* (non-blocking queries) the pictures concerned have not been handed to the device yet: ask again */
  VVR_ERR_BUSY         = -6,   /* DPB slot still in use / too many pictures in flight               */
};

/* ------------------------------------------------------------------------------------------------------------------
 * picture-level description
 * ---------------------------------------------------------------------------------------------------------------- */
#define VVR_MAX_REFS      16
#define VVR_MAX_ALF_APS    8
#define VVR_ALF_CLASSES   25
#define VVR_ALF_LUMA_TAPS 13   /* 12 symmetric taps + centre (centre entry unused)   */
#define VVR_ALF_CHR_TAPS   7   /*  6 symmetric taps + centre                         */
#define VVR_ALF_MAX_CHR_ALT 8
#define VVR_CCALF_FILTERS  4
#define VVR_CCALF_TAPS     7   /* MAX_NUM_CC_ALF_CHROMA_COEFF - 1 signalled taps + pad */

/* tool_flags */
enum {
  VVR_TOOL_SAO_LUMA     = 1u << 0,   /* slice_sao_luma_flag                                        */
  VVR_TOOL_SAO_CHROMA   = 1u << 1,
  VVR_TOOL_ALF          = 1u << 2,   /* sps ALF on and at least one component enabled in the slice */
  VVR_TOOL_CCALF        = 1u << 3,
  VVR_TOOL_LMCS         = 1u << 4,   /* slice LMCS enabled (luma mapping)                          */
  VVR_TOOL_LMCS_CSCALE  = 1u << 5,   /* chroma residual scaling                                    */
  VVR_TOOL_DEBLOCK_OFF  = 1u << 6,   /* deblocking disabled for the picture                        */
  VVR_TOOL_DEP_QUANT    = 1u << 7,
  VVR_TOOL_BDOF         = 1u << 8,   /* sps BDOF on and not disabled in the picture header         */
  VVR_TOOL_DMVR         = 1u << 9,
  VVR_TOOL_PROF         = 1u << 10,
  VVR_TOOL_JCCR_SIGN    = 1u << 11,  /* ph_joint_cbcr_sign_flag                                    */
  VVR_TOOL_STILL_REF    = 1u << 12,  /* picture is still referenced: DMVR refined MVs are returned */
  VVR_TOOL_LFNST        = 1u << 13,  /* sps LFNST on (TrQuant.cpp:301)                              */
  VVR_TOOL_MTS          = 1u << 14,  /* sps MTS on (explicit+implicit selection resolved per TU)    */
  VVR_TOOL_CCLM_COLLOC  = 1u << 15,  /* sps_chroma_vertical_collocated_flag: CCLM down-samples luma with the 5-tap cross filter */
  VVR_TOOL_WP           = 1u << 16,  /* explicit weighted prediction applies to this picture: P slice with pps_weighted_pred_flag or
                                        B slice with pps_weighted_bipred_flag (InterPrediction.cpp:707,735-742); vvr_picture.wp set */
  VVR_TOOL_SCALING_LIST = 1u << 17,  /* explicit scaling list in use for the slice (Quant.cpp:330-336); vvr_picture.scaling set */
  VVR_TOOL_SCALING_LIST_NO_LFNST = 1u << 18,  /* sps_scaling_matrix_for_lfnst_disabled_flag */
  VVR_TOOL_IMPLICIT_MTS = 1u << 19,  /* MTS on without sps_explicit_mts_intra_enabled_flag: intra luma blocks use the implicit DST-7 rule.
                                        Informative (vvr_tu.tr_type is already resolved); a checker that drives the reference decoder needs it */
  VVR_TOOL_IBC          = 1u << 20,  /* sps_ibc_enabled_flag: the picture may hold VVR_PRED_IBC CUs (InterPrediction::xIntraBlockCopy,
                                        InterPrediction.cpp:1995).  Block vector in vvr_cu.mv[0][0] (1/16 units, integer sample
                                        positions); it must point at samples of the same CTU row that precede the CU in decoding order
                                        and still sit in the IBC virtual buffer (CodingStructure::fillIBCbuffer, CodingStructure.cpp:550) */
  VVR_TOOL_LADF         = 1u << 21,  /* sps_ladf_enabled_flag: luma-adaptive deblocking, parameters in vvr_pic_header.ladf_* (LoopFilter.cpp:1363,1519).
                                        Informative like VVR_TOOL_IMPLICIT_MTS: the back-end looks at ladf_num_intervals                       */
  VVR_TOOL_NO_LF_ACROSS_SLICES = 1u << 22,  /* !pps_loop_filter_across_slices_enabled_flag: SAO and ALF do not look across slice boundaries (the deblocking
                                        edges there are already switched off in the edge-parameter table the host derives)                     */
  VVR_TOOL_NO_LF_ACROSS_TILES  = 1u << 23,  /* !pps_loop_filter_across_tiles_enabled_flag, likewise for tile boundaries                              */
  VVR_TOOL_AFFINE_MV_ON_DEVICE = 1u << 24,  /* the sub-block MVs of affine CUs are spanned by the back-end from the CU's control-point MVs (cu.mv[list][0..2];
                                               PU::setAllAffineMv, UnitTools.cpp:2689): vvr_picture.motion is not read for affine CUs and need not hold them   */
  VVR_TOOL_COL_MOTION   = 1u << 25,  /* the back-end keeps the picture's collocated motion (the TMVP storage of later pictures): vvr_picture.motion at every
                                        second 4x4 unit in both directions, with the MVs of DMVR CUs replaced by the refined ones as soon as the DMVR
                                        kernel has them (DecCu::TaskFinishMotionInfo, DecCu.cpp:161-253).  vvr_read_col_motion() hands it out; the host
                                        neither reads the delta MVs nor patches / subsamples its motion field                                      */
  VVR_TOOL_LFP_ON_DEVICE = 1u << 26, /* the back-end derives the deblocking edge parameters itself (the reference's LF_INIT task, LoopFilter::
                                        calcFilterStrengthsCTU, LoopFilter.cpp:495-1360) from the CU / TU records: vvr_picture.lfp is not read and may be NULL.
                                        vvr_picture.motion then has to hold the cells of SbTMVP and GPM CUs - and of affine CUs unless
                                        VVR_TOOL_AFFINE_MV_ON_DEVICE is set - (the motion of every other CU is in its record); a slice that switches
                                        deblocking off in a picture that deblocks says so with VVR_TOOL_DEBLOCK_OFF in its vvr_slice_header.tool_flags       */
};

typedef struct vvr_alf_params {     /* final filters, AdaptiveLoopFilter::reconstructCoeff (AdaptiveLoopFilter.cpp:888) stays on the host */
  int16_t luma_coeff[VVR_MAX_ALF_APS][VVR_ALF_CLASSES][VVR_ALF_LUMA_TAPS];   /* un-transposed, per class                 */
  int16_t luma_clip [VVR_MAX_ALF_APS][VVR_ALF_CLASSES][VVR_ALF_LUMA_TAPS];   /* clipping VALUES (m_alfClippVls resolved)  */
  int16_t chroma_coeff[VVR_ALF_MAX_CHR_ALT][VVR_ALF_CHR_TAPS];               /* per alternative (shared by Cb and Cr)     */
  int16_t chroma_clip [VVR_ALF_MAX_CHR_ALT][VVR_ALF_CHR_TAPS];
  int16_t ccalf_coeff[2][VVR_CCALF_FILTERS][VVR_CCALF_TAPS + 1];
  uint8_t num_luma_aps;             /* alfCtbFilterIndex >= 16 selects luma_coeff[idx-16]                                  */
  uint8_t pad[7];
} vvr_alf_params;

typedef struct vvr_lmcs_params {    /* Reshape::constructReshaper (Reshape.cpp:318) stays on the host */
  int16_t fwd_lut[1024 * 4];        /* forward map of every sample value (rspFwdCore, Buffer.cpp:321), 1 << bit_depth entries used */
  int16_t inv_lut[1024 * 4];        /* inverse map (m_invLUT)                                                                      */
  int16_t chroma_scale[16];         /* m_chromaAdjHelpLUT                                                                          */
  int16_t pivot[17];                /* m_reshapePivot                                                                              */
  int16_t min_bin, max_bin;         /* lmcs_min_bin_idx, LmcsMaxBinIdx (Reshape::getPWLIdxInv, :280)                               */
  /* the syntax-level model the tables were built from (lmcs_data()): the back-end does not read it; it lets a checker that
   * drives the reference decoder's own Reshape class rebuild the same tables */
  int16_t model_delta_cw[16];       /* lmcsDeltaCW[i] (signed)  */
  int16_t model_delta_crs;          /* lmcsDeltaCrs             */
  int16_t pad[4];
} vvr_lmcs_params;

typedef struct vvr_wp_entry {        /* WPScalingParam (Slice.h:2215) of one reference picture and component, as parsed */
  int16_t weight;                    /* iWeight ( 1 << log2_denom when the flag is off )                       */
  int16_t offset;                    /* iOffset, in 8-bit units (scaled by 1 << (bit_depth - 8) when applied)  */
  uint8_t present;                   /* bPresentFlag (luma_weight_lX_flag / chroma_weight_lX_flag)             */
  uint8_t pad[3];
} vvr_wp_entry;

typedef struct vvr_wp_params {       /* pred_weight_table(): Slice::m_weightPredTable (Slice.h:2560) */
  uint8_t      log2_denom[2];        /* uiLog2WeightDenom of luma / chroma                           */
  uint8_t      pad[6];
  vvr_wp_entry e[2][VVR_MAX_REFS][3];/* [list][refIdx][Y, Cb, Cr]                                    */
} vvr_wp_params;

typedef struct vvr_scaling_list {    /* ScalingList after scaling_list_data() decoding (prediction / DPCM resolved) */
  uint8_t coef[28][64];              /* m_scalingListCoef[id]: ids 0-1 are 2x2 (4 entries), 2-7 4x4 (16), 8-27 8x8 (64), raster order */
  uint8_t dc[28];                    /* m_scalingListDC[id] (ids >= 14: the DC of the up-sampled 16x16 .. 64x64 matrices) */
  uint8_t pad[4];
} vvr_scaling_list;

typedef struct vvr_pic_header {
  uint32_t abi_version;             /* VVR_ABI_VERSION                                                  */
  uint32_t tool_flags;              /* VVR_TOOL_*                                                       */
  uint16_t width, height;           /* luma samples                                                     */
  uint8_t  chroma_format;           /* 0 = 4:0:0, 1 = 4:2:0                                             */
  uint8_t  bit_depth;               /* 8..10 (Main 10)                                                  */
  uint8_t  log2_ctu;                /* 5..7                                                             */
  uint8_t  slice_type;              /* 0 B, 1 P, 2 I  (SliceType, CommonDef.h)                          */
  int32_t  poc;
  int16_t  out_slot;                /* DPB slot that receives the reconstruction                        */
  int8_t   num_ref[2];
  int16_t  ref_slot[2][VVR_MAX_REFS];
  int32_t  ref_poc [2][VVR_MAX_REFS];
  int8_t   deblock_beta_offset_div2[3];   /* Y, Cb, Cr (LoopFilter.cpp:1473,1637)                       */
  int8_t   deblock_tc_offset_div2[3];
  uint8_t  log2_sao_offset_scale[2];      /* luma, chroma                                               */
  int8_t   min_qp_ts;               /* 4 + 6*internalMinusInputBitDepth (Quant.cpp:104)                  */
  uint8_t  ladf_num_intervals;      /* 0: LADF off; else sps_num_ladf_intervals_minus2 + 2 (2..5), LoopFilter::deriveLADFShift (LoopFilter.cpp:1363) */
  int8_t   ladf_qp_offset[5];       /* SPS::getLadfQpOffset(k): [0] = sps_ladf_lowest_interval_qp_offset   */
  uint8_t  pad;
  int16_t  ladf_lower_bound[5];     /* SPS::getLadfIntervalLowerBound(k), luma level; [0] unused           */
  /* virtual boundaries of the picture header (ph_virtual_boundaries_present_flag; PicHeader::getVirtualBoundariesPosX / PosY, Slice.h): luma
   * positions, multiples of 8, inside the picture, ascending.  The in-loop filters do not work across them: edges on a boundary are not
   * deblocked (that is part of the edge tables the host supplies, LoopFilter.cpp:669-690), SAO leaves the two sample columns / rows at a
   * boundary alone for the edge classes that look across it (SampleAdaptiveOffset.cpp:823), ALF filters every part of a CTU the boundaries cut
   * out with its own replicated border (AdaptiveLoopFilter.cpp:142-175,764-850).                                                              */
  uint8_t  num_ver_vb, num_hor_vb;  /* 0..3 each                                                          */
  uint16_t wrap_offset;             /* 0: off; else pps_ref_wraparound_enabled_flag with PPS::getWrapAroundOffset() luma samples: motion compensation reads
                                       a reference picture as if it wrapped around horizontally at that period (360-degree video; wrapClipMv, Mv.cpp:112,
                                       Picture::extendPicBorderWrap, Picture.cpp:410).  Multiple of 8, CTU size + 16 .. picture width                        */
  uint8_t  pad2[2];
  uint16_t vb_pos_x[3], vb_pos_y[3];
  uint8_t  pad3[4];
} vvr_pic_header;

/* ------------------------------------------------------------------------------------------------------------------
 * coding unit  (source of each field: CodingUnit, source/Lib/CommonLib/Unit.h:314-420)
 * ---------------------------------------------------------------------------------------------------------------- */
enum { VVR_PRED_INTER = 0, VVR_PRED_INTRA = 1, VVR_PRED_IBC = 2 };
enum { VVR_TREE_JOINT = 0, VVR_TREE_LUMA = 1, VVR_TREE_CHROMA = 2 };   /* which components the CU carries */

/* cu.flags */
enum {
  VVR_CU_ROOT_CBF   = 1u << 0,
  VVR_CU_SKIP       = 1u << 1,
  VVR_CU_MERGE      = 1u << 2,
  VVR_CU_AFFINE     = 1u << 3,
  VVR_CU_AFFINE_6P  = 1u << 4,
  VVR_CU_CIIP       = 1u << 5,
  VVR_CU_GEO        = 1u << 6,
  VVR_CU_SBTMVP     = 1u << 7,   /* mergeType == MRG_TYPE_SUBPU_ATMVP                                       */
  VVR_CU_MIP        = 1u << 8,
  VVR_CU_MIP_TRANSP = 1u << 9,
  VVR_CU_SMVD       = 1u << 10,
  VVR_CU_MMVD       = 1u << 11,
};

/* cu.mc_mode: the branch InterPrediction::motionCompensation (InterPrediction.cpp:1372-1459) takes for this CU.
 * Resolving it needs POCs/flags only and is done once per CU by the host glue (integration/vvr_extract.h::resolveMcMode). */
enum {
  VVR_MC_NONE = 0,
  VVR_MC_UNI,          /* one list, or identical-motion shortcut (xCheckIdenticalMotion, :404)          */
  VVR_MC_BI,           /* xPredInterBi: two lists + addAvg / BCW                                        */
  VVR_MC_BDOF,         /* xSubPuBio (:551)                                                              */
  VVR_MC_DMVR,         /* xProcessDMVR (:1847), BDOF decided per sub-block                              */
  VVR_MC_DMVR_BDOF,
  VVR_MC_AFFINE,       /* xPredAffineBlk (:934) (+PROF)                                                 */
  VVR_MC_SBTMVP,       /* xSubPuMC (:438)                                                               */
  VVR_MC_GEO,          /* motionCompensationGeo (:1461)                                                 */
};

typedef struct vvr_cu {
  uint16_t x, y;                 /* luma position                                                            */
  uint8_t  w, h;                 /* luma size (chroma-tree CU: luma-equivalent area)                         */
  uint8_t  tree;                 /* VVR_TREE_*                                                               */
  uint8_t  pred_mode;            /* VVR_PRED_*                                                               */
  uint16_t flags;                /* VVR_CU_*                                                                 */
  int8_t   qp;                   /* cu.qp (luma QP without QpBdOffset)                                       */
  uint8_t  mc_mode;              /* VVR_MC_*                                                                 */
  /* intra */
  uint8_t  intra_dir[2];         /* FINAL modes (PU::getFinalIntraMode, UnitTools.cpp:587): 0 planar, 1 DC, 2..66, 67..69 LM/MDLM_L/MDLM_T; MIP: mode id */
  uint8_t  multi_ref_idx;        /* 0..2                                                                     */
  uint8_t  isp_mode;             /* 0 none, 1 HOR_INTRA_SUBPARTITIONS, 2 VER                                  */
  uint8_t  bdpcm[2];             /* luma, chroma: 0 off, 1 hor, 2 ver                                        */
  uint8_t  lfnst_idx;            /* 0..2                                                                     */
  uint8_t  sbt_info;             /* CodingUnit::_sbtInfo                                                     */
  /* inter */
  uint8_t  inter_dir;            /* 1 L0, 2 L1, 3 bi                                                         */
  int8_t   ref_idx[2];           /* -1 = unused                                                              */
  uint8_t  bcw_idx;              /* 0..4, BCW_DEFAULT = 2                                                    */
  uint8_t  imv;                  /* 3 = IMV_HPEL selects the alternative half-pel filter                     */
  uint8_t  geo_split_dir;
  uint8_t  geo_dir_ref[2];       /* interDirrefIdxGeo0/1: [i] = (interDir << 4) | refIdx, interDir 1 = L0, 2 = L1 */
  uint8_t  ciip_neigh_intra;     /* bit0: above neighbour intra, bit1: left neighbour intra (IntraPrediction.cpp:917-927) */
  uint8_t  lfnst_intra_mode;     /* intra mode used for LFNST set selection before wide-angle remap (TrQuant.cpp:213-221) */
  uint8_t  pad0[2];
  int32_t  mv[2][3][2];          /* [list][cpmv idx][hor,ver], 1/16 luma sample; [l][0] for translational     */
  int32_t  geo_mv[2][2];         /* GPM: the two uni-prediction MVs                                           */
  uint32_t first_tu, num_tu;     /* range in the TU array (decode order)                                     */
  uint32_t dmvr_off;             /* first entry of this CU in the DMVR delta-MV output array                 */
  uint32_t pad1;
} vvr_cu;

/* ------------------------------------------------------------------------------------------------------------------
 * transform unit  (TransformUnit, Unit.h:285-304)
 * ---------------------------------------------------------------------------------------------------------------- */
enum { VVR_MTS_DCT2 = 0, VVR_MTS_SKIP = 1, VVR_MTS_DST7_DST7 = 2, VVR_MTS_DCT8_DST7 = 3, VVR_MTS_DST7_DCT8 = 4, VVR_MTS_DCT8_DCT8 = 5 };

typedef struct vvr_tu {
  uint16_t x, y;                 /* luma position                                                            */
  uint8_t  w, h;                 /* luma size (chroma block = w/2 x h/2 for 4:2:0)                            */
  uint8_t  comp_mask;            /* bit c set: block of component c exists in this TU (dual tree / ISP)       */
  uint8_t  cbf;                  /* bit c: coded block flag of component c                                    */
  uint8_t  joint_cbcr;           /* 0 or jointCbCr mask (1,2,3)                                               */
  uint8_t  mts_idx[3];           /* VVR_MTS_* per component                                                   */
  uint8_t  max_scan_x[3];        /* last significant column per component (TransformUnit::maxScanPosX)       */
  uint8_t  max_scan_y[3];
  int8_t   qp[3];                /* QpParam::Qps[0] per component incl. QpBdOffset (Quant.cpp:65-101);        */
                                 /* for a joint-CbCr TU qp[1]/qp[2] hold the joint QP where ICT mode == 2     */
  uint8_t  tr_type[3];           /* (trTypeVer << 2) | trTypeHor with 0 DCT2, 1 DCT8, 2 DST7                   */
                                 /* = TrQuant::getTrTypes (TrQuant.cpp:330-407), resolved by the host glue     */
  uint8_t  pad0;
  uint32_t coef_off[3];          /* offset (int16 units) of this block's level corner in the coefficient stream */
  uint32_t cu;                   /* owning CU                                                                  */
} vvr_tu;

/* ------------------------------------------------------------------------------------------------------------------
 * per-4x4 side tables (picture raster order, stride = ceil(width/4))
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct vvr_motion {      /* MotionInfo, MotionInfo.h:122 */
  int32_t mv[2][2];              /* [list][hor,ver]                                                            */
  int8_t  ref_idx[2];            /* -1 unused (MI_NOT_VALID for intra; IBC: both -1, mv[0] = block vector, UnitTools.cpp:3018) */
  uint8_t pad[2];
} vvr_motion;

typedef struct vvr_lfp {         /* LoopFilterParam, TypeDef.h:694-707; semantics SURVEY.md Appendix D         */
  int8_t  qp[3];
  uint8_t bs;                    /* 2 bits per component                                                       */
  uint8_t side_max_filt_length;  /* [6:4] P, [2:0] Q, bit 7 transform edge                                     */
  uint8_t flags;                 /* bit0 filterEdge luma, bit1 filterEdge chroma, bit5 chroma large block      */
  uint8_t pad[2];
} vvr_lfp;

/* ------------------------------------------------------------------------------------------------------------------
 * per-CTU loop filter controls
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct vvr_sao_ctu {     /* SAOBlkParam after reconstructBlkSAOParam (merge resolved, offsets scaled)   */
  uint8_t mode[3];               /* 0 off, 1 on                                                                */
  uint8_t type[3];               /* 0 EO_0, 1 EO_90, 2 EO_135, 3 EO_45, 4 BO                                    */
  uint8_t band_pos[3];           /* BO: first band                                                             */
  int8_t  offset[3][4];          /* EO: classes {valley, half-valley, half-peak, peak}; BO: 4 consecutive bands */
  uint8_t pad[3];
} vvr_sao_ctu;

typedef struct vvr_alf_ctu {     /* CtuAlfData, CodingStructure.h:75 */
  uint8_t cc_idc[2];             /* 0 off, else filter index + 1                                               */
  uint8_t enable[3];
  uint8_t alt[2];                /* chroma alternative                                                         */
  uint8_t pad;
  int16_t luma_filter_idx;       /* < 16: fixed set, else APS idx - 16                                         */
  uint8_t pad2[2];
} vvr_alf_ctu;

typedef struct vvr_subpic {      /* SubPic (Slice.h:820): one sub-picture of the layout the SPS signals (sps_subpic_info_present_flag)        */
  uint16_t x0, y0, x1, y1;       /* luma samples, inclusive: getSubPicLeft / Top / Right / Bottom; x0, y0 on the CTU grid                      */
  uint8_t  treated_as_pic;       /* sps_subpic_treated_as_pic_flag: motion compensation of its CUs reads nothing outside the sub-picture       */
                                 /* (clipMvInSubpic, Mv.cpp:84; Picture::getSubPicBuf + DecLibRecon::createSubPicRefBufs, DecLibRecon.cpp:388)   */
  uint8_t  lf_across;            /* sps_loop_filter_across_subpic_enabled_flag: 0 = SAO and ALF of its CTUs do not look into other sub-pictures */
  uint8_t  pad[2];
} vvr_subpic;

/* ------------------------------------------------------------------------------------------------------------------
 * one picture to reconstruct
 * ---------------------------------------------------------------------------------------------------------------- */
/* What a slice header sets for its slice only (DecLibRecon switches on ctuData.slice per CTU, DecLibRecon.cpp:787-790,856-860; every stage reads
 * cu.slice / the CTU's slice: Quant.cpp:306,336, DecCu.cpp:383,460,489, LoopFilter.cpp:423,1473,1637, Reshape.cpp:385, AdaptiveLoopFilter.cpp:515,558,603).
 * NOT here, by construction of the description: the slice's reference picture lists - hdr.ref_slot / ref_poc hold the UNION of the slices' lists
 * per list (at most VVR_MAX_REFS pictures) and every cu.ref_idx / vvr_motion.ref_idx indexes that union (whoever flattens the picture renumbers
 * them; an I slice simply has no inter CUs) - and SAO / ALF on-off switches, which are resolved into the per-CTU records.                       */
#define VVR_SLICE_TOOL_MASK ( VVR_TOOL_DEP_QUANT | VVR_TOOL_LMCS | VVR_TOOL_LMCS_CSCALE | VVR_TOOL_SCALING_LIST | VVR_TOOL_WP )
typedef struct vvr_slice_header {
  uint32_t tool_flags;               /* this slice's value of the switches a slice header carries: VVR_TOOL_DEP_QUANT (sh_dep_quant_used_flag), VVR_TOOL_LMCS
                                        (sh_lmcs_used_flag), VVR_TOOL_LMCS_CSCALE (the picture's flag and sh_lmcs_used_flag), VVR_TOOL_SCALING_LIST
                                        (sh_explicit_scaling_list_used_flag), VVR_TOOL_WP (a P / B slice with weights); all other bits are ignored: those
                                        tools follow hdr.tool_flags                                                                                       */
  int8_t   deblock_beta_offset_div2[3];   /* Y, Cb, Cr: of the slice the deblocked CTU belongs to (LoopFilter.cpp:421,1473,1637)                        */
  int8_t   deblock_tc_offset_div2[3];
  uint8_t  slice_type;               /* 0 B, 1 P, 2 I (informative: the CUs say how they are predicted)                                                */
  uint8_t  alf_set;                  /* which of vvr_picture.alf_params[] holds the filters of the APSs this slice refers to (luma list, chroma, CC-ALF)   */
  uint8_t  wp_set;                   /* which of vvr_picture.wp[] holds this slice's pred_weight_table()                                               */
  uint8_t  pad[3];
} vvr_slice_header;

/* Reference picture resampling (RPR; sps_ref_pic_resampling_enabled_flag, scaling windows of the PPSs): how the current picture sees each of its
 * reference pictures.  A reference picture is "scaled" when its size or its scaling window differs from the current picture's (Picture::isRefScaled,
 * Picture.h:265); a prediction from such a picture is interpolated at positions that advance by the scaling ratio per sample, with low-pass filter
 * sets above ratios of 1.25 and 1.75 (InterPrediction::xPredInterBlkRPR, InterPrediction.cpp:2081-2217), its motion vectors are not clipped, and the
 * CU takes no BDOF, DMVR or PROF (InterPrediction.cpp:1431-1435,1029).  Indexed like hdr.ref_slot (the union of the slices' lists).  The DPB slot of a
 * scaled reference picture holds a picture of `width` x `height` luma samples in its top left corner (vvr_config.max_width / max_height bound every
 * picture of a context; a picture is reconstructed at hdr.width x hdr.height).
 * SbTMVP: every 8x8 sub-block is predicted on its own from the sub-block's position.  (The reference joins sub-blocks of equal motion unless the
 * slice's first reference pictures are scaled, InterPrediction.cpp:477; sub-block motion refers to the first reference picture of each list, so a
 * sub-block that reads a scaled picture is never part of a joined block.)
 * Not combined with reference wrap-around or with sub-pictures treated as pictures (the reference keeps no wrap copy of a scaled picture,
 * Picture.h:278, and a coded video sequence with such sub-pictures does not change its picture size).                                            */
typedef struct vvr_rpr_ref {
  int32_t  ratio[2];               /* Slice::getScalingRatio (CU::getRprScaling, UnitTools.cpp:92): x, y; 1 << 14 = same scale; 1 << 11 .. 1 << 15        */
  int32_t  win_left, win_top;      /* scaling window of the reference picture's PPS: left / top offset in luma samples (offset * SPS::getWinUnitX / Y)    */
  uint16_t width, height;          /* luma size of the reference picture                                                                                  */
  uint8_t  scaled;                 /* Picture::isRefScaled( current PPS )                                                                                 */
  uint8_t  hor_collocated_chroma;  /* sps_chroma_horizontal_collocated_flag / ..vertical.. of the reference picture's SPS (InterPrediction.cpp:2126)      */
  uint8_t  ver_collocated_chroma;
  uint8_t  pad;
} vvr_rpr_ref;
typedef struct vvr_rpr_params {
  int32_t     win_left, win_top;   /* scaling window of the current picture's PPS, luma samples                                                           */
  vvr_rpr_ref ref[2][VVR_MAX_REFS];
} vvr_rpr_params;

typedef struct vvr_picture {
  vvr_pic_header        hdr;
  uint32_t              num_cu, num_tu;
  const vvr_cu*         cu;            /* decode order (CTU raster, z-scan inside the CTU)                      */
  const vvr_tu*         tu;
  const uint32_t*       ctu_first_cu;  /* [num_ctu + 1] first CU of every CTU                                    */
  const int16_t*        coef;          /* packed quantised levels                                               */
  uint64_t              num_coef;
  const vvr_motion*     motion;        /* [h4][w4], may be NULL for intra pictures                               */
  const vvr_lfp*        lfp[2];        /* [EDGE_VER, EDGE_HOR][h4][w4]; not read with VVR_TOOL_LFP_ON_DEVICE        */
  const vvr_sao_ctu*    sao;           /* [num_ctu] or NULL                                                      */
  const vvr_alf_ctu*    alf;           /* [num_ctu] or NULL                                                      */
  const vvr_alf_params* alf_params;    /* NULL when ALF is off; [num_alf_sets] tables when slices refer to different APSs */
  const vvr_lmcs_params* lmcs;         /* NULL when LMCS is off                                                  */
  const vvr_wp_params*  wp;            /* NULL unless VVR_TOOL_WP; [num_wp_sets] tables when slices carry different weights */
  const vvr_scaling_list* scaling;     /* NULL unless VVR_TOOL_SCALING_LIST                                      */
  /* Slices and tiles.  The CTUs of the description are listed in picture raster order whatever order the bit stream coded them in.  Across a
   * slice or tile boundary nothing is available to intra prediction, CCLM or the LMCS chroma-scaling neighbourhood (CodingStructure::
   * getCURestricted, CodingStructure.cpp:464), and SAO / ALF stop there when the VVR_TOOL_NO_LF_ACROSS_* flag of the kind of boundary is set
   * (SampleAdaptiveOffset.cpp:741-830, AdaptiveLoopFilter.cpp:118-200).
   * Slices with headers of their own (ABI 4): `slices[ctu_slice[ctu]]` carries what a slice header can set differently from its neighbours -
   * see vvr_slice_header.  slices == NULL: every slice takes the values of `hdr` (and alf_params / wp are single tables). */
  const uint16_t*       ctu_slice;     /* [num_ctu] slice index of every CTU, NULL = one slice                   */
  const uint16_t*       ctu_tile;      /* [num_ctu] tile index of every CTU, NULL = one tile                     */
  /* Sub-pictures: rectangles of whole CTUs that tile the picture (NULL / 0 or 1: the picture is its only sub-picture).  Not combined with
   * reference wrap-around (the reference does not support the pair either, Picture.h:114).                                                   */
  const vvr_subpic*     subpics;
  uint32_t              num_subpics;
  const vvr_slice_header* slices;      /* [num_slices] or NULL                                                   */
  const vvr_rpr_params* rpr;           /* NULL: no reference picture of this picture is scaled (ABI 5)           */
  uint32_t              num_slices;    /* (ctu_slice values are < num_slices when slices != NULL)                */
  uint32_t              num_alf_sets;  /* entries of alf_params[] (0 or 1: one table), selected by vvr_slice_header.alf_set */
  uint32_t              num_wp_sets;   /* entries of wp[] (0 or 1: one table), selected by vvr_slice_header.wp_set          */
  int                   resident;      /* 0: all array pointers are host memory (copied H2D by vvr_submit);      */
                                       /* 1: all array pointers are DEVICE memory already resident in HBM        */
} vvr_picture;

/* ------------------------------------------------------------------------------------------------------------------
 * context  (replaces DecLibRecon instances + the DPB picture buffers they write)
 * ---------------------------------------------------------------------------------------------------------------- */
enum { VVR_STOP_NONE = 0, VVR_STOP_RECO = 1, VVR_STOP_DEBLOCK = 2, VVR_STOP_SAO = 3 };

typedef struct vvr_config {
  uint32_t abi_version;
  int32_t  device;               /* HIP device ordinal                                                          */
  uint16_t max_width, max_height;
  uint8_t  chroma_format, bit_depth, log2_ctu;
  uint8_t  num_slots;            /* DPB slots (pictures resident in HBM)                                        */
  uint8_t  num_streams;          /* pictures reconstructing concurrently (reference: 2, DecLib.h:70)            */
  uint8_t  host_threads;         /* worker threads that build the device work lists of submitted pictures (the reference spreads the set-up of
                                    decompressPicture over its thread pool, DecLibRecon.cpp:429-682).  0: the submitting thread does it inside
                                    vvr_submit; N > 0: vvr_submit only queues the picture, N pictures are prepared concurrently and enqueued
                                    on the device in submission order                                                                        */
  uint8_t  stop_after;           /* conformance aid (the reference has per-stage CRC traces for the same purpose, LoopFilter.cpp:399-406): 0 =
                                    full reconstruction; VVR_STOP_RECO / _DEBLOCK / _SAO: the pictures of this context stop after that stage, so
                                    that they can be compared with the reference's picture at the same point                                  */
  uint8_t  ring_entries;         /* entries of the upload ring (pinned staging + HBM image of one picture each); 0: 2 * num_streams +
                                    2 * host_threads + 4, enough for the pictures in the workers' hands plus those in flight on the device   */
  uint8_t  read_buffers;         /* pinned staging buffers of vvr_read_picture (one picture each) allocated with the context; 0: the first calls that
                                    need one allocate it (pinning 30 MB takes milliseconds: a decoder that reads every picture back asks for them here) */
  uint8_t  pad[7];
  void*    ext_planes;           /* optional: caller-owned device memory for the DPB, num_slots * vvr_slot_bytes */
                                 /* (mirrors vvdec_decoder_open_with_allocator, vvdec.h.in:576)                 */
} vvr_config;

typedef struct vvr_context vvr_context;

/* DecLibRecon::create (DecLibRecon.cpp:132): HIP streams/events, constant tables, DPB planes. */
VVR_API int          vvr_create(const vvr_config* cfg, vvr_context** out);
/* DecLibRecon::destroy */
VVR_API void         vvr_destroy(vvr_context* ctx);
/* DecLibRecon::decompressPicture (DecLibRecon.cpp:429): asynchronous; returns a job id >= 0 or an error code.  Called from ONE submitting
 * thread.  Header, tables and the set of arrays are validated before the call returns (with host_threads == 0 the CU / TU records as well);
 * the arrays the description points to must stay valid and unchanged until vvr_inputs_done(job) or vvr_wait(job) has returned (with
 * host_threads == 0 they are consumed before vvr_submit returns).  Errors that only show later (with worker threads: bad CU / TU records;
 * work lists; device) are parked on the job, the way the reference parks exceptions on reconDone, and come back from vvr_wait.  Pictures
 * take effect in submission order (one that shares no DPB slot with a picture still being prepared may be enqueued on the device ahead of
 * it); every job should eventually be waited for (vvr_wait / vvr_sync). */
VVR_API int          vvr_submit(vvr_context* ctx, const vvr_picture* pic);
/* blocks until the host arrays of job `job` are no longer needed (its device work lists are built and staged in pinned memory) */
VVR_API int          vvr_inputs_done(vvr_context* ctx, int job);
/* Host memory the device reads directly (pinned), owned by the context (freed with it at the latest).  A parser that writes its records into
 * such memory (the way vvdec_decoder_open_with_allocator, vvdec.h.in:576, lets the application own the picture buffers) saves the back-end the
 * staging copy: the cu / tu / coef / lfp arrays of a submitted picture that lie in it are copied to HBM from where they are, and must then stay
 * unchanged until vvr_inputs_done(job) (which waits for that copy) or vvr_wait(job). */
VVR_API void*        vvr_host_alloc(vvr_context* ctx, size_t bytes);
VVR_API void         vvr_host_free(vvr_context* ctx, void* p);
/* DecLibRecon::waitForPrevDecompressedPic (DecLibRecon.cpp:684): blocks until job `job` is reconstructed; returns its status. */
VVR_API int          vvr_wait(vvr_context* ctx, int job);
/* the same question without waiting (the ready check of a completion task on the decoder's thread pool, ThreadPool.cpp addBarrierTask: no pool
 * thread sleeps in vvr_wait): VVR_OK - job `job` is reconstructed (vvr_wait would return at once, with this status); VVR_NOT_READY - not yet;
 * negative - it failed.  The job stays to be waited for. */
VVR_API int          vvr_test(vvr_context* ctx, int job);
/* wait for everything in flight */
VVR_API int          vvr_sync(vvr_context* ctx);
/* geometry of a DPB slot: byte size, and per-plane offset / stride (bytes) / rows */
VVR_API size_t       vvr_slot_bytes(const vvr_config* cfg);
VVR_API int          vvr_plane_layout(const vvr_context* ctx, int comp, size_t* offset, size_t* stride_bytes, int* width, int* height);
/* device address of plane `comp` of `slot` (for zero-copy consumers, RCCL broadcast of reference pictures) */
VVR_API void*        vvr_plane_ptr(vvr_context* ctx, int slot, int comp);
/* vvdecFrame-style export (vvdecimpl.cpp:957 xAddPicture): copies plane `comp` of `slot` into a host buffer of 16-bit samples */
VVR_API int          vvr_read_plane(vvr_context* ctx, int slot, int comp, uint16_t* dst, size_t dst_stride_samples);
/* output of one plane the way the reference hands frames to the application (VVDecImpl::copyComp, vvdecimpl.cpp:818-880, called with the
 * conformance window applied): the window (x, y, w, h in samples of the component) is copied to dst with dst_stride_bytes between rows;
 * bytes_per_sample 2 = 16-bit samples, 1 = the low byte of every sample (8-bit streams; "only narrowing conversions", :853).  Crop and
 * narrowing run on the device: exactly w * h * bytes_per_sample bytes cross PCIe.  Waits for all work on the slot. */
VVR_API int          vvr_read_output(vvr_context* ctx, int slot, int comp, int x, int y, int w, int h, int bytes_per_sample, void* dst, size_t dst_stride_bytes);
/* decoded picture hash of a slot, as the decoded-picture-hash SEI defines it and the reference checks it (calcMD5 / calcCRC / calcChecksum,
 * PicYuvMD5.cpp:99-221): one digest per component over the whole plane in raster order, samples as 1 byte (bit depth 8) or 2 bytes little
 * endian.  digest receives num_components x digest_len bytes (MD5 16, CRC 2, checksum 4), *digest_len the length of one.  CRC and checksum
 * are computed on the device (only per-row partial results cross PCIe); for MD5, a serial chain over the bytes of a plane, the device packs
 * the plane to exactly those bytes and the host hashes them. */
enum { VVR_HASH_MD5 = 0, VVR_HASH_CRC = 1, VVR_HASH_CHECKSUM = 2 };
VVR_API int          vvr_picture_hash(vvr_context* ctx, int slot, int method, uint8_t* digest, int* digest_len);
/* the finished picture in `slot` (every plane, at the picture's size) into the caller's buffers - what a decoder does with each picture it hands to
 * the application (the planes of a vvdecFrame live in the Picture's own buffers, vvdecimpl.cpp:1058).  Waits for NOTHING: the caller has waited for
 * the picture (vvr_wait); other pictures in flight are not held up (vvr_read_plane drains the context).  Each plane crosses PCIe in one transfer
 * into pinned memory of the context, from where `threads` (1..16) threads lay the rows out at dst[c] with dst_stride_samples[c]: a copy straight
 * into pageable memory runs at a fraction of the link's rate.  May be called by several threads at once (one staging buffer per call in flight). */
VVR_API int          vvr_read_picture(vvr_context* ctx, int slot, uint16_t* const* dst, const size_t* dst_stride_samples, int threads);
/* upload a reference picture produced elsewhere (another GPU / a test) into a slot */
VVR_API int          vvr_write_plane(vvr_context* ctx, int slot, int comp, const uint16_t* src, size_t src_stride_samples);
/* size of the picture a slot holds (luma samples; a picture lies in the top left corner of its slot): vvr_submit sets it to the size of the picture
 * it reconstructs into the slot, this call is for pictures that come from outside (vvr_write_plane) when they are smaller than the context's
 * pictures - a coded video sequence with reference picture resampling.  vvr_read_plane, vvr_write_plane and vvr_picture_hash move / cover that
 * many samples; a new context's slots have the context's size.                                                                              */
VVR_API int          vvr_slot_picture_size(vvr_context* ctx, int slot, int width, int height);
/* DMVR refined delta MVs of job (TaskFinishMotionInfo, DecCu.cpp:161): copies num_entries * 2 int32 */
VVR_API int          vvr_read_dmvr(vvr_context* ctx, int job, int32_t* dst, size_t num_entries);
/* Collocated motion of a picture submitted with VVR_TOOL_COL_MOTION (blocks until the job is done): ceil(w4 / 2) * ceil(h4 / 2) records in raster order,
 * record (x, y) = the final motion of the 4x4 unit (2x, 2y) - the layout of ColocatedMotionInfo = MotionInfo (MotionInfo.h:157), what
 * DecCu::TaskFinishMotionInfo leaves in CtuData::colMotion.  Returns the number of records of the picture (copies at most num_entries), < 0 on error. */
VVR_API int          vvr_read_col_motion(vvr_context* ctx, int job, vvr_motion* dst, size_t num_entries);
/* Two-step submission, for measurements of the device pipeline alone (bench.py's device_only_fps) and for pictures that are reconstructed more
 * than once: vvr_prepare validates the description, runs the host glue (builds the device work lists the reference iterates over in
 * DecCu::TaskTrafoCtu / TaskInterCtu, DecCu.cpp:106-134) and makes everything resident in HBM - it allocates device memory of the picture's
 * size and copies through pinned staging of the context with a blocking call, i.e. it is a set-up call, not part of a pipeline; vvr_submit_prepared only enqueues kernels.
 * A host that streams pictures uses vvr_submit with vvr_config.host_threads > 0: the same steps on the library's worker threads and upload ring,
 * without an allocation per picture. */
typedef struct vvr_prepared vvr_prepared;
VVR_API int          vvr_prepare(vvr_context* ctx, const vvr_picture* host_pic, vvr_prepared** out);
VVR_API int          vvr_submit_prepared(vvr_context* ctx, vvr_prepared* prepared);
VVR_API void         vvr_free_prepared(vvr_context* ctx, vvr_prepared* prepared);
/* the HIP stream (hipStream_t) job `job` runs on, and the per-kernel timing of the last waited job (bench/profiling) */
VVR_API void*        vvr_job_stream(vvr_context* ctx, int job);
/* External producers and consumers of DPB slots - the collective that replicates a reference picture to the GPUs whose pictures predict from it
 *  - ordered on the DEVICE: the host never waits for a picture.
 *   vvr_stream_wait_job   `stream` (hipStream_t of the caller) waits for picture `job`: what a sender does before the collective reads the slot;
 *   vvr_stream_wait_slot  `stream` waits for every picture submitted so far that reads or writes `slot` (and for earlier external users): what a
 *                         receiver does before the collective overwrites the slot;
 *   vvr_slot_external_event  pictures submitted from now on that use `slot` wait for `event` (hipEvent_t, recorded by the caller behind its
 *                         collective) first; writes != 0: the external work wrote the slot (earlier users were ordered before it by
 *                         vvr_stream_wait_slot), 0: it only reads it (a sender: later pictures must not overwrite the slot under it).  The
 *                         back-end keeps the handle until the slot is next written (by a picture or by an external writer) or until
 *                         vvr_sync finds the event complete - nowhere else does it look: the caller keeps the event alive until it is
 *                         complete AND a vvr_sync has returned since (or the slot has been overwritten).  The event should be recorded
 *                         before it is registered: one that is recorded later is still waited for by every picture handed to the device
 *                         after the record, but a vvr_sync in between forgets it (an event that was never recorded reads as complete).
 * A picture can only be waited for once it has been handed to the device (its work lists are built by worker threads): blocking = 0 returns
 * VVR_NOT_READY instead of waiting for that on the host.                                                                                      */
VVR_API int          vvr_stream_wait_job(vvr_context* ctx, int job, void* stream, int blocking);
VVR_API int          vvr_stream_wait_slot(vvr_context* ctx, int slot, void* stream, int blocking);
VVR_API int          vvr_slot_external_event(vvr_context* ctx, int slot, void* event, int writes);
VVR_API const char*  vvr_last_error(const vvr_context* ctx);
VVR_API const char*  vvr_version(void);
/* sizeof() of ABI struct number `which` as this library was compiled (0 vvr_pic_header, 1 vvr_cu, 2 vvr_tu, 3 vvr_motion,
 * 4 vvr_lfp, 5 vvr_sao_ctu, 6 vvr_alf_ctu, 7 vvr_alf_params, 8 vvr_lmcs_params, 9 vvr_picture, 10 vvr_config,
 * 11 vvr_kernel_stat, 12 vvr_wp_params, 13 vvr_scaling_list, 14 vvr_subpic, 15 vvr_slice_header; anything else 0): lets a binding written in another language verify its struct mirror at load time */
VVR_API size_t       vvr_abi_sizeof(int which);

/* kernel statistics accumulated with HIP events on the launch streams when enabled */
typedef struct vvr_kernel_stat {
  char     name[32];
  uint64_t launches;
  double   total_ms;
  double   algo_bytes;           /* algorithmic bytes moved by these launches (SURVEY.md §8(d) model) */
} vvr_kernel_stat;
VVR_API int          vvr_enable_stats(vvr_context* ctx, int on);
VVR_API int          vvr_get_stats(vvr_context* ctx, vvr_kernel_stat* out, int max_entries);
/* practical HBM ceiling of this device: bytes per second (read + written) of the library's device copy kernel over as much of the DPB
 * as the context's scratch planes hold (hundreds of MB for a 4K context: far beyond the caches), HIP-event timed, averaged over `iters` launches;
 * the DPB is only read, the scratch planes hold nothing between pictures */
VVR_API double       vvr_measure_copy_bandwidth(vvr_context* ctx, int iters);

/* Host glue helpers (pure functions, no device): what the reference computes per CU/TU on the CPU before the
 * arithmetic starts.  They are part of the ABI so that the parser-side integration and the tests use one definition. */
/* TrQuant::getTrTypes (TrQuant.cpp:330) -> (ver<<2)|hor */
VVR_API uint8_t      vvr_resolve_tr_type(const vvr_pic_header* hdr, const vvr_cu* cu, const vvr_tu* tu, int comp, int implicit_mts, int explicit_mts_intra, int explicit_mts_inter);

#ifdef __cplusplus
}
#endif
#endif /* VVR_H */
