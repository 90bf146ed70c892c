// vvdec_amd/csrc/vvr_device.h — device-side view shared by the HIP kernels and the host scheduler (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "../../include/vvr.h"

typedef int16_t pel_t;

// One picture's planes in HBM.  Rows are 128-byte aligned (stride in samples is a multiple of 64) so that a wavefront
// reading 64 consecutive samples touches exactly one or two 128-B lines.  No border margins: reference reads clamp
// their coordinates, which equals reading the reference decoder's border-extended picture (Picture.cpp:400-518).
struct DevPlanes {
  pel_t* p[3];
  int    stride[3];
  int    w[3], h[3];
};

// Work items built by the host glue (vvr_prepare) from the CU/TU records ---------------------------------------------

// One motion-compensation tile: at most 16x16 luma samples (+ the co-located 8x8 chroma samples) of one inter CU.
// 16x16 is the unit VVC itself uses for DMVR/BDOF processing (DMVR_SUBCU 16x16, MAX_BDOF_APPLICATION_REGION 16).
#define MC_ITEM_SUBBLOCK 1     /* SbTMVP: the tile is one 8x8 sub-block, motion from the motion field */
struct McItem {
  uint16_t x, y;       // luma position
  uint8_t  w, h;       // luma size: 4, 8 or 16
  uint16_t flags;      // MC_ITEM_*
  uint32_t cu;         // index into the CU array
  // plain (k_mc) tiles are self-contained: the kernel starts its reference fetch after ONE dependent load (this record) instead of
  // item -> CU -> motion field; GPM tiles still read their CU
  int32_t  mv[2][2];   // motion vectors of the two lists (1/16 sample), unclipped
  int8_t   ref[2];     // reference indices (-1: list not used)
  uint8_t  bcw;        // BCW weight index (2 = equal weights)
  uint8_t  clipW4;     // 0: the MV clipping refers to the CU's width; else the width / 4 of the block it refers to (SbTMVP under reference wrap-around: the joined piece)
  uint16_t clipX, clipY;   // position the MV clipping refers to (the CU; for SbTMVP the sub-block itself or, under reference wrap-around, the joined piece it belongs to)
};
// An inter CU whose tiles the DEVICE writes (k_expand_mc): the host only counts the tiles of such a CU - plain, BDOF and DMVR tiles are a function of
// the CU record alone (SbTMVP and affine tiles carry motion of the motion field, which stays on the host: those the host writes itself)
struct McCuRef { uint32_t cu; uint32_t first; };      // index into the CU array; bits 30..31: list (0 plain, 1 BDOF, 2 DMVR), bits 0..29: index of the CU's first tile in that list's device-written part
#define MC_ITEM_UNI   2    /* one prediction only: a single list, or identical motion in both (xCheckIdenticalMotion) */
#define MC_ITEM_HPEL  4    /* half-sample AMVR: alternative luma half-sample filter */
#define MC_ITEM_GEO   8
#define MC_ITEM_AFFINE 16  /* (tiles of k_mc_rpr only) affine CU: sub-block MVs like the tiles of k_mc_affine */

// One transform block that carries a residual.
struct TbItem {
  uint32_t tu;         // index into the TU array
  uint8_t  comp;       // component the coded levels belong to
  uint8_t  mode;       // TB_ADD: reco += residual (inter CU, prediction already in the picture); TB_STORE: write residual plane
  uint8_t  ict;        // 0, or 4 + ICT mode (-3..3 -> 1..7): joint Cb-Cr, the item writes both chroma blocks
  uint8_t  pad;        // TB_P_*: what k_itrans must know of the CU before it has the CU record (it asks for the record, the levels and the basis rows at once)
};
enum { TB_ADD = 0, TB_STORE = 1 };
enum { TB_P_CUGEOM = 1 /* chroma block of an ISP CU: position and size are the CU's */, TB_P_BDPCM = 2 /* BDPCM: every level of the block is coded */,
       TB_P_LFNST = 4 /* the CU applies LFNST to this component: cu.lfnst_idx > 0 && ( cu.tree != VVR_TREE_JOINT || comp == 0 ) */ };

// One intra-predicted transform block (decode order inside its CTU).  The reference-sample availability counts are the
// m_neighborSize[] values IntraPrediction::xFillReferenceSamples derives by walking the CU/TU tree (IntraPrediction.cpp:1104-1139);
// that walk is host glue here (vvr_prepare), the kernel only consumes the counts.
#define IT_F_RESI     1
#define IT_F_BDPCM_H  2
#define IT_F_MIP      8      /* luma: matrix-based intra prediction: mode = matrix index, bit 4 = transposed */
#define IT_F_ISP      6   /* both BDPCM bits (never together otherwise): luma intra sub-partition; IntraItem::tu then holds x - cuX | ( y - cuY ) << 6
                             | log2 cuW << 12 | log2 cuH << 15 | vertical split << 18 | residual flags of the partitions of a group << 19 (4 bits) | group << 23 (1: two 2-wide, 2: four 1-wide partitions) */
#define IT_MODE_RESI_ADD 255 /* mode value: no prediction, the (LMCS-scaled) residual is added to the inter prediction already in the picture */
#define IT_MODE_IBC      254 /* mode value: intra block copy, the prediction is a copy of reconstructed samples of this picture; IntraItem::tu = dx & 0xffff | dy << 16 (component samples) */
#define IT_F_CSCALE   8      /* chroma: LMCS chroma residual scaling applies to the residual */
#define IT_F_BDPCM_V  4      /* bits 4..5: multi-reference-line index; bits 6..7: CIIP intra weight (0 = ordinary intra block) */
struct IntraItem {        // 16 bytes, self-contained: the kernel never touches the CU/TU records on its serial path
  uint16_t x, y;          // block position in the component plane
  uint8_t  lw, lh;        // log2 size
  uint8_t  mode;          // 0 planar, 1 DC, 2..66 angular (before the wide-angle remap)
  uint8_t  flags;         // IT_F_*
  uint8_t  nTL;           // bit 0: top-left reference sample available; bits 1..3: row part of the block this item predicts, bits 4..5: log2( parts )
                          // (a block of more than IT_SPLIT_SAMPLES samples is predicted by 2, 4 or 8 wavefronts, each a band of rows: one item per band)
  uint8_t  nA, nL;        // available units (4 luma samples): above incl. above-right, left incl. below-left
  uint8_t  comp;          // bits 0..1: component; bits 2..7: `indep` - the block reads nothing that the `indep` blocks before it in its unit produce
                          // (it starts when every block of the unit up to index - indep - 1 is done)
  uint32_t tu;
};
#define IT_PART( it )    ( ( (it).nTL >> 1 ) & 7 )
#define IT_LPARTS( it )  ( ( (it).nTL >> 4 ) & 3 )
#define IT_COMP( it )    ( (it).comp & 3 )
#define IT_INDEP( it )   ( (it).comp >> 2 )
#define IT_PART_SAMPLES 1024   /* samples of an item whose residual is kept in the wavefront's LDS scratch (larger items - MIP, CCLM, IBC blocks that are not split - read it from the residual plane) */
#define IT_SPLIT_SAMPLES 256   /* ordinary prediction modes: a block of more samples becomes 2, 4 or 8 items (bands of rows), so that a wavefront predicts a band in one round of four samples per lane */
#define IT_MAX_LPARTS 3

// One unit of the intra stage (blocks of one component inside one CTU that read reference samples from each other, or a group of such
// clusters at the same dependency depth), processed by one workgroup.
#define VVR_INTRA_MAX_DEPS 26
struct IntraUnit {
  uint32_t ent;           // (component << 24) | CTU address; bit 29: has blocks with LMCS chroma residual scaling; bit 30: another unit waits for this one (it must publish its flag)
  uint32_t i0, i1;        // item range
  uint32_t bbox;          // (whole-CTU units only) part of the CTU tile the unit's blocks read: y0 | y1 << 8 | c0 << 16 | c1 << 24 (rows from CTU top - 3, 16-byte chunks from CTU left - 8, chunk + 1)
  uint32_t ndeps;
  uint32_t deps[VVR_INTRA_MAX_DEPS];   // tickets (= indices into the unit table) this unit waits for
  uint32_t iA;                         // [i0, iA): IT_MODE_RESI_ADD items (no mutual dependencies, done first and in parallel); 32 dwords
};

struct PicDev {         // everything a kernel needs about one picture (passed by value)
  vvr_pic_header     hdr;
  const vvr_cu*      cu;
  const vvr_tu*      tu;
  const int16_t*     coef;
  const vvr_motion*  affMotion;      // motion of the 4x4 sub-blocks of the affine tiles: 16 entries (4 x 4) per tile, first entry = McItem::mv[0][0]
  const vvr_lfp*     lfp[2];
  const vvr_sao_ctu* sao;
  const vvr_alf_ctu* alf;
  const vvr_alf_params* alf_params;
  const vvr_lmcs_params* lmcs;       // LMCS tables (NULL when off)
  const vvr_scaling_list* scaling;   // explicit scaling lists (NULL unless VVR_TOOL_SCALING_LIST)
  const vvr_wp_params*   wp;         // explicit weighted prediction table (NULL unless VVR_TOOL_WP on a P / B picture)
  const uint16_t*    ctuSlice;       // slice / tile index of every CTU (NULL: one slice / one tile): where SAO and ALF stop when the picture says so
  const uint16_t*    ctuTile;
  const vvr_slice_header* slices;    // headers of the slices (indexed by ctuSlice), NULL: every slice takes hdr's values; alf_params / wp then are arrays
  int                numAlfSets, numWpSets;
  const vvr_rpr_params* rpr;         // reference picture resampling: how the picture sees its reference pictures (NULL: none is scaled)
  const vvr_subpic*  subpics;        // sub-pictures (NULL: the picture is its only sub-picture) and the sub-picture of every CTU: MC of a CU in a sub-picture
  const uint16_t*    ctuSubpic;      // treated as a picture stays inside it; SAO / ALF of a CTU whose sub-picture says so do not look into other sub-pictures
  const uint32_t*    csVpdu;         // LMCS chroma residual scaling, per VPDU: x | y << 13 | hasLeft << 26 | hasAbove << 27 of the luma neighbourhood the factor is averaged over
  vvr_motion*        colMotion;      // collocated motion of the picture (pinned host memory, device-mapped; NULL unless VVR_TOOL_COL_MOTION): the DMVR kernel patches it
  int                colStride;      // records per row = ( w4 + 1 ) / 2
  int                vpdusX, vpduLog2;
  int                w4, h4, ctus_x, ctus_y;
};

struct RefSet { const pel_t* p[2 * VVR_MAX_REFS][3]; };   // reference planes indexed [list * 16 + refIdx][comp]; geometry = the current picture's

// kernel launchers (vvr_kernels.hip) ----------------------------------------------------------------------------------
void launch_mc     ( hipStream_t s, const PicDev& pic, const RefSet& refs, DevPlanes reco, const McItem* items, int numItems, const McItem* items2, int numItems2, int bdof );      // tiles of two arrays (host-written, device-written) in one launch
void launch_itrans ( hipStream_t s, const PicDev& pic, DevPlanes reco, DevPlanes resi, const TbItem* items, int numItems, int sizeClass );
// the deblocking edge parameters of the picture from its CU / TU records (vvr_lf_init.h): cell maps, motion of sub-block CUs, one thread per cell and direction
void launch_lf_init( hipStream_t s, const PicDev& pic, uint32_t numCu, uint32_t numTu, struct LfCell* cell, struct LfCell* cellC, struct LfMv* mv, uint32_t* ref, const struct LfSbCell* sbCells, int numSbCells, vvr_lfp* out0, vvr_lfp* out1 );
void launch_deblock( hipStream_t s, const PicDev& pic, DevPlanes reco, int dir );
void launch_deblock_tile( hipStream_t s, const PicDev& pic, DevPlanes src, DevPlanes dst, int dir, bool lmcs );      // one direction out of place (tiles); vertical edges: inverse LMCS in the load
void launch_sao    ( hipStream_t s, const PicDev& pic, DevPlanes src, DevPlanes dst );
void launch_alf    ( hipStream_t s, const PicDev& pic, DevPlanes src, DevPlanes dst );
bool sao_alf_fused ( const PicDev& pic );      // SAO + ALF in one pass (launch_sao_alf) apply to this picture; else launch_sao, launch_alf
void launch_sao_alf( hipStream_t s, const PicDev& pic, DevPlanes src, DevPlanes dst, bool sao, bool alf );
void launch_lmcs   ( hipStream_t s, const PicDev& pic, DevPlanes reco, int inverse );
void launch_copy_planes( hipStream_t s, DevPlanes src, DevPlanes dst );
void launch_copy_bytes( hipStream_t s, const void* src, void* dst, size_t bytes );
void launch_output_window( hipStream_t s, const pel_t* src, int stride, int w, int h, int bytesPerSample, void* dst );      // window rows packed back to back, 1 or 2 bytes per sample
void launch_plane_hash_rows( hipStream_t s, const pel_t* plane, int stride, int w, int h, int two, int crcMode, uint32_t* out );   // per row: checksum share / CRC piece
void launch_mc_affine( hipStream_t s, const PicDev& pic, const RefSet& refs, DevPlanes reco, const McItem* items, int numItems );
void launch_mc_rpr( hipStream_t s, const PicDev& pic, const RefSet& refs, DevPlanes reco, const McItem* items, int numItems );      // tiles of CUs with a scaled reference picture
void launch_mc_dmvr( hipStream_t s, const PicDev& pic, const RefSet& refs, DevPlanes reco, const McItem* items, int numItems, int32_t* dmvrOut );
void launch_intra  ( hipStream_t s, const PicDev& pic, DevPlanes reco, DevPlanes resi, const IntraItem* items, int numItems, const IntraUnit* units, int numUnits, int ticket0, int ticket1, int numWorkgroups, int* sync, int wide,
                     uint32_t* maps = nullptr, size_t mapInts = 0, int mapW4 = 0, int mapH4 = 0, int* errWord = nullptr );      // maps (a picture whose units are all whole CTUs of intra CUs): the per-cell words - the CTU wavefront is resolved block by block (k_intra<.., FINE>)      // the units [ticket0, ticket1); wide: an I picture the stream waits for (eight wavefronts per workgroup)
void launch_resi_add( hipStream_t s, const PicDev& pic, DevPlanes reco, DevPlanes resi, const IntraItem* items, int numItems );      // scaled chroma residuals of inter blocks (between the luma and the chroma units)
size_t intra_sync_ints( int numUnits, int numItems );      // ints `sync` has to hold: ticket, unit flags, the blocks' parameter records
// the intra stage of a picture with scattered intra blocks (vvr_intra_leaf.inc): one wavefront per block of `items` (decoding order per component), ordered through
// per-cell words in `maps` (intra_leaf_map_ints words for the largest picture of the context: all zero between launches)
#define IT_MODE_CSFAC 253      /* mode value: the LMCS chroma scaling factor of VPDU IntraItem::tu (no samples) */
size_t intra_leaf_map_ints( int w4, int h4, int vpdus );
// The picture's first launch (k_prep): the passes over its records that depend on nothing but the uploaded image - the motion-compensation tiles of the CUs
// the host only counted, the cell maps of the deblocking edge derivation (lfMaps), the cells and the ticket of the scattered intra blocks (numItems)
struct PrepWork
{
  const McCuRef* mcCus = nullptr; int numMcCus = 0; McItem *plain = nullptr, *bdof = nullptr, *dmvr = nullptr;
  bool lfMaps = false; uint32_t numCu = 0, numTu = 0; struct LfCell *cell = nullptr, *cellC = nullptr; struct LfMv* mv = nullptr; uint32_t* ref = nullptr; const struct LfSbCell* sb = nullptr; int numSb = 0;
  const IntraItem *items = nullptr, *resi = nullptr; int numItems = 0, numResi = 0; uint32_t* maps = nullptr; size_t mapInts = 0; int mapW4 = 0, mapH4 = 0;
};
void launch_prep( hipStream_t s, const PicDev& pic, const PrepWork& w );
void launch_intra_leaf( hipStream_t s, const PicDev& pic, DevPlanes reco, DevPlanes resi, const IntraItem* items, int numItems, const IntraItem* resiItems, int numResi /* residual-add blocks grouped by VPDU: done by the VPDU's IT_MODE_CSFAC item */,
                        uint32_t* maps, size_t mapInts, int mapW4, int mapH4, int* errWord /* the job's error word: pinned host memory the device writes when a bounded wait gave up */ );

