// vvdec_amd/csrc/vvr_kernels.hip — hand-written HIP kernels of the VVC reconstruction back-end for gfx950 (CDNA4).
//
// These are integer stencil / filter kernels bounded by HBM bandwidth; there is no MFMA anywhere (the only contraction,
// the inverse transform, has N <= 64 and integer rounding semantics).  Design rules followed (cdna_hip_programming.md §2, §6):
// 64-wide wavefronts, 256-thread workgroups, reference / coefficient tiles staged through LDS, coalesced row-wise
// plane accesses (rows are 128-B aligned), constant tables read through the scalar/L1 path.
//
// Each kernel cites the reference function whose arithmetic it reproduces (paths relative to source/Lib of VVdeC);
// bit-exactness is checked against the CPU restatement in oracle/ and the reference-driven golden fixtures.
#include <type_traits>
#include "vvr_device.h"
#include "vvr_lf_init.h"

namespace tbl {
#include "../../tables/vvc_tables.inc"
}

namespace {

__device__ __forceinline__ int clip3( int lo, int hi, int v ) { return v < lo ? lo : ( v > hi ? hi : v ); }
// clip to the sample range: ONE v_med3_i32 (the compiler cannot prove 0 <= 2^bd - 1 for a run-time bit depth and emits min / compare / select)
__device__ __forceinline__ int clip_pel( int v, int bd ) { const int hi = ( 1 << bd ) - 1; int r; asm( "v_med3_i32 %0, %1, 0, %2" : "=v"( r ) : "v"( v ), "v"( hi ) ); return r; }
__device__ __forceinline__ int iabs( int v ) { return v < 0 ? -v : v; }
__device__ __forceinline__ int ilog2( int v ) { return 31 - __clz( v ); }
__device__ __forceinline__ int sgn( int v ) { return ( v > 0 ) - ( v < 0 ); }
// LMCS: the luma prediction of an inter CU is mapped forward before anything is added to it (DecCu.cpp:458-476, Reshape::rspFwdCore) - applied where
// the motion-compensation kernels store their luma samples (the table is 2 KB and stays in the vector cache), not by a pass of its own
__device__ __forceinline__ int lmcs_fwd_luma( const int16_t* __restrict__ fwdLut, int c, int v ) { return ( fwdLut && c == 0 ) ? (int) fwdLut[v] : v; }

} // namespace

// ---- device copies of the constant tables (initialised from the generated include at load time) ----------------------
#define TR_ALIGN __attribute__( ( aligned( 16 ) ) )      /* (k_itrans stages the basis rows with dword loads) */
__device__ TR_ALIGN int16_t d_dct2_2[4], d_dct2_4[16], d_dct2_8[64], d_dct2_16[256], d_dct2_32[1024], d_dct2_64[4096];
__device__ TR_ALIGN int16_t d_dct8_4[16], d_dct8_8[64], d_dct8_16[256], d_dct8_32[1024];
__device__ TR_ALIGN int16_t d_dst7_4[16], d_dst7_8[64], d_dst7_16[256], d_dst7_32[1024];
__device__ int8_t  d_lfnst8x8[4][2][48][16], d_lfnst4x4[4][2][16][16];
__device__ uint8_t d_lfnst_lut[97], d_lfnst_scan8x8_xy[16][2], d_lfnst_scan4x4_xy[16][2];
__device__ int32_t d_inv_quant_scales[2][6];
__device__ uint8_t d_mip_matrix_4x4[16][16][4], d_mip_matrix_8x8[8][16][8], d_mip_matrix_16x16[6][64][7];
__device__ int16_t d_geo_params[64][2], d_geo_weight_offset[64][4][4][2];
__device__ int8_t  d_geo_weights[6][112 * 112], d_geo_angle2mask[32], d_geo_angle2mirror[32];
__device__ int16_t d_luma_filter[16][8], d_luma_filter_4x4[16][8], d_luma_alt_hpel[8], d_chroma_filter[32][4];
__device__ int16_t d_luma_filter_rpr1[16][8], d_luma_filter_rpr2[16][8], d_affine_luma_filter_rpr1[16][8], d_affine_luma_filter_rpr2[16][8], d_chroma_filter_rpr1[32][4], d_chroma_filter_rpr2[32][4];
__device__ int8_t  d_bcw_weights[5];
__device__ uint16_t d_db_tc_table[66];
__device__ uint8_t d_db_beta_table[64];
__device__ int16_t d_alf_fixed_coeff[64][13];
__device__ uint8_t d_alf_class_to_filter[16][25];

static int vvr_upload_mc_taps();
#define UPLOAD( name ) do { hipError_t e = hipMemcpyToSymbol( HIP_SYMBOL( d_##name ), tbl::vvc_##name, sizeof( tbl::vvc_##name ) ); if( e != hipSuccess ) return (int) e; } while( 0 )
int vvr_upload_tables()
{
  UPLOAD( dct2_2 ); UPLOAD( dct2_4 ); UPLOAD( dct2_8 ); UPLOAD( dct2_16 ); UPLOAD( dct2_32 ); UPLOAD( dct2_64 );
  UPLOAD( dct8_4 ); UPLOAD( dct8_8 ); UPLOAD( dct8_16 ); UPLOAD( dct8_32 );
  UPLOAD( dst7_4 ); UPLOAD( dst7_8 ); UPLOAD( dst7_16 ); UPLOAD( dst7_32 );
  UPLOAD( lfnst8x8 ); UPLOAD( lfnst4x4 ); UPLOAD( lfnst_lut ); UPLOAD( lfnst_scan8x8_xy ); UPLOAD( lfnst_scan4x4_xy );
  UPLOAD( inv_quant_scales ); UPLOAD( luma_filter ); UPLOAD( luma_filter_4x4 ); UPLOAD( luma_alt_hpel ); UPLOAD( chroma_filter );
  UPLOAD( bcw_weights ); UPLOAD( db_tc_table ); UPLOAD( db_beta_table ); UPLOAD( alf_fixed_coeff ); UPLOAD( alf_class_to_filter );
  UPLOAD( mip_matrix_4x4 ); UPLOAD( mip_matrix_8x8 ); UPLOAD( mip_matrix_16x16 );
  UPLOAD( luma_filter_rpr1 ); UPLOAD( luma_filter_rpr2 ); UPLOAD( affine_luma_filter_rpr1 ); UPLOAD( affine_luma_filter_rpr2 ); UPLOAD( chroma_filter_rpr1 ); UPLOAD( chroma_filter_rpr2 );
  UPLOAD( geo_params ); UPLOAD( geo_weight_offset ); UPLOAD( geo_weights ); UPLOAD( geo_angle2mask ); UPLOAD( geo_angle2mirror );
  return vvr_upload_mc_taps();
}

// =====================================================================================================================
// k_mc — motion compensation of one <=16x16 luma tile (+ chroma) per workgroup.
//   InterPrediction::xPredInterBlk (InterPrediction.cpp:751) + InterpolationFilter::filter<N> (InterpolationFilter.cpp:556)
//   + filterCopy (:424) + AreaBuf::addAvg / addWeightedAvg (Buffer.cpp:441,372) + clipMvInPic (Mv.cpp:64).
// The (w+7)x(h+7) reference window is staged once in LDS with clamped coordinates (= border-extended reference), the
// horizontal pass writes its 16-bit intermediates to LDS, the vertical pass reads them back column-wise.
// =====================================================================================================================
#define IF_INTERNAL_OFFS 8192
#define BDOF_S 20           // row stride of the padded 18x18 BDOF prediction block

// One (list, component) prediction segment of a tile.
struct McSeg {
  int x0, y0;            // reference-plane position of the window's first COPIED sample (may be outside: reads are clamped)
  int ww, wh;            // size of the window held in LDS
  int xFrac, yFrac;
  int w, h;              // block size
  int ox, oy;            // position of the block's integer-sample origin inside the window
  int padOff, cw, chh;   // DMVR padded copy (xPrefetchPad): window sample (u,v) = copied sample (clamp(u + shX - padOff, 0, cw - 1), clamp(v + shY - padOff, 0, chh - 1))
  int shX, shY;          // displacement of the window inside the padded copy (the integer part of the DMVR refinement)
  int bx0, by0, bx1, by1;// what may be read of the reference plane (component samples, inclusive): the plane, or the CU's sub-picture (mc_bounds)
  int wrapOff;           // > 0: the reference is read as if it wrapped around horizontally at this period (component samples): the reference's wrap copy
};


// XCD-aware mapping (cdna_hip_programming.md T1): consecutive workgroups are dealt round-robin to the 8 XCDs; give each XCD
// a contiguous run of tiles so that the halo rows shared by neighbouring tiles hit the same L2.
// XCD-aware work assignment: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2.  Remapping the linear id
// so that every XCD owns one contiguous range of work items keeps neighbours (which share cache lines: halos, 128-byte lines cut by
// tile edges) in the same L2.  (PMC: k_alf_luma fetched 3.5x its input through the fabric before.)
__device__ __forceinline__ int xcd_contiguous( int lin, int nwg )
{
  const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7;
  return ( xcd < r ? xcd * ( q + 1 ) : r * ( q + 1 ) + ( xcd - r ) * q ) + ( lin >> 3 );
}
__device__ __forceinline__ int mc_item_index()
{
  const int nwg = gridDim.x, bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return ( xcd < r ? xcd * ( q + 1 ) : r * ( q + 1 ) + ( xcd - r ) * q ) + ( bid >> 3 );
}

// What motion compensation of a CU may read of a reference picture (luma samples, inclusive): the picture - or, for a CU in a sub-picture that is
// treated as a picture, that sub-picture: the reference predicts such CUs from a copy of the sub-picture with its own replicated border
// (Picture::getSubPicBuf, DecLibRecon::createSubPicRefBufs, DecLibRecon.cpp:388-421), which is the clamp to its rectangle here.
// Slices with headers of their own (vvr_picture.slices, ABI 4): the header of the slice a luma position lies in; the tool switches that hold there
// (the slice's value for the switches a slice header carries - VVR_SLICE_TOOL_MASK -, the picture's for the rest); the slice's ALF / weight tables.
// Without slice headers everything is the picture's (one uniform branch).
__device__ __forceinline__ const vvr_slice_header* slice_at( const PicDev& pic, int lx, int ly )
{
  if( !pic.slices ) return nullptr;
  return &pic.slices[pic.ctuSlice[( ly >> pic.hdr.log2_ctu ) * pic.ctus_x + ( lx >> pic.hdr.log2_ctu )]];
}
__device__ __forceinline__ uint32_t flags_at( const PicDev& pic, int lx, int ly )
{
  // (values, never a choice between a pointer into the kernel arguments and one into global memory: that would push the arguments into scratch)
  uint32_t f = pic.hdr.tool_flags;
  if( pic.slices ) f = ( f & ~(uint32_t) VVR_SLICE_TOOL_MASK ) | ( slice_at( pic, lx, ly )->tool_flags & VVR_SLICE_TOOL_MASK );
  return f;
}
// deblocking offsets ( beta | tc << 8, both + 64 ) of component c for the slice a luma position lies in
__device__ __forceinline__ int deblock_offsets_at( const PicDev& pic, int lx, int ly, int c )
{
  int beta = c == 0 ? pic.hdr.deblock_beta_offset_div2[0] : c == 1 ? pic.hdr.deblock_beta_offset_div2[1] : pic.hdr.deblock_beta_offset_div2[2];
  int tc   = c == 0 ? pic.hdr.deblock_tc_offset_div2[0]   : c == 1 ? pic.hdr.deblock_tc_offset_div2[1]   : pic.hdr.deblock_tc_offset_div2[2];
  if( pic.slices ) { const vvr_slice_header* s = slice_at( pic, lx, ly ); beta = s->deblock_beta_offset_div2[c]; tc = s->deblock_tc_offset_div2[c]; }
  return ( beta + 64 ) | ( ( tc + 64 ) << 8 );
}
__device__ __forceinline__ const vvr_wp_params* wp_at( const PicDev& pic, int lx, int ly )
{
  if( !pic.wp ) return nullptr;
  const vvr_slice_header* s = slice_at( pic, lx, ly );
  if( s && !( s->tool_flags & VVR_TOOL_WP ) ) return nullptr;
  return &pic.wp[s && pic.numWpSets > 1 ? s->wp_set : 0];
}
__device__ __forceinline__ const vvr_alf_params* alf_set_at( const PicDev& pic, int lx, int ly )
{
  const vvr_slice_header* s = slice_at( pic, lx, ly );
  return &pic.alf_params[s && pic.numAlfSets > 1 ? s->alf_set : 0];
}
// LMCS forward map of the slice a block lies in (nullptr: the slice does not use LMCS)
__device__ __forceinline__ const int16_t* lmcs_fwd_at( const PicDev& pic, int lx, int ly )
{
  if( !pic.lmcs ) return nullptr;
  return ( flags_at( pic, lx, ly ) & VVR_TOOL_LMCS ) ? pic.lmcs->fwd_lut : nullptr;
}
struct McBounds { int x0, y0, x1, y1; };
__device__ __forceinline__ McBounds mc_bounds( const PicDev& pic, int cuX, int cuY )
{
  McBounds b = { 0, 0, (int) pic.hdr.width - 1, (int) pic.hdr.height - 1 };
  if( pic.subpics )
  {
    const vvr_subpic sp = pic.subpics[pic.ctuSubpic[( cuY >> pic.hdr.log2_ctu ) * pic.ctus_x + ( cuX >> pic.hdr.log2_ctu )]];
    if( sp.treated_as_pic ) { b.x0 = sp.x0; b.y0 = sp.y0; b.x1 = sp.x1; b.y1 = sp.y1; }
  }
  return b;
}
// clipMvInPic / clipMvInSubpic (Mv.cpp:64,84) against the block at luma position (x, y)
__device__ __forceinline__ void mc_clip_mv( const PicDev& pic, const McBounds& b, int x, int y, int& mvx, int& mvy )
{
  const int ctu = 1 << pic.hdr.log2_ctu;
  const int horMax = ( b.x1 + 1 + 8 - x - 1 ) * 16, horMin = ( -ctu - 8 - ( x - b.x0 ) + 1 ) * 16;
  const int verMax = ( b.y1 + 1 + 8 - y - 1 ) * 16, verMin = ( -ctu - 8 - ( y - b.y0 ) + 1 ) * 16;
  mvx = min( horMax, max( horMin, mvx ) ); mvy = min( verMax, max( verMin, mvy ) );
}

// Reference wrap-around (vvr_pic_header.wrap_offset; pps_ref_wraparound_enabled_flag).  wrapClipMv (Mv.cpp:112) for the block at luma (x, y), bw wide:
// an MV that points further out than the margins of the reference's wrap copy is moved by one period and clamped; returns whether the wrap copy is
// the one to read (it is not after a move).
__device__ __forceinline__ bool mc_wrap_clip_mv( const PicDev& pic, int x, int y, int bw, int& mvx, int& mvy )
{
  const int ctu = 1 << pic.hdr.log2_ctu;
  const int horMax = ( pic.hdr.width + ctu - bw + 8 - x - 1 ) * 16, horMin = ( -ctu - 8 - x + 1 ) * 16;
  const int verMax = ( pic.hdr.height + 8 - y - 1 ) * 16, verMin = ( -ctu - 8 - y + 1 ) * 16;
  bool wrapRef = true;
  if( mvx > horMax ) { mvx -= pic.hdr.wrap_offset * 16; mvx = min( horMax, max( horMin, mvx ) ); wrapRef = false; }
  if( mvx < horMin ) { mvx += pic.hdr.wrap_offset * 16; mvx = min( horMax, max( horMin, mvx ) ); wrapRef = false; }
  mvy = min( verMax, max( verMin, mvy ) );
  return wrapRef;
}
// The MV clip of the regular prediction paths (clipMv = clipMvInPic, then wrapClipMv once more on the result: InterPrediction.cpp:651-656, 1751-1752,
// 1810-1815): without wrap-around the clamp of mc_clip_mv; with it the MV after wrapClipMv, and the second call - which finds the MV inside its range -
// always selects the wrap copy.  Returns the period to read the reference with (luma samples), 0 = ordinary clamped reads.
__device__ __forceinline__ int mc_clip_mv_w( const PicDev& pic, const McBounds& b, int x, int y, int bw, int& mvx, int& mvy )
{
  if( !pic.hdr.wrap_offset ) { mc_clip_mv( pic, b, x, y, mvx, mvy ); return 0; }
  mc_wrap_clip_mv( pic, x, y, bw, mvx, mvy );
  return pic.hdr.wrap_offset;
}
// column of the reference's wrap copy (Picture::extendPicBorderWrap, Picture.cpp:410-518): margin sample -k-1 = sample off-k-1 while k < off, else the edge
__device__ __forceinline__ int mc_ref_col( int x, int lo, int hi, int pw, int off )
{
  if( off ) { if( x < 0 ) return -x <= off ? x + off : 0; if( x >= pw ) return x - pw < off ? x - off : pw - 1; return x; }
  return clip3( lo, hi, x );
}

// window of one segment into LDS: clamped coordinates = border-extended reference (Picture::extendPicBorder)
// lanes map to (row, column) with 32 columns per row group: no integer division, rows are contiguous 2-byte runs
template<int NT>
__device__ __forceinline__ void mc_load_window( pel_t* win, int wst, const McSeg& g, const pel_t* __restrict__ ref, int stride, int pw, int ph, int tid )
{
  const int col = tid & 31, row0 = tid >> 5;
  // The window of nearly every tile lies inside the picture (the sub-picture): no coordinate is clamped, a lane walks down its column with one
  // address increment per row.  (The segment record sits in LDS: what decides the branch is made uniform first.)
  const int ux0 = __builtin_amdgcn_readfirstlane( g.x0 ), uy0 = __builtin_amdgcn_readfirstlane( g.y0 ), uww = __builtin_amdgcn_readfirstlane( g.ww ), uwh = __builtin_amdgcn_readfirstlane( g.wh );
  const int plain = __builtin_amdgcn_readfirstlane( ( g.wrapOff | g.padOff | g.shX | g.shY ) == 0 && g.cw == g.ww && g.chh == g.wh
                                                    && g.x0 >= g.bx0 && g.x0 + g.ww - 1 <= g.bx1 && g.y0 >= g.by0 && g.y0 + g.wh - 1 <= g.by1 );
  if( plain )
  {
    if( col < uww )
    {
      constexpr int RS = NT / 32;            // rows one pass of the wavefront(s) covers
      const pel_t* __restrict__ rp = ref + (size_t) uy0 * stride + ux0 + col;
      pel_t* wp = win + col;
      const int last = uwh - 1 - ( ( uwh - 1 - row0 ) % RS );      // last row of this lane's parity (the tail repeats it: same value to the same place)
      for( int yb = row0; yb < uwh; yb += 4 * RS )
      {
        int yy[4]; pel_t v[4];
#pragma unroll
        for( int u = 0; u < 4; u++ ) { yy[u] = min( yb + u * RS, last ); v[u] = rp[yy[u] * stride]; }
#pragma unroll
        for( int u = 0; u < 4; u++ ) wp[yy[u] * wst] = v[u];
      }
    }
    return;
  }
  if( col < g.ww )
  {
    const int sx = mc_ref_col( g.x0 + clip3( 0, g.cw - 1, col + g.shX - g.padOff ), g.bx0, g.bx1, pw, g.wrapOff );
    const pel_t* __restrict__ rc = ref + sx;
    // four rows per step, all four loads issued before the first LDS store: one memory round trip per four rows instead of one per
    // row (the tail repeats the last row: same value to the same place)
    for( int yb = row0; yb < g.wh; yb += 4 * ( NT / 32 ) )
    {
      int yy[4]; pel_t v[4];
#pragma unroll
      for( int u = 0; u < 4; u++ )
      {
        yy[u] = min( yb + u * ( NT / 32 ), g.wh - 1 - ( ( g.wh - 1 - row0 ) % ( NT / 32 ) ) );      // last row of this lane's parity
        const int sy = clip3( g.by0, g.by1, g.y0 + clip3( 0, g.chh - 1, yy[u] + g.shY - g.padOff ) );
        v[u] = rc[(size_t) sy * stride];
      }
#pragma unroll
      for( int u = 0; u < 4; u++ ) win[yy[u] * wst + col] = v[u];
    }
  }
}

// BDOF of one <= 16x16 luma tile (applyBiOptFlow :1290, gradFilterCore :213, BiOptFlowCore :162, calcBIOSums :134, addBIOAvg4 :108);
// the windows must hold the integer samples around the block (ox, oy >= 1 beyond the filter support).
// Round 6: no gradient arrays, no lane shuffles through LDS.  A lane owns ONE COLUMN of one row of 4x4 units (64 lanes = 4 unit rows x 16 columns): it forms the
// gradients and the five products of the unit row's six window rows from the predictions themselves (the block is stored with two rows of slack above and below its
// one-sample border, so that the eight rows the six positions read are plain offsets; the two positions a block edge replicates are copied from their neighbours
// afterwards), sums them down its column, and the six-column sums of a unit come together with DPP moves inside the 16-lane row - the column before / behind the unit
// from the neighbouring quad, or the lane's own value where the window is clamped at the block's edge.  Every lane then refines and writes the four samples of its column.
#define BDOF_PR 2           /* slack rows above the border row of the padded prediction block */
struct BdofShared {
  pel_t blk[2][( 16 + 2 + 2 * BDOF_PR ) * BDOF_S];       // 14-bit luma predictions with a one-sample border, stride BDOF_S; sample (x, y) at [( BDOF_PR + 1 + y ) * BDOF_S + 1 + x]
};
#define BDOF_AT( x, y ) ( ( BDOF_PR + 1 + ( y ) ) * BDOF_S + 1 + ( x ) )
template<int NT>
__device__ __forceinline__ void mc_bdof_luma( BdofShared& bs, const pel_t* win0, const pel_t* win1, int wst, const McSeg* seg /* luma segment of list 0 */, int segStride,
                                              int bd, const DevPlanes& reco, int x0, int y0, int w, int h, int tid, const int16_t* __restrict__ fwdLut /* LMCS forward map or nullptr */ )
{
  const int headroom = 14 - bd > 2 ? 14 - bd : 2;
  const McSeg& g0 = seg[0]; const McSeg& g1 = seg[segStride];
  // (1) bs.blk holds the 14-bit predictions of both lists (written by the vertical filter stage); add the border of nearest
  //     integer samples around them (xPredInterBlk :863-890): lanes 0 .. w + 1 the rows above and below, the next h lanes the columns left and right, both lists
  if( tid < w + 2 + h )
  {
    const bool row = tid < w + 2;
    const int q = row ? tid : tid - ( w + 2 );
#pragma unroll
    for( int l = 0; l < 2; l++ )
    {
      const McSeg& g = l ? g1 : g0;
      const pel_t* wn = l ? win1 : win0;
      const int xOff = g.xFrac < 8 ? 1 : 0, yOff = g.yFrac < 8 ? 1 : 0;
      // padded coordinates (bi, bj): (0, 0) = the corner above-left of the block
      const int biA = row ? q : 0, bjA = row ? 0 : 1 + q, biB = row ? q : w + 1, bjB = row ? h + 1 : 1 + q;
      const int sA = wn[( g.oy - yOff + bjA ) * wst + g.ox - xOff + biA], sB = wn[( g.oy - yOff + bjB ) * wst + g.ox - xOff + biB];
      bs.blk[l][BDOF_AT( biA - 1, bjA - 1 )] = (pel_t) ( ( sA << headroom ) - IF_INTERNAL_OFFS );
      bs.blk[l][BDOF_AT( biB - 1, bjB - 1 )] = (pel_t) ( ( sB << headroom ) - IF_INTERNAL_OFFS );
    }
  }
  __syncthreads();
  if( tid >= 64 ) return;
  const int X = tid & 15, yu = tid >> 4;                 // column, row of units
  const bool act = X < w && 4 * yu < h;
  const int shiftNum = 15 - bd, offset = ( 1 << ( shiftNum - 1 ) ) + 2 * IF_INTERNAL_OFFS;
  int sAGX = 0, sAGY = 0, sDIX = 0, sDIY = 0, sSG = 0;
  int dgx[4], dgy[4], psum[4];                           // per output row of the column: gx0 - gx1, gy0 - gy1, p0 + p1
  {
    const int xc = act ? X : 0, yb = act ? 4 * yu : 0;
    const pel_t* c0 = &bs.blk[0][BDOF_AT( xc, yb - 2 )];
    const pel_t* c1 = &bs.blk[1][BDOF_AT( xc, yb - 2 )];
    // rows yb - 2 .. yb + 5 of the column (row -2 / h + 1 of a unit row at the block's edge is slack: only the positions replaced below read it)
    int P0[8], P1[8];
#pragma unroll
    for( int j = 0; j < 8; j++ ) { P0[j] = c0[j * BDOF_S]; P1[j] = c1[j * BDOF_S]; }
    int aGX[6], aGY[6], dIX[6], dIY[6], sG[6];
#pragma unroll
    for( int k = 0; k < 6; k++ )                         // window row yb - 1 + k = register row k + 1
    {
      const int l0m = c0[( k + 1 ) * BDOF_S - 1], l0p = c0[( k + 1 ) * BDOF_S + 1], l1m = c1[( k + 1 ) * BDOF_S - 1], l1p = c1[( k + 1 ) * BDOF_S + 1];
      const int gx0 = ( l0p >> 6 ) - ( l0m >> 6 ), gx1 = ( l1p >> 6 ) - ( l1m >> 6 );
      const int gy0 = ( P0[k + 2] >> 6 ) - ( P0[k] >> 6 ), gy1 = ( P1[k + 2] >> 6 ) - ( P1[k] >> 6 );
      const int tGX = ( gx0 + gx1 ) >> 1, tGY = ( gy0 + gy1 ) >> 1;
      const int tDI = ( P1[k + 1] >> 4 ) - ( P0[k + 1] >> 4 );
      const int sgX = clip3( -1, 1, tGX ), sgY = clip3( -1, 1, tGY );
      aGX[k] = iabs( tGX ); aGY[k] = iabs( tGY );
      dIX[k] = __mul24( sgX, tDI ); dIY[k] = __mul24( sgY, tDI ); sG[k] = __mul24( sgY, tGX );
      if( k >= 1 && k <= 4 ) { dgx[k - 1] = gx0 - gx1; dgy[k - 1] = gy0 - gy1; psum[k - 1] = P0[k + 1] + P1[k + 1]; }
    }
    // the window rows outside the block repeat the block's first / last row (the padding of :236-264)
    const bool top = yb == 0, bot = yb + 4 >= h;
    if( top ) { aGX[0] = aGX[1]; aGY[0] = aGY[1]; dIX[0] = dIX[1]; dIY[0] = dIY[1]; sG[0] = sG[1]; }
    if( bot ) { aGX[5] = aGX[4]; aGY[5] = aGY[4]; dIX[5] = dIX[4]; dIY[5] = dIY[4]; sG[5] = sG[4]; }
#pragma unroll
    for( int k = 0; k < 6; k++ ) { sAGX += aGX[k]; sAGY += aGY[k]; sDIX += dIX[k]; sDIY += dIY[k]; sSG += sG[k]; }
  }
  // the six columns of the unit's window: the quad's four + the column before and behind it (the quad's own first / last column where the window leaves the block)
  const bool lastCol = X == w - 1;
  auto win6 = [&]( int v ) -> int
  {
    int q4 = v + __builtin_amdgcn_update_dpp( 0, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true );
    q4 += __builtin_amdgcn_update_dpp( 0, q4, 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, true );
    int before = __builtin_amdgcn_update_dpp( v, v, 0x111 /* row_shr:1: the column before; lane 0 of the row keeps its own */, 0xf, 0xf, false );
    before = __builtin_amdgcn_update_dpp( 0, before, 0x00 /* quad_perm [0,0,0,0] */, 0xf, 0xf, true );
    int behind = __builtin_amdgcn_update_dpp( v, v, 0x101 /* row_shl:1: the column behind; lane 15 keeps its own */, 0xf, 0xf, false );
    behind = lastCol ? v : behind;
    behind = __builtin_amdgcn_update_dpp( 0, behind, 0xFF /* quad_perm [3,3,3,3] */, 0xf, 0xf, true );
    return q4 + before + behind;
  };
  sAGX = win6( sAGX ); sAGY = win6( sAGY ); sDIX = win6( sDIX ); sDIY = win6( sDIY ); sSG = win6( sSG );
  if( act )
  {
    int tmpx = sAGX == 0 ? 0 : ( ( sDIX * 4 ) >> ( 31 - __clz( sAGX ) ) );        // rightShiftMSB (:92): shift by floor(log2(denominator))
    tmpx = clip3( -15, 15, tmpx );
    const int mains = sSG >> 12, secs = sSG & 4095;
    int tmpData = tmpx * mains;
    tmpData = ( ( tmpData * ( 1 << 12 ) ) + tmpx * secs ) >> 1;
    int tmpy = sAGY == 0 ? 0 : ( ( ( sDIY * 4 ) - tmpData ) >> ( 31 - __clz( sAGY ) ) );
    tmpy = clip3( -15, 15, tmpy );
    pel_t* dst = reco.p[0] + (size_t) ( y0 + 4 * yu ) * reco.stride[0] + x0 + X;
#pragma unroll
    for( int r = 0; r < 4; r++ )
    {
      const int b = tmpx * dgx[r] + tmpy * dgy[r];
      const int v = clip_pel( ( psum[r] + b + offset ) >> shiftNum, bd );
      dst[(size_t) r * reco.stride[0]] = (pel_t) lmcs_fwd_luma( fwdLut, 0, v );
    }
  }
}

// NT threads per tile.  With NT = 64 a tile is one wavefront: barriers are free, and the single exposure to global-memory
// latency (phase A) is hidden by the other resident tiles.
// ---------------------------------------------------------------------------------------------------------------------
// Register-blocked separable interpolation used by k_mc and k_mc_dmvr (round 6: the instruction diet).  Every segment goes through the same
// two stages - horizontal filter to 14-bit intermediates, vertical filter to the result - with the identity filter (64 at the centre
// tap) where the MV has no fractional part in that direction.  That is bit-exact with the reference's four code paths (copy /
// horizontal only / vertical only / separable, InterpolationFilter.cpp:556-651): the intermediate rounding of a pass with the identity
// filter is exact (64 * s = s << 6), see DESIGN.md §5.
//   * window rows lie row-major in LDS (two samples per dword); a tile whose windows lie inside the picture loads them as DWORDS, 16 lanes
//     per row, and realigns an odd start with one DPP move + v_alignbit (mc3_load_*): a quarter of the load / store instructions of the
//     per-sample loader, which stays for windows that need clamping, wrap-around or the padded copies of DMVR (mc_load_window);
//   * a stage-1 work item filters 8 columns of TWO window rows (four 16-byte LDS reads, 64 v_dot2_i32_i16) and writes the eight
//     results as eight dwords { row 2r, row 2r + 1 } into the column-major intermediate [column][row pair]: half the LDS stores of
//     a row per item, no per-column predicate, the rounding offset folded into the accumulator's start value;
//   * a stage-2 work item filters 8 rows of one column for both lists from two 16-byte reads per list; chroma columns run through the
//     SAME eight-tap code (their taps padded with zeros) so luma and chroma lanes of the wavefront do not diverge; the way the two
//     predictions are combined (average / BCW / GPM / weighted prediction / BDOF input) is decided once per tile, not per sample.
// The taps come from ONE table of packed pairs (d_mcTaps) indexed per lane: a lane of list 1 reads list 1's row.
// Range: the 14-bit intermediates and the second-stage sums >> 6 stay inside int16 for samples of up to 10 bits (positive taps sum to at
// most 88, negative ones to -24: |stage 1| <= 14330, |stage 2 >> 6| <= 25072), so the reference's stores to int16 never wrap and no
// sign-extension is spent on them.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mc_dot2( uint32_t a, uint32_t b, int c )
{
  typedef short s2v __attribute__( ( ext_vector_type( 2 ) ) );
  return __builtin_amdgcn_sdot2( __builtin_bit_cast( s2v, a ), __builtin_bit_cast( s2v, b ), c, false );
}

__device__ __forceinline__ int mc_dot2_init( uint32_t a, uint32_t b, int initUniform )
{
  int r; asm( "v_dot2_i32_i16 %0, %1, %2, %3" : "=v"( r ) : "v"( a ), "v"( b ), "s"( initUniform ) ); return r;
}
// rows of d_mcTaps: luma (frac 0..15) regular / for 4x4 blocks / with the alternative half-sample filter at frac 8 / both; chroma (frac 0..31), taps 4..7 zero
#define MCT_REG    0
#define MCT_4X4    16
#define MCT_ALT    32
#define MCT_CHROMA 64
__device__ uint4 d_mcTaps[96];
static int vvr_upload_mc_taps()
{
  uint32_t t[96][4];
  auto pack = []( const int16_t* c, int n, uint32_t* o ) { for( int i = 0; i < 4; i++ ) o[i] = 2 * i < n ? (uint32_t) (uint16_t) c[2 * i] | (uint32_t) (uint16_t) c[2 * i + 1] << 16 : 0u; };
  for( int f = 0; f < 16; f++ )
  {
    pack( tbl::vvc_luma_filter[f], 8, t[MCT_REG + f] ); pack( tbl::vvc_luma_filter_4x4[f], 8, t[MCT_4X4 + f] );
    pack( f == 8 ? tbl::vvc_luma_alt_hpel : tbl::vvc_luma_filter[f], 8, t[MCT_ALT + f] ); pack( f == 8 ? tbl::vvc_luma_alt_hpel : tbl::vvc_luma_filter_4x4[f], 8, t[MCT_ALT + MCT_4X4 + f] );
  }
  for( int f = 0; f < 32; f++ ) pack( tbl::vvc_chroma_filter[f], 4, t[MCT_CHROMA + f] );
  return (int) hipMemcpyToSymbol( HIP_SYMBOL( d_mcTaps ), t, sizeof( t ) );
}
// (InterpolationFilter.cpp:1078-1085 / 669-676: luma 4x4 blocks use the 6-tap table; :105 alternative half-pel filter - it wins at frac 8)
__device__ __forceinline__ int mc_tap_row( int c, int frac, bool f4, bool altHpel ) { return c ? MCT_CHROMA + frac : ( altHpel ? MCT_ALT : 0 ) + ( f4 ? MCT_4X4 : 0 ) + frac; }

// 8 outputs out[j] = init + sum_t s[j + t] * c[t], s = the 16 samples in lo / hi (two per dword), NTAPS = 8 or 4
template<int NTAPS>
__device__ __forceinline__ void mc_fir8( const uint4 lo, const uint4 hi, const uint4 Cv, int init, int ( &out )[8] )
{
  const uint32_t D[8] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
  const uint32_t C[4] = { Cv.x, Cv.y, Cv.z, Cv.w };
  uint32_t S[7];
#pragma unroll
  for( int i = 0; i < 7; i++ ) S[i] = __builtin_amdgcn_alignbit( D[i + 1], D[i], 16 );
  // the first tap takes the start value as the third operand of the three-address form (a uniform value in a scalar register): no register to initialise per output
#pragma unroll
  for( int m = 0; m < 4; m++ ) { out[2 * m] = mc_dot2_init( D[m], C[0], init ); out[2 * m + 1] = mc_dot2_init( S[m], C[0], init ); }
#pragma unroll
  for( int t = 1; t < NTAPS / 2; t++ )
  {
#pragma unroll
    for( int m = 0; m < 4; m++ ) { out[2 * m] = mc_dot2( D[m + t], C[t], out[2 * m] ); out[2 * m + 1] = mc_dot2( S[m + t], C[t], out[2 * m + 1] ); }
  }
}

#define MC2_WST_L 24        // luma window: 23 x 23 samples, row stride 24 (48 bytes: every row and every 8-sample group is 16-byte aligned)
#define MC2_WST_C 16        // chroma window: 11 x 11 samples, row stride 16
#define MC3_TPL   12        // intermediates: dwords per luma column (24 rows as 12 pairs; 48 bytes: 16-byte reads at rows 0 and 8)
#define MC3_TPC   8         // dwords per chroma column (12 rows as 6 pairs, padded to 8)
#define MC3_TBC   72        // dwords per chroma (list, component) block: 8 columns + 8 dwords that spread the blocks over the banks

// Explicit weighted prediction (WeightPrediction::getWpScaling / addWeightUni / addWeightBi, WeightPrediction.cpp:66-157,238-338,164-236):
// final stage of plain, SbTMVP, CIIP and affine predictions of a picture with VVR_TOOL_WP (unless BCW weights or GPM are in use);
// p, p0, p1 are the 14-bit intermediate predictions, headroom = max( 2, 14 - bitDepth )
__device__ __forceinline__ int wp_uni( const vvr_wp_params* __restrict__ wp, int l, int ri, int c, int p, int bd, int headroom )
{
  const vvr_wp_entry e = wp->e[l][ri][c];
  const int den = wp->log2_denom[c ? 1 : 0], shift = den + headroom, offset = e.offset * ( 1 << ( bd - 8 ) );
  if( e.weight != ( 1 << den ) ) return clip_pel( ( ( e.weight * ( p + IF_INTERNAL_OFFS ) + ( 1 << ( shift - 1 ) ) ) >> shift ) + offset, bd );
  return clip_pel( ( ( p + IF_INTERNAL_OFFS + ( 1 << ( headroom - 1 ) ) ) >> headroom ) + offset, bd );
}
__device__ __forceinline__ int wp_bi( const vvr_wp_params* __restrict__ wp, int r0, int r1, int c, int p0, int p1, int bd, int headroom )
{
  const vvr_wp_entry e0 = wp->e[0][r0][c], e1 = wp->e[1][r1][c];
  const int den = wp->log2_denom[c ? 1 : 0], shift = den + 1 + headroom, offset = ( e0.offset + e1.offset ) * ( 1 << ( bd - 8 ) );
  return clip_pel( ( e0.weight * ( p0 + IF_INTERNAL_OFFS ) + e1.weight * ( p1 + IF_INTERNAL_OFFS ) + ( ( 1 << shift ) >> 1 ) + offset * ( 1 << ( shift - 1 ) ) ) >> shift, bd );
}

// LDS working set of the two filter stages (one tile)
struct Mc2Shared {
  __attribute__( ( aligned( 16 ) ) ) pel_t winL[2][24 * MC2_WST_L];          // (row 23 is scratch: the second row of the last row pair)
  __attribute__( ( aligned( 16 ) ) ) pel_t winC[2][2][12 * MC2_WST_C];
  __attribute__( ( aligned( 16 ) ) ) uint32_t tmpL[2][16 * MC3_TPL];         // [column][row pair]
  __attribute__( ( aligned( 16 ) ) ) uint32_t tmpC[4][MC3_TBC];              // [list * 2 + Cb / Cr][column][row pair]
  McSeg seg[2][3];                                                           // (only tiles that take the per-sample loader)
  const pel_t* refp[2][3];
};
// what the stages need to know about the filters of a tile: rows of d_mcTaps per list, [luma, chroma]
struct McTapRows { int h[2][2], v[2][2]; };

// a pointer that is the same for every lane, moved to scalar registers
template<class P> __device__ __forceinline__ P* mc_uniform_ptr( P* p )
{
  const uint64_t a = (uint64_t) p;
  return (P*) ( (uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane( (int) (uint32_t) a ) | ( (uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane( (int) ( a >> 32 ) ) << 32 ) );
}
// ---- windows of a tile that lies inside the picture: dword loads, 16 (luma) / 8 (chroma) lanes per window row --------------------
__device__ __forceinline__ uint32_t mc_dpp_next_lane( uint32_t v ) { return (uint32_t) __builtin_amdgcn_update_dpp( 0, (int) v, 0x101 /* row_shl:1 */, 0xf, 0xf, true ); }

// The loads of a window and what becomes of them are two steps (issue / commit): a tile issues the loads of all its windows, then waits once
// (four windows loaded one after the other were four memory round trips of a wavefront that lives for ten)
template<int NT> struct Mc3Luma   { static constexpr int RP = NT / 16, NP = ( 23 + RP - 1 ) / RP; uint32_t v[NP]; };      // rows per pass; passes of a 23-row window
template<int NT> struct Mc3Chroma { static constexpr int RP = NT / 8,  NP = ( 22 + RP - 1 ) / RP; uint32_t v[NP]; };      // 2 x 11 rows at most
template<int NT>
__device__ __forceinline__ void mc3_issue_luma( Mc3Luma<NT>& R, const pel_t* __restrict__ ref, int stride, int x0, int y0, int ww, int wh, int tid )
{
  constexpr int RP = Mc3Luma<NT>::RP, NP = Mc3Luma<NT>::NP;
  const int q = tid & 15, r0 = tid >> 4;
  const int odd = x0 & 1;
  const int ndw = ( ww + 1 + odd ) >> 1;              // dwords read per row: samples x0 - odd .. x0 - odd + 2 ndw - 1
  const uint32_t* __restrict__ base = reinterpret_cast<const uint32_t*>( ref + (size_t) y0 * stride + ( x0 - odd ) );
  const uint32_t sd = (uint32_t) stride >> 1;
  const int last = wh - 1 - ( max( wh - 1 - r0, 0 ) % RP );    // the last row of this lane's residue class: the tail repeats it (same value to the same place), no branch
  const bool tall = wh > 4 * RP;                               // (uniform: a tile of 4 or 8 rows needs at most 15 window rows)
#pragma unroll
  for( int i = 0; i < NP; i++ )
  {
    const int rr = min( r0 + i * RP, last );
    R.v[i] = 0;
    if( ( i < 4 || tall ) && q < ndw ) R.v[i] = base[__umul24( (uint32_t) rr, sd ) + (uint32_t) q];
  }
}
template<int NT>
__device__ __forceinline__ void mc3_commit_luma( const Mc3Luma<NT>& R, pel_t* __restrict__ win, int x0, int ww, int wh, int tid )
{
  constexpr int RP = Mc3Luma<NT>::RP, NP = Mc3Luma<NT>::NP;
  const int q = tid & 15, r0 = tid >> 4;
  const int sh = ( x0 & 1 ) << 4;
  const int nst = ( ww + 1 ) >> 1;                    // dwords kept per row
  uint32_t* __restrict__ wdw = reinterpret_cast<uint32_t*>( win );
  const int last = wh - 1 - ( max( wh - 1 - r0, 0 ) % RP );
  const bool tall = wh > 4 * RP;
  uint32_t w[NP];
#pragma unroll
  for( int i = 0; i < NP; i++ ) w[i] = R.v[i];
  if( sh )        // (uniform: a window that starts at an even column is stored as it was loaded)
  {
#pragma unroll
    for( int i = 0; i < NP; i++ ) w[i] = __builtin_amdgcn_alignbit( mc_dpp_next_lane( R.v[i] ), R.v[i], 16 );
  }
  if( q < nst )
  {
#pragma unroll
    for( int i = 0; i < NP; i++ ) if( i < 4 || tall ) wdw[__umul24( (uint32_t) min( r0 + i * RP, last ), MC2_WST_L / 2 ) + (uint32_t) q] = w[i];
  }
}
template<int NT>
__device__ __forceinline__ void mc3_load_luma( pel_t* __restrict__ win, const pel_t* __restrict__ ref, int stride, int x0, int y0, int ww, int wh, int tid )
{
  Mc3Luma<NT> R;
  mc3_issue_luma<NT>( R, ref, stride, x0, y0, ww, wh, tid );
  mc3_commit_luma<NT>( R, win, x0, ww, wh, tid );
}
// Cb and Cr windows of a list (same geometry): 8 lanes per row, the rows of Cb, then those of Cr
template<int NT>
__device__ __forceinline__ void mc3_issue_chroma( Mc3Chroma<NT>& R, const pel_t* __restrict__ refCb, const pel_t* __restrict__ refCr, int stride, int x0, int y0, int ww, int wh, int tid )
{
  constexpr int RP = Mc3Chroma<NT>::RP, NP = Mc3Chroma<NT>::NP;
  const int q = tid & 7, r0 = tid >> 3;
  const int odd = x0 & 1;
  const int ndw = ( ww + 1 + odd ) >> 1;
  const size_t off = (size_t) y0 * stride + ( x0 - odd );
  const uint32_t sd = (uint32_t) stride >> 1;
  const int last = 2 * wh - 1 - ( max( 2 * wh - 1 - r0, 0 ) % RP );
#pragma unroll
  for( int i = 0; i < NP; i++ )
  {
    const int r2 = min( r0 + i * RP, last ), cc = r2 >= wh, r = r2 - ( cc ? wh : 0 );
    const uint32_t* __restrict__ base = reinterpret_cast<const uint32_t*>( ( cc ? refCr : refCb ) + off );
    R.v[i] = 0;
    if( q < ndw ) R.v[i] = base[__umul24( (uint32_t) r, sd ) + (uint32_t) q];
  }
}
template<int NT>
__device__ __forceinline__ void mc3_commit_chroma( const Mc3Chroma<NT>& R, pel_t* __restrict__ winCb /* Cr follows: 12 rows further */, int x0, int ww, int wh, int tid )
{
  constexpr int RP = Mc3Chroma<NT>::RP, NP = Mc3Chroma<NT>::NP;
  const int q = tid & 7, r0 = tid >> 3;
  const int sh = ( x0 & 1 ) << 4;
  const int nst = ( ww + 1 ) >> 1;
  const int last = 2 * wh - 1 - ( max( 2 * wh - 1 - r0, 0 ) % RP );
  uint32_t w[NP];
#pragma unroll
  for( int i = 0; i < NP; i++ ) w[i] = R.v[i];
  if( sh )
  {
#pragma unroll
    for( int i = 0; i < NP; i++ ) w[i] = __builtin_amdgcn_alignbit( mc_dpp_next_lane( R.v[i] ), R.v[i], 16 );
  }
  uint32_t* __restrict__ wdw = reinterpret_cast<uint32_t*>( winCb );
  if( q < nst )
  {
#pragma unroll
    for( int i = 0; i < NP; i++ )
    {
      const int r2 = min( r0 + i * RP, last ), cc = r2 >= wh, r = r2 - ( cc ? wh : 0 );
      wdw[( cc ? 12 * ( MC2_WST_C / 2 ) : 0 ) + r * ( MC2_WST_C / 2 ) + q] = w[i];
    }
  }
}
template<int NT>
__device__ __forceinline__ void mc3_load_chroma( pel_t* __restrict__ winCb, const pel_t* __restrict__ refCb, const pel_t* __restrict__ refCr, int stride, int x0, int y0, int ww, int wh, int tid )
{
  Mc3Chroma<NT> R;
  mc3_issue_chroma<NT>( R, refCb, refCr, stride, x0, y0, ww, wh, tid );
  mc3_commit_chroma<NT>( R, winCb, x0, ww, wh, tid );
}

// the windows of a segment record (k_mc's other tiles, k_mc_dmvr): the dword loader where the window lies inside what may be read and is no displaced copy, else sample by sample
__device__ __forceinline__ bool mc_seg_plain( const McSeg& g, int stride )
{
  const int odd = g.x0 & 1;
  const bool pl = ( g.wrapOff | g.padOff | g.shX | g.shY ) == 0 && g.cw == g.ww && g.chh == g.wh && g.x0 >= g.bx0 && g.y0 >= g.by0 && g.x0 + g.ww - 1 <= g.bx1 && g.y0 + g.wh - 1 <= g.by1
                  && g.x0 - odd + 2 * ( ( g.ww + 1 + odd ) >> 1 ) <= stride;
  return __builtin_amdgcn_readfirstlane( pl ) != 0;
}
template<int NT>
__device__ __forceinline__ void mc_load_seg_luma( pel_t* win, const McSeg& g, const pel_t* ref, int stride, int pw, int ph, int tid )
{
  if( mc_seg_plain( g, stride ) )
    mc3_load_luma<NT>( win, mc_uniform_ptr( ref ), stride, __builtin_amdgcn_readfirstlane( g.x0 ), __builtin_amdgcn_readfirstlane( g.y0 ), __builtin_amdgcn_readfirstlane( g.ww ), __builtin_amdgcn_readfirstlane( g.wh ), tid );
  else mc_load_window<NT>( win, MC2_WST_L, g, ref, stride, pw, ph, tid );
}
template<int NT>
__device__ __forceinline__ void mc_load_seg_chroma( pel_t* winCb, pel_t* winCr, const McSeg& gCb, const McSeg& gCr, const pel_t* refCb, const pel_t* refCr, int stride, int pw, int ph, int tid )
{
  if( mc_seg_plain( gCb, stride ) )
    mc3_load_chroma<NT>( winCb, mc_uniform_ptr( refCb ), mc_uniform_ptr( refCr ), stride, __builtin_amdgcn_readfirstlane( gCb.x0 ), __builtin_amdgcn_readfirstlane( gCb.y0 ), __builtin_amdgcn_readfirstlane( gCb.ww ),
                           __builtin_amdgcn_readfirstlane( gCb.wh ), tid );
  else { mc_load_window<NT>( winCb, MC2_WST_C, gCb, refCb, stride, pw, ph, tid ); mc_load_window<NT>( winCr, MC2_WST_C, gCr, refCr, stride, pw, ph, tid ); }
}
// ---- any window sample by sample (clamped / wrapped / displaced copies), in two steps like the dword loaders: 32 lanes per row
template<int NT, int ROWS> struct McWinS { static constexpr int RS = NT / 32, NP = ( ROWS + RS - 1 ) / RS; int v[NP]; };
template<int NT, int ROWS>
__device__ __forceinline__ void mcs_issue( McWinS<NT, ROWS>& R, const McSeg& g, const pel_t* __restrict__ ref, int stride, int pw, int tid )
{
  constexpr int RS = McWinS<NT, ROWS>::RS, NP = McWinS<NT, ROWS>::NP;
  const int col = min( tid & 31, g.ww - 1 ), row0 = tid >> 5;      // (lanes beyond the window read its last column and store nothing)
  const int sx = mc_ref_col( g.x0 + clip3( 0, g.cw - 1, col + g.shX - g.padOff ), g.bx0, g.bx1, pw, g.wrapOff );
  const int last = g.wh - 1 - ( max( g.wh - 1 - row0, 0 ) % RS );   // last row of this lane's residue class (the tail repeats it: same value to the same place)
#pragma unroll
  for( int i = 0; i < NP; i++ )
  {
    const int yy = min( row0 + i * RS, last );
    const int sy = clip3( g.by0, g.by1, g.y0 + clip3( 0, g.chh - 1, yy + g.shY - g.padOff ) );
    R.v[i] = ref[__umul24( (uint32_t) sy, (uint32_t) stride ) + (uint32_t) sx];      // (clamped coordinates: inside the plane, whose rows and columns number fewer than 2^24)
  }
}
template<int NT, int ROWS>
__device__ __forceinline__ void mcs_commit( const McWinS<NT, ROWS>& R, pel_t* win, int wst, const McSeg& g, int tid )
{
  constexpr int RS = McWinS<NT, ROWS>::RS, NP = McWinS<NT, ROWS>::NP;
  const int col = tid & 31, row0 = tid >> 5;
  const int last = g.wh - 1 - ( max( g.wh - 1 - row0, 0 ) % RS );
  if( col < g.ww )
  {
#pragma unroll
    for( int i = 0; i < NP; i++ ) win[min( row0 + i * RS, last ) * wst + col] = (pel_t) R.v[i];
  }
}
// the luma windows of both lists with one wait (k_mc_dmvr: the bilinear windows)
template<int NT, int ROWS>
__device__ __forceinline__ void mc_load_seg_luma_pair( Mc2Shared& m, int stride, int pw, int tid )
{
  const bool plain = mc_seg_plain( m.seg[0][0], stride ) && mc_seg_plain( m.seg[1][0], stride );
  if( plain )
  {
    Mc3Luma<NT> R[2];
#pragma unroll
    for( int l = 0; l < 2; l++ )
    {
      const McSeg& g = m.seg[l][0];
      mc3_issue_luma<NT>( R[l], mc_uniform_ptr( m.refp[l][0] ), stride, __builtin_amdgcn_readfirstlane( g.x0 ), __builtin_amdgcn_readfirstlane( g.y0 ), __builtin_amdgcn_readfirstlane( g.ww ), __builtin_amdgcn_readfirstlane( g.wh ), tid );
    }
#pragma unroll
    for( int l = 0; l < 2; l++ )
    {
      const McSeg& g = m.seg[l][0];
      mc3_commit_luma<NT>( R[l], m.winL[l], __builtin_amdgcn_readfirstlane( g.x0 ), __builtin_amdgcn_readfirstlane( g.ww ), __builtin_amdgcn_readfirstlane( g.wh ), tid );
    }
  }
  else
  {
    McWinS<NT, ROWS> R[2];
#pragma unroll
    for( int l = 0; l < 2; l++ ) mcs_issue<NT, ROWS>( R[l], m.seg[l][0], m.refp[l][0], stride, pw, tid );
#pragma unroll
    for( int l = 0; l < 2; l++ ) mcs_commit<NT, ROWS>( R[l], m.winL[l], MC2_WST_L, m.seg[l][0], tid );
  }
}
// all windows of a bi-predicted tile with one wait (k_mc_dmvr: the final prediction)
template<int NT>
__device__ __forceinline__ void mc_load_seg_all( Mc2Shared& m, int ncomp, int strideL, int strideC, int pwL, int pwC, int tid )
{
  bool plain = mc_seg_plain( m.seg[0][0], strideL ) && mc_seg_plain( m.seg[1][0], strideL );
  if( ncomp == 3 ) plain = plain && mc_seg_plain( m.seg[0][1], strideC ) && mc_seg_plain( m.seg[1][1], strideC );
  if( plain )
  {
    Mc3Luma<NT> RL[2]; Mc3Chroma<NT> RC[2];
#pragma unroll
    for( int l = 0; l < 2; l++ )
    {
      const McSeg& g = m.seg[l][0];
      mc3_issue_luma<NT>( RL[l], mc_uniform_ptr( m.refp[l][0] ), strideL, __builtin_amdgcn_readfirstlane( g.x0 ), __builtin_amdgcn_readfirstlane( g.y0 ), __builtin_amdgcn_readfirstlane( g.ww ), __builtin_amdgcn_readfirstlane( g.wh ), tid );
      if( ncomp == 3 )
      {
        const McSeg& c = m.seg[l][1];
        mc3_issue_chroma<NT>( RC[l], mc_uniform_ptr( m.refp[l][1] ), mc_uniform_ptr( m.refp[l][2] ), strideC, __builtin_amdgcn_readfirstlane( c.x0 ), __builtin_amdgcn_readfirstlane( c.y0 ), __builtin_amdgcn_readfirstlane( c.ww ), __builtin_amdgcn_readfirstlane( c.wh ), tid );
      }
    }
#pragma unroll
    for( int l = 0; l < 2; l++ )
    {
      const McSeg& g = m.seg[l][0];
      mc3_commit_luma<NT>( RL[l], m.winL[l], __builtin_amdgcn_readfirstlane( g.x0 ), __builtin_amdgcn_readfirstlane( g.ww ), __builtin_amdgcn_readfirstlane( g.wh ), tid );
      if( ncomp == 3 )
      {
        const McSeg& c = m.seg[l][1];
        mc3_commit_chroma<NT>( RC[l], m.winC[l][0], __builtin_amdgcn_readfirstlane( c.x0 ), __builtin_amdgcn_readfirstlane( c.ww ), __builtin_amdgcn_readfirstlane( c.wh ), tid );
      }
    }
  }
  else
  {
    // (two waits - luma, chroma: all six windows in flight at once cost the registers of one resident sub-block in sixteen)
    {
      McWinS<NT, 23> RL[2];
#pragma unroll
      for( int l = 0; l < 2; l++ ) mcs_issue<NT, 23>( RL[l], m.seg[l][0], m.refp[l][0], strideL, pwL, tid );
#pragma unroll
      for( int l = 0; l < 2; l++ ) mcs_commit<NT, 23>( RL[l], m.winL[l], MC2_WST_L, m.seg[l][0], tid );
    }
    if( ncomp == 3 )
    {
      McWinS<NT, 11> RC[2][2];
#pragma unroll
      for( int l = 0; l < 2; l++ ) { mcs_issue<NT, 11>( RC[l][0], m.seg[l][1], m.refp[l][1], strideC, pwC, tid ); mcs_issue<NT, 11>( RC[l][1], m.seg[l][2], m.refp[l][2], strideC, pwC, tid ); }
#pragma unroll
      for( int l = 0; l < 2; l++ ) { mcs_commit<NT, 11>( RC[l][0], m.winC[l][0], MC2_WST_C, m.seg[l][1], tid ); mcs_commit<NT, 11>( RC[l][1], m.winC[l][1], MC2_WST_C, m.seg[l][2], tid ); }
    }
  }
}

// The tap rows of the two stages, read before the windows are waited for (one memory round trip for everything a tile reads before it computes): what lane
// `tid` needs in stage 1 (s1[0]: the merged pass or the luma pass, s1[1]: the chroma pass) and in stage 2 (s2[list]).  Same lane -> work item mapping as the stages.
struct McTapsPre { uint4 s1[2], s2[2]; };
template<int NT, int PARTS = 3 /* 1: stage 1, 2: stage 2 */>
__device__ __forceinline__ void mc2_prefetch_taps( McTapsPre& P, int nl, int ncomp, int tw, int th, int tid, const McTapRows& T )
{
  const int h0L = T.h[0][0], h1L = T.h[1][0], h0C = T.h[0][1], h1C = T.h[1][1], v0L = T.v[0][0], v1L = T.v[1][0], v0C = T.v[0][1], v1C = T.v[1][1];
  const int lgU = tw > 8 ? 1 : 0, perL = ( ( th + 8 ) >> 1 ) << lgU, nA = nl * perL;
  const int rpCU = ( ( th >> 1 ) + 4 ) >> 1, perC = 2 * rpCU, nB = ncomp == 3 ? nl * perC : 0;
  const bool merged = nA + nB <= NT;
  const int rowL = tid >= perL ? h1L : h0L;
  const int rowA = ( merged && tid >= nA ) ? ( tid - nA >= perC ? h1C : h0C ) : rowL;
  // (no branches: a load that may not happen would have to be waited for where the paths meet; rows nobody uses are row 0)
  if constexpr( PARTS & 1 )
  {
    P.s1[0] = d_mcTaps[rowA];
    P.s1[1] = d_mcTaps[tid >= perC ? h1C : h0C];
  }
  if constexpr( PARTS & 2 )
  {
    const bool isC = tid >= tw * ( ( th + 7 ) >> 3 );
    P.s2[0] = d_mcTaps[isC ? v0C : v0L];
    P.s2[1] = d_mcTaps[isC ? v1C : v1L];
  }
}

// ---- stage 1: horizontal filter of every window row to 14-bit intermediates, two rows x eight columns per work item --------------
// (a tile is at most 16x16: every pass is one work item per lane at most - 48 luma items, 24 chroma items)
template<int NT>
__device__ __forceinline__ void mc2_stage1( Mc2Shared& m, int nl, int ncomp, int tw, int th, int headroom, int tid, const McTapRows& T, const McTapsPre& pre )
{
  static_assert( NT >= 64, "one work item per lane and pass" );
  const int shift1 = 6 - headroom, offset1 = __builtin_amdgcn_readfirstlane( -IF_INTERNAL_OFFS * ( 1 << shift1 ) );
  const int h0L = T.h[0][0], h1L = T.h[1][0], h0C = T.h[0][1], h1C = T.h[1][1];      // (values first: a per-lane choice between struct members would put the struct into scratch)
  // When the luma and the chroma work items of the tile fit into one pass of the workgroup (one prediction list, small tiles, two wavefronts per tile), the
  // chroma row pairs go through the eight-tap code beside the luma ones (taps 4..7 of a chroma row of d_mcTaps are zero, a chroma window row holds 16 samples)
  const int lgU = tw > 8 ? 1 : 0, perL = ( ( th + 8 ) >> 1 ) << lgU, nA = nl * perL;      // a 16-wide tile has two groups of eight columns; th + 7 window rows in pairs
  const int rpCU = ( ( th >> 1 ) + 4 ) >> 1, perC = 2 * rpCU, nB = ncomp == 3 ? nl * perC : 0;      // th / 2 + 3 window rows in pairs, Cb and Cr
  if( nA + nB <= NT )
  {
    if( tid < nA + nB )
    {
      const bool isC = tid >= nA;
      int k; const pel_t* src; uint32_t* dst;
      if( !isC ) { k = tid >= perL; const int r0 = tid - ( k ? perL : 0 ), rp = r0 >> lgU, g = r0 & lgU; src = &m.winL[k][2 * rp * MC2_WST_L + 8 * g]; dst = &m.tmpL[k][8 * g * MC3_TPL + rp]; }
      else { int q = tid - nA; k = q >= perC; q -= k ? perC : 0; const int cc = q >= rpCU, rp = q - ( cc ? rpCU : 0 ); src = &m.winC[k][cc][2 * rp * MC2_WST_C]; dst = &m.tmpC[2 * k + cc][rp]; }
      const int tapRow = isC ? ( k ? h1C : h0C ) : ( k ? h1L : h0L );
      const int ws = isC ? MC2_WST_C : MC2_WST_L, cs = isC ? MC3_TPC : MC3_TPL;
      const uint4 C = pre.s1[0];
      int a[8], b[8];
      mc_fir8<8>( *reinterpret_cast<const uint4*>( src ), *reinterpret_cast<const uint4*>( src + 8 ), C, offset1, a );
      mc_fir8<8>( *reinterpret_cast<const uint4*>( src + ws ), *reinterpret_cast<const uint4*>( src + ws + 8 ), C, offset1, b );
#pragma unroll
      for( int j = 0; j < 8; j++ ) dst[j * cs] = __builtin_amdgcn_perm( (uint32_t) ( b[j] >> shift1 ), (uint32_t) ( a[j] >> shift1 ), 0x05040100u );
    }
    __syncthreads();
    return;
  }
  if( tid < nA )
  {
    const int k = tid >= perL, r0 = tid - ( k ? perL : 0 ), rp = r0 >> lgU, g = r0 & lgU;
    const uint4 C = pre.s1[0];
    const pel_t* src = &m.winL[k][2 * rp * MC2_WST_L + 8 * g];
    int a[8], b[8];
    mc_fir8<8>( *reinterpret_cast<const uint4*>( src ), *reinterpret_cast<const uint4*>( src + 8 ), C, offset1, a );
    mc_fir8<8>( *reinterpret_cast<const uint4*>( src + MC2_WST_L ), *reinterpret_cast<const uint4*>( src + MC2_WST_L + 8 ), C, offset1, b );
    uint32_t* dst = &m.tmpL[k][8 * g * MC3_TPL + rp];
#pragma unroll
    for( int j = 0; j < 8; j++ ) dst[j * MC3_TPL] = __builtin_amdgcn_perm( (uint32_t) ( b[j] >> shift1 ), (uint32_t) ( a[j] >> shift1 ), 0x05040100u );
  }
  if( tid < nB )
  {
    const int k = tid >= perC, q = tid - ( k ? perC : 0 ), cc = q >= rpCU, rp = q - ( cc ? rpCU : 0 );
    const uint4 C = pre.s1[1];
    const pel_t* src = &m.winC[k][cc][2 * rp * MC2_WST_C];
    int a[8], b[8];
    mc_fir8<4>( *reinterpret_cast<const uint4*>( src ), *reinterpret_cast<const uint4*>( src + 8 ), C, offset1, a );
    mc_fir8<4>( *reinterpret_cast<const uint4*>( src + MC2_WST_C ), *reinterpret_cast<const uint4*>( src + MC2_WST_C + 8 ), C, offset1, b );
    uint32_t* dst = &m.tmpC[2 * k + cc][rp];
#pragma unroll
    for( int j = 0; j < 8; j++ ) dst[j * MC3_TPC] = __builtin_amdgcn_perm( (uint32_t) ( b[j] >> shift1 ), (uint32_t) ( a[j] >> shift1 ), 0x05040100u );
  }
  __syncthreads();
}

// how the predictions of a tile become samples (decided once per tile)
enum { MCM_UNI = 0, MCM_AVG, MCM_BCW, MCM_GEO, MCM_WP_UNI, MCM_WP_BI };

// ---- stage 2: vertical filter, 8 rows of one column per work item, both lists in the same lane, then the combination
//      (AreaBuf::addAvg / addWeightedAvg, Buffer.cpp:441,372; GPM weights, InterpolationFilter.cpp:1217) or the BDOF input.
// bsp: BDOF buffers (the 14-bit luma predictions go there instead of being averaged) or nullptr
template<int NT>
__device__ __forceinline__ void mc2_stage2( Mc2Shared& m, BdofShared* bsp, int nl, int ncomp, int mode, const vvr_cu& cu, int bcwIdx, int bd, int headroom,
                                            const DevPlanes& reco, int tx, int ty, int tw, int th, int tid, const int16_t* __restrict__ fwdLut /* LMCS forward map or nullptr */,
                                            const McTapRows& T, const vvr_wp_params* __restrict__ wp /* MCM_WP_*: explicit weighted prediction */, int wpL, int wpR0, int wpR1, const McTapsPre& pre )
{
  static_assert( NT >= 64, "one work item per lane" );
  const int wL = tw, hL = th, wC = tw >> 1, hC = th >> 1;
  const int lwL = wL == 16 ? 4 : wL == 8 ? 3 : 2;
  const int itemsL = wL * ( ( hL + 7 ) >> 3 ), itemsC = ncomp == 3 ? 2 * wC : 0;            // chroma: at most 8 rows = one group
  const bool full = hL == 16;                          // every work item has eight rows (luma 2 x 8, chroma 8)
  // (uniform values in scalar registers: a per-lane choice between kernel arguments would become a load from the argument segment)
  const uint32_t st0 = (uint32_t) __builtin_amdgcn_readfirstlane( reco.stride[0] ), st1 = (uint32_t) __builtin_amdgcn_readfirstlane( reco.stride[1] );
  pel_t* const p0 = mc_uniform_ptr( reco.p[0] ); pel_t* const p1 = mc_uniform_ptr( reco.p[1] ); pel_t* const p2 = mc_uniform_ptr( reco.p[2] );
  const int16_t* __restrict__ const fwdU = mc_uniform_ptr( fwdLut );
  const int shift2 = 6 + headroom, offset2 = ( 1 << ( shift2 - 1 ) ) + ( IF_INTERNAL_OFFS << 6 );
  const int init2 = __builtin_amdgcn_readfirstlane( mode == MCM_UNI ? offset2 : 0 );
  const int idx = tid;                                 // (32 luma + 16 chroma work items at most)
  if( idx < itemsL + itemsC )
  {
    int c, x, g8;
    if( idx < itemsL ) { c = 0; g8 = idx >> lwL; x = idx & ( wL - 1 ); } else { const int q = idx - itemsL; c = 1 + ( q >= wC ); x = q - ( c - 1 ) * wC; g8 = 0; }
    const int cs = c ? 1 : 0, hh = c ? hC : hL, nrows = min( 8, hh - 8 * g8 );
    int p[2][8];
#pragma unroll
    for( int k = 0; k < 2; k++ )
    {
      if( k < nl )
      {
        const uint32_t* src = c ? &m.tmpC[2 * k + c - 1][x * MC3_TPC] : &m.tmpL[k][x * MC3_TPL + 4 * g8];
        const uint4 C = pre.s2[k];         // (chroma: taps 4..7 are zero)
        mc_fir8<8>( *reinterpret_cast<const uint4*>( src ), *reinterpret_cast<const uint4*>( src + 4 ), C, init2, p[k] );
      }
    }
    if( bsp && c == 0 )
    {
#pragma unroll
      for( int i = 0; i < 8; i++ )
        if( i < nrows ) { bsp->blk[0][BDOF_AT( x, 8 * g8 + i )] = (pel_t) ( p[0][i] >> 6 ); bsp->blk[1][BDOF_AT( x, 8 * g8 + i )] = (pel_t) ( p[1][i] >> 6 ); }
      return;
    }
    int out[8];
    if( mode == MCM_UNI )
    {
#pragma unroll
      for( int i = 0; i < 8; i++ ) out[i] = clip_pel( p[0][i] >> shift2, bd );
    }
    else if( mode == MCM_AVG )
    {
      const int shift = headroom + 1, offset = ( 1 << ( shift - 1 ) ) + 2 * IF_INTERNAL_OFFS;
#pragma unroll
      for( int i = 0; i < 8; i++ ) out[i] = clip_pel( ( ( p[0][i] >> 6 ) + ( p[1][i] >> 6 ) + offset ) >> shift, bd );
    }
    else if( mode == MCM_BCW )
    {
      const int w1 = d_bcw_weights[bcwIdx], w0 = 8 - w1, shift = headroom + 3, offset = ( 1 << ( shift - 1 ) ) + ( IF_INTERNAL_OFFS << 3 );
#pragma unroll
      for( int i = 0; i < 8; i++ ) out[i] = clip_pel( ( ( p[0][i] >> 6 ) * w0 + ( p[1][i] >> 6 ) * w1 + offset ) >> shift, bd );
    }
    else if( mode == MCM_GEO )
    {
      // GPM: weight of partition 0 from the mask tables, addressed in luma units relative to the CU with the mirroring of the split angle
      const int MS = 112;
      const int angle = d_geo_params[cu.geo_split_dir][0];
      const int wIdx = ilog2( cu.w ) - 3, hIdx = ilog2( cu.h ) - 3;
      const int ox = d_geo_weight_offset[cu.geo_split_dir][hIdx][wIdx][0], oy = d_geo_weight_offset[cu.geo_split_dir][hIdx][wIdx][1];
      const int8_t* gW = d_geo_weights[d_geo_angle2mask[angle]];
      const int mir = d_geo_angle2mirror[angle];
      const int lx = ( ( ( tx >> cs ) + x ) << cs ) - cu.x, ly0 = ( ( ( ty >> cs ) + 8 * g8 ) << cs ) - cu.y;
      int gBase, gSY;
      if( mir == 2 )      { gBase = ( MS - 1 - oy - ly0 ) * MS + ox + lx; gSY = -( MS << cs ); }
      else if( mir == 1 ) { gBase = ( oy + ly0 ) * MS + ( MS - 1 - ox ) - lx; gSY = MS << cs; }
      else                { gBase = ( oy + ly0 ) * MS + ox + lx; gSY = MS << cs; }
      const int shift = headroom + 3, offset = ( 1 << ( shift - 1 ) ) + ( IF_INTERNAL_OFFS << 3 );
#pragma unroll
      for( int i = 0; i < 8; i++ )
      {
        const int wt = gW[gBase + min( i, nrows - 1 ) * gSY];
        out[i] = clip_pel( ( wt * ( p[0][i] >> 6 ) + ( 8 - wt ) * ( p[1][i] >> 6 ) + offset ) >> shift, bd );
      }
    }
    else if( mode == MCM_WP_UNI )
    {
#pragma unroll
      for( int i = 0; i < 8; i++ ) out[i] = wp_uni( wp, wpL, wpL ? wpR1 : wpR0, c, p[0][i] >> 6, bd, headroom );
    }
    else
    {
#pragma unroll
      for( int i = 0; i < 8; i++ ) out[i] = wp_bi( wp, wpR0, wpR1, c, p[0][i] >> 6, p[1][i] >> 6, bd, headroom );
    }
    if( fwdU && c == 0 )
    {
      // LMCS: luma predictions are stored forward-mapped
#pragma unroll
      for( int i = 0; i < 8; i++ ) out[i] = fwdU[(uint32_t) out[i] & 0xfffu];
    }
    const uint32_t st = c ? st1 : st0;
    pel_t* const dstp = c == 0 ? p0 : c == 1 ? p1 : p2;
    const uint32_t o0 = __umul24( (uint32_t) ( ( ty >> cs ) + 8 * g8 ), st ) + (uint32_t) ( ( tx >> cs ) + x );      // (a plane has fewer than 2^24 rows and columns; the product needs 32 bits: mul24 keeps the low 32)
    if( full )
    {
#pragma unroll
      for( int i = 0; i < 8; i++ ) dstp[o0 + (uint32_t) i * st] = (pel_t) out[i];
    }
    else
    {
#pragma unroll
      for( int i = 0; i < 8; i++ ) if( i < nrows ) dstp[o0 + (uint32_t) i * st] = (pel_t) out[i];
    }
  }
}

// BDOF = true: the launch holds only tiles of CUs in BDOF mode (they need 3.5 KB more LDS for the gradient buffers; keeping them out
// of the plain launch raises the number of resident tiles per CU there).
template<int NT, bool BDOF>
__global__ __launch_bounds__( NT ) void k_mc( PicDev pic, RefSet refs, DevPlanes reco, const McItem* __restrict__ items, int numItems, const McItem* __restrict__ items2, int numItems2 )
{
  __shared__ Mc2Shared m;
  __shared__ typename std::conditional<BDOF, BdofShared, int>::type bs;
  const int item = mc_item_index();
  if( item >= numItems + numItems2 ) return;
  const McItem it = item < numItems ? items[item] : items2[item - numItems];       // (tiles the host wrote - SbTMVP -, then the tiles k_prep wrote)
  const int tid = threadIdx.x;
  // the tile record is self-contained (motion of the CU, or of the 8x8 sub-block for SbTMVP, with the identical-motion shortcut already
  // decided, xCheckIdenticalMotion :404); only GPM tiles read their CU (split direction, the two uni-directional motions).  The record is the same
  // for every lane: its fields go to scalar registers, and so does everything derived from them
#define MC_U( v ) __builtin_amdgcn_readfirstlane( (int) ( v ) )
  const int ix = MC_U( it.x ), iy = MC_U( it.y ), iw = MC_U( it.w ), ih = MC_U( it.h ), iflags = MC_U( it.flags ), bcwIdx = MC_U( it.bcw );
  const int mRef[2] = { MC_U( it.ref[0] ), MC_U( it.ref[1] ) };
  const int clipX = MC_U( it.clipX ), clipY = MC_U( it.clipY ), clipW4 = MC_U( it.clipW4 );
  const vvr_cu& cu = pic.cu[MC_U( it.cu )];               // (dereferenced for GPM tiles only)
  const int16_t* __restrict__ fwdLut = lmcs_fwd_at( pic, ix, iy );      // LMCS (where the tile's slice uses it): luma predictions are stored forward-mapped
  const int bd = pic.hdr.bit_depth;
  const bool geo = ( iflags & MC_ITEM_GEO ) != 0;         // motionCompensationGeo (:1461): two uni-predictions kept at 14 bit, blended with the GPM masks
  const bool uni = ( iflags & MC_ITEM_UNI ) != 0;
  const bool altHpel = ( iflags & MC_ITEM_HPEL ) != 0;
  const bool biPred = mRef[0] >= 0 && mRef[1] >= 0;
  const int ncomp = pic.hdr.chroma_format ? 3 : 1;
  const int l0 = uni ? ( ( biPred || mRef[0] >= 0 ) ? 0 : 1 ) : 0;
  const int nl = uni ? 1 : 2;
  const int headroom = 14 - bd > 2 ? 14 - bd : 2;
  const bool f4 = iw == 4 && ih == 4;
  McTapRows T = {};
  // ---- a tile whose windows lie inside the picture (no wrap-around, no sub-picture of its own): geometry in scalar registers, dword loads.
  // The window always spans the full filter support (the block's integer origin sits at (half, half)), whatever the fractional part of the MV
  bool fast = !pic.hdr.wrap_offset && !pic.subpics;
  int wx[2], wy[2], cx[2], cy[2], li[2], ri[2];
  McTapsPre pre;
  if( fast )
  {
#pragma unroll
    for( int k = 0; k < 2; k++ )
    {
      if( k < nl )
      {
        const int l = geo ? ( MC_U( cu.geo_dir_ref[k] ) >> 4 ) - 1 : uni ? l0 : k;
        const int refIdx = geo ? ( MC_U( cu.geo_dir_ref[k] ) & 15 ) : ( l ? mRef[1] : mRef[0] );
        int mvx = geo ? MC_U( cu.geo_mv[k][0] ) : MC_U( l ? it.mv[1][0] : it.mv[0][0] ), mvy = geo ? MC_U( cu.geo_mv[k][1] ) : MC_U( l ? it.mv[1][1] : it.mv[0][1] );
        const McBounds B = { 0, 0, (int) pic.hdr.width - 1, (int) pic.hdr.height - 1 };
        mc_clip_mv( pic, B, clipX, clipY, mvx, mvy );       // clipped with the position of m_currCuArea (InterPrediction.cpp:651-656)
        li[k] = l; ri[k] = refIdx;
        wx[k] = ix + ( mvx >> 4 ) - 3; wy[k] = iy + ( mvy >> 4 ) - 3;
        cx[k] = ( ix >> 1 ) + ( mvx >> 5 ) - 1; cy[k] = ( iy >> 1 ) + ( mvy >> 5 ) - 1;
        T.h[k][0] = mc_tap_row( 0, mvx & 15, f4, altHpel ); T.v[k][0] = mc_tap_row( 0, mvy & 15, f4, altHpel );
        T.h[k][1] = mc_tap_row( 1, mvx & 31, false, false ); T.v[k][1] = mc_tap_row( 1, mvy & 31, false, false );
        // (the dword loads read up to one sample beyond the window's last column: that one must lie in the row's padded stride)
        const int oddL = wx[k] & 1, oddC = cx[k] & 1;
        fast = fast && wx[k] >= 0 && wy[k] >= 0 && wx[k] + iw + 7 <= reco.w[0] && wy[k] + ih + 7 <= reco.h[0] && wx[k] - oddL + 2 * ( ( iw + 8 + oddL ) >> 1 ) <= reco.stride[0];
        if( ncomp == 3 )
          fast = fast && cx[k] >= 0 && cy[k] >= 0 && cx[k] + ( iw >> 1 ) + 3 <= reco.w[1] && cy[k] + ( ih >> 1 ) + 3 <= reco.h[1] && cx[k] - oddC + 2 * ( ( ( iw >> 1 ) + 4 + oddC ) >> 1 ) <= reco.stride[1];
      }
    }
  }
  if( fast )
  {
    // every load of the tile's windows is in flight before the first of them is waited for, and so are the tap rows of both stages
    Mc3Luma<NT> RL[2]; Mc3Chroma<NT> RC[2];
    mc2_prefetch_taps<NT>( pre, nl, ncomp, iw, ih, tid, T );
#pragma unroll
    for( int k = 0; k < 2; k++ )
    {
      if( k < nl )
      {
        const int ridx = li[k] * VVR_MAX_REFS + ri[k];
        mc3_issue_luma<NT>( RL[k], refs.p[ridx][0], reco.stride[0], wx[k], wy[k], iw + 7, ih + 7, tid );
        if( ncomp == 3 ) mc3_issue_chroma<NT>( RC[k], refs.p[ridx][1], refs.p[ridx][2], reco.stride[1], cx[k], cy[k], ( iw >> 1 ) + 3, ( ih >> 1 ) + 3, tid );
      }
    }
#pragma unroll
    for( int k = 0; k < 2; k++ )
    {
      if( k < nl )
      {
        mc3_commit_luma<NT>( RL[k], m.winL[k], wx[k], iw + 7, ih + 7, tid );
        if( ncomp == 3 ) mc3_commit_chroma<NT>( RC[k], m.winC[k][0], cx[k], ( iw >> 1 ) + 3, ( ih >> 1 ) + 3, tid );
      }
    }
    if constexpr( BDOF )
    {
      // (the BDOF border reads the fractional parts of the luma segments)
      if( tid < 2 ) { m.seg[tid][0].xFrac = ( tid ? T.h[1][0] : T.h[0][0] ) & 15; m.seg[tid][0].yFrac = ( tid ? T.v[1][0] : T.v[0][0] ) & 15; m.seg[tid][0].ox = m.seg[tid][0].oy = 3; }
    }
  }
  else
  {
    // ---- any other tile: segment geometry by six lanes, shared through LDS; windows sample by sample with clamped / wrapped coordinates
    if( tid < 6 )
    {
      const int k = tid / 3, c = tid - 3 * k;
      if( k < nl && c < ncomp )
      {
        const int l = geo ? ( cu.geo_dir_ref[k] >> 4 ) - 1 : uni ? l0 : k;
        const int refIdx = geo ? ( cu.geo_dir_ref[k] & 15 ) : ( l ? mRef[1] : mRef[0] );
        int mvx = geo ? cu.geo_mv[k][0] : ( l ? it.mv[1][0] : it.mv[0][0] ), mvy = geo ? cu.geo_mv[k][1] : ( l ? it.mv[1][1] : it.mv[0][1] );
        const McBounds B = mc_bounds( pic, clipX, clipY );
        // clipped with the position and size of m_currCuArea (InterPrediction.cpp:651-656): the CU, or the piece of an SbTMVP CU that xSubPuMC predicts as one block (:514-543)
        const int wrapOff = mc_clip_mv_w( pic, B, clipX, clipY, pic.hdr.wrap_offset ? ( clipW4 ? 4 * clipW4 : (int) cu.w ) : 0, mvx, mvy );
        McSeg g;
        const int cs = c ? 1 : 0, shf = 4 + cs, ntaps = c ? 4 : 8, half = ntaps / 2 - 1;
        g.wrapOff = wrapOff >> cs; g.bx0 = B.x0 >> cs; g.by0 = B.y0 >> cs; g.bx1 = B.x1 >> cs; g.by1 = B.y1 >> cs;
        g.w = iw >> cs; g.h = ih >> cs;
        g.xFrac = mvx & ( ( 1 << shf ) - 1 ); g.yFrac = mvy & ( ( 1 << shf ) - 1 );
        g.ox = half; g.oy = half;
        g.ww = g.w + ntaps - 1; g.wh = g.h + ntaps - 1;
        g.x0 = ( ix >> cs ) + ( mvx >> shf ) - half;
        g.y0 = ( iy >> cs ) + ( mvy >> shf ) - half;
        g.padOff = 0; g.cw = g.ww; g.chh = g.wh; g.shX = g.shY = 0;
        m.seg[k][c] = g;
        m.refp[k][c] = refs.p[l * VVR_MAX_REFS + refIdx][c];
      }
    }
    __syncthreads();
#pragma unroll
    for( int k = 0; k < 2; k++ )
    {
      if( k >= nl ) break;
      mc_load_seg_luma<NT>( m.winL[k], m.seg[k][0], m.refp[k][0], reco.stride[0], reco.w[0], reco.h[0], tid );
      if( ncomp == 3 ) mc_load_seg_chroma<NT>( m.winC[k][0], m.winC[k][1], m.seg[k][1], m.seg[k][2], m.refp[k][1], m.refp[k][2], reco.stride[1], reco.w[1], reco.h[1], tid );
      T.h[k][0] = mc_tap_row( 0, MC_U( m.seg[k][0].xFrac ), f4, altHpel ); T.v[k][0] = mc_tap_row( 0, MC_U( m.seg[k][0].yFrac ), f4, altHpel );
      if( ncomp == 3 ) { T.h[k][1] = mc_tap_row( 1, MC_U( m.seg[k][1].xFrac ), false, false ); T.v[k][1] = mc_tap_row( 1, MC_U( m.seg[k][1].yFrac ), false, false ); }
    }
    mc2_prefetch_taps<NT>( pre, nl, ncomp, iw, ih, tid, T );
  }
  __syncthreads();
  mc2_stage1<NT>( m, nl, ncomp, iw, ih, headroom, tid, T, pre );
  const vvr_wp_params* __restrict__ wpT = wp_at( pic, ix, iy );      // the weight table of the tile's slice
  const bool wpOn = !BDOF && wpT && !geo && bcwIdx == 2;          // xPredInterBi (:707,735-742)
  const int mode = wpOn ? ( uni ? MCM_WP_UNI : MCM_WP_BI ) : uni ? MCM_UNI : geo ? MCM_GEO : bcwIdx != 2 ? MCM_BCW : MCM_AVG;
  mc2_stage2<NT>( m, BDOF ? reinterpret_cast<BdofShared*>( &bs ) : nullptr, nl, ncomp, mode, cu, bcwIdx, bd, headroom, reco, ix, iy, iw, ih, tid, fwdLut, T,
                  wpOn ? wpT : nullptr, l0, mRef[0], mRef[1], pre );
  if constexpr( BDOF )
  {
    __syncthreads();
    mc_bdof_luma<NT>( bs, m.winL[0], m.winL[1], MC2_WST_L, &m.seg[0][0], 3, bd, reco, ix, iy, iw, ih, tid, fwdLut );
  }
}

// =====================================================================================================================
// k_mc_dmvr — decoder-side motion vector refinement + final prediction of one <= 16x16 sub-block (one wavefront).
//   InterPrediction::xProcessDMVR (InterPrediction.cpp:1847): xinitMC (:1804) bilinear prediction at 10 bit, SAD at the centre
//   (RdCost::xGetSAD8/16, every second row), early out, 25-point mirrored integer search (xBIPMVRefine :1702), parametric
//   sub-sample refinement (xDMVRSubPixelErrorSurface :1785, xSubPelErrorSrfc :1647, div_for_maxq7 :1612), then the final 8/4-tap
//   prediction from the padded local copy when the integer MV moved (xPrefetchPad :1525, xFinalPaddedMCForDMVR :1731) and
//   the average or BDOF (bioAppliedSubblk :1984).  The delta MV goes to dmvrOut[cu.dmvr_off + sub-block] for the host
//   (DecCu::TaskFinishMotionInfo, DecCu.cpp:161).
// =====================================================================================================================
#define DM_WST_L MC2_WST_L
struct DmvrShared {
  Mc2Shared m;                   // windows / intermediates / taps of the final prediction (stage 1 reuses winL for the bilinear windows, and the intermediates' place -
                                 // tmpL + tmpC, 2688 bytes - for the bilinear predictions of the extended sub-block: 2 x 20 x 20 samples at 10 bit, two per dword in the search)
  unsigned sad[25];
  int dmv[2], bioSub, minCost;
  BdofShared bs;
};

// quotient of the parametric error-surface minimum, in 1/16 sample: N / ( 2 * D ) truncated to three bits, sign restored (the result lies in
// -7 .. 7; xSubPelErrorSrfc, InterPrediction.cpp:1647-1700): three steps of a restoring division against 8 D, 4 D, 2 D
__device__ __forceinline__ int dmvr_div_for_maxq7( long long N, long long D )
{
  const bool neg = N < 0;
  if( neg ) N = -N;
  int q = 0;
  for( int b = 3; b >= 1; b-- )
  {
    q <<= 1;
    if( N >= ( D << b ) ) { N -= D << b; q |= 1; }
  }
  return neg ? -q : q;
}

// (16 sub-blocks per compute unit: 64 vector registers and 10 KB of LDS at most - a 4K B picture's 3700 sub-blocks are resident at once)
template<int NT>
__global__ __launch_bounds__( NT ) __attribute__( ( amdgpu_waves_per_eu( 8, 8 ) ) ) void k_mc_dmvr( PicDev pic, RefSet refs, DevPlanes reco, const McItem* __restrict__ items, int numItems, int32_t* __restrict__ dmvrOut )
{
  __shared__ DmvrShared sh;
  const int item = mc_item_index();
  if( item >= numItems ) return;
  const McItem it = items[item];
  const int16_t* __restrict__ fwdLut = lmcs_fwd_at( pic, it.x, it.y );      // LMCS (where the tile's slice uses it): luma predictions are stored forward-mapped (lmcs_fwd_luma)
  const vvr_cu& cu = pic.cu[it.cu];
  const int bd = pic.hdr.bit_depth;
  const int tid = threadIdx.x;
  const int ncomp = pic.hdr.chroma_format ? 3 : 1;
  const int w = it.w, h = it.h;
  const bool bio = cu.mc_mode == VVR_MC_DMVR_BDOF;
  pel_t ( * const bil )[20 * 20] = reinterpret_cast<pel_t( * )[20 * 20]>( &sh.m.tmpL[0][0] );
  static_assert( sizeof( sh.m.tmpL ) + sizeof( sh.m.tmpC ) >= 2 * 20 * 20 * sizeof( pel_t ) && offsetof( Mc2Shared, tmpC ) == offsetof( Mc2Shared, tmpL ) + sizeof( sh.m.tmpL ), "the bilinear predictions lie where the intermediates will" );
  // ---- stage 1: bilinear predictions of the sub-block extended by 2 samples (start MVs clipped against the CU, then moved by -2)
  // (the start vectors, the reference indices and the CU's position and width come with the tile record: the CU record is first needed by the search's last step)
  if( tid < 2 )
  {
    const int l = tid;
    const int cuX = it.clipX, cuY = it.clipY, cuW = 4 * it.clipW4;
    int mvx = l ? it.mv[1][0] : it.mv[0][0], mvy = l ? it.mv[1][1] : it.mv[0][1];      // (values, never a per-lane index into the record: that would put it into scratch)
    // (xinitMC runs once per CU, InterPrediction.cpp:1859: the start MVs are clipped against the CU - its position and, with wrap-around, its width.  Until round 4 the
    // wrap-around case clipped against the sub-block: the same samples unless a clamp is involved, i.e. for vectors beyond a wrap period - tests/bitstreams_open)
    const McBounds B = mc_bounds( pic, cuX, cuY );
    const int wrapOff = mc_clip_mv_w( pic, B, cuX, cuY, pic.hdr.wrap_offset ? cuW : 0, mvx, mvy );
    mvx -= 32; mvy -= 32;
    McSeg g; g.wrapOff = wrapOff; g.bx0 = B.x0; g.by0 = B.y0; g.bx1 = B.x1; g.by1 = B.y1;
    g.w = w + 4; g.h = h + 4; g.xFrac = mvx & 15; g.yFrac = mvy & 15;
    g.x0 = it.x + ( mvx >> 4 ); g.y0 = it.y + ( mvy >> 4 ); g.ww = g.w + 1; g.wh = g.h + 1; g.ox = g.oy = 0; g.padOff = 0; g.cw = g.ww; g.chh = g.wh; g.shX = g.shY = 0;
    sh.m.seg[l][0] = g;
    sh.m.refp[l][0] = refs.p[l * VVR_MAX_REFS + ( l ? it.ref[1] : it.ref[0] )][0];
  }
  __syncthreads();
  mc_load_seg_luma_pair<NT, 21>( sh.m, reco.stride[0], reco.w[0], tid );
  __syncthreads();
  {
    // InterpolationFilter::filter<2> (:589-600) / filterCopy biMCForDMVR (:445-477) at IF_INTERNAL_PREC_BILINEAR = 10
    // One form for the four cases of the reference (copy / horizontal / vertical / both): the horizontal stage to 10 bit, the vertical one on its results.
    // A whole-sample direction is the weights ( 16, 0 ), which leaves the other stage's value as it is: ( 16 t + 8 ) >> 4 = t, and for fewer than 10 bits
    // ( 16 p + offF ) >> shiftF = p << ( 10 - bd ) with ( ( A << ( 10 - bd ) ) + 8 ) >> 4 = ( A + offF ) >> shiftF for the vertical-only case.
    const int shiftF = 4 - ( 10 - bd ), offF = shiftF > 0 ? 1 << ( shiftF - 1 ) : 0;
    const int ew = w + 4, eh = h + 4, n1 = ew * eh;
    const int inv = ( 65536 + ew - 1 ) / ew;          // r / ew = ( r * inv ) >> 16 for r < 400 (ew = 12: 5462, ew = 20: 3277)
    const int fx0 = __builtin_amdgcn_readfirstlane( sh.m.seg[0][0].xFrac ), fy0 = __builtin_amdgcn_readfirstlane( sh.m.seg[0][0].yFrac );
    const int fx1 = __builtin_amdgcn_readfirstlane( sh.m.seg[1][0].xFrac ), fy1 = __builtin_amdgcn_readfirstlane( sh.m.seg[1][0].yFrac );
    for( int i = tid; i < 2 * n1; i += NT )
    {
      const int l = i >= n1, r = i - ( l ? n1 : 0 ), y = ( r * inv ) >> 16, x = r - y * ew;
      const int fx = l ? fx1 : fx0, fy = l ? fy1 : fy0;
      const pel_t* p = &sh.m.winL[l][y * DM_WST_L + x];
      const int t0 = ( p[0] * ( 16 - fx ) + p[1] * fx + offF ) >> shiftF;
      const int t1 = ( p[DM_WST_L] * ( 16 - fx ) + p[DM_WST_L + 1] * fx + offF ) >> shiftF;
      bil[l][y * 20 + x] = (pel_t) ( ( t0 * ( 16 - fy ) + t1 * fy + 8 ) >> 4 );
    }
  }
  __syncthreads();
  // ---- stage 2: SAD at the centre on every second row, early termination, 25-point search with mirrored offsets
  {
    int part = 0;
    const int lw = w == 16 ? 4 : 3;
    for( int i = tid; i < w * ( h >> 1 ); i += NT )
    {
      const int x = i & ( w - 1 ), y = ( i >> lw ) << 1;
      part += iabs( bil[0][( 2 + y ) * 20 + 2 + x] - bil[1][( 2 + y ) * 20 + 2 + x] );
    }
    for( int o = 32; o; o >>= 1 ) part += __shfl_xor( part, o );
    if constexpr( NT > 64 )
    {   // (several wavefronts: their sums meet in LDS)
      if( tid == 0 ) sh.minCost = 0;
      __syncthreads();
      if( ( tid & 63 ) == 0 ) atomicAdd( &sh.minCost, part );
      __syncthreads();
      part = sh.minCost;
    }
    unsigned minCost = (unsigned) part << 1;                               // xGetSAD: uiSum <<= subShift
    minCost >>= 1; minCost -= minCost >> 2;
    const bool search = !( minCost < (unsigned) ( w * h ) );
    if( search )
    {
      // LPC lanes per candidate (2 with one wavefront, 8 with four): each takes every LPC-th of the even rows
      constexpr int LPC = NT / 32;
      const int cand = tid / LPC, half = tid % LPC;
      unsigned sad = 0;
      if( cand < 25 && cand != 12 )
      {
        // two samples per v_sad_u16 (the 10-bit bilinear predictions are not negative); the two blocks start at columns 2 + hor and 2 - hor: both even or
        // both odd - an odd start is realigned from the dwords around it
        const int ver = cand / 5 - 2, hor = cand - ( cand / 5 ) * 5 - 2;
        const int oa = ( 2 + ver ) * 20 + 2 + hor, ob = ( 2 - ver ) * 20 + 2 - hor, odd = oa & 1, nd = w >> 1;
        const uint32_t* a = reinterpret_cast<const uint32_t*>( bil[0] ) + ( ( oa - odd ) >> 1 );
        const uint32_t* b = reinterpret_cast<const uint32_t*>( bil[1] ) + ( ( ob - odd ) >> 1 );
        const int sh = odd << 4;
        for( int y = half * 2; y < h; y += 2 * LPC )
        {
          // the row's nine dwords of both blocks first (one wait for the LDS), then the sums; columns beyond the block are read (they lie inside the 20-sample row) and left out
          const uint32_t* ar = a + y * 10; const uint32_t* br = b + y * 10;
          uint32_t ra[9], rb[9];
#pragma unroll
          for( int d = 0; d < 9; d++ ) { ra[d] = ar[d]; rb[d] = br[d]; }
#pragma unroll
          for( int d = 0; d < 8; d++ )
            if( d < nd ) sad = __builtin_amdgcn_sad_u16( __builtin_amdgcn_alignbit( ra[d + 1], ra[d], sh ), __builtin_amdgcn_alignbit( rb[d + 1], rb[d], sh ), sad );
        }
      }
      for( int o = 1; o < LPC; o <<= 1 ) sad += __shfl_xor( sad, o );
      if( cand < 25 && !half ) sh.sad[cand] = cand == 12 ? minCost : ( ( sad << 1 ) >> 1 );      // X5: ( SAD << subShift ) >> 1
    }
    __syncthreads();
    if( tid == 0 )
    {
      int total0 = 0, total1 = 0;
      if( search )
      {
        int d0 = 0, d1 = 0;
        for( int ver = -2; ver <= 2; ver++ ) for( int hor = -2; hor <= 2; hor++ )
        {
          const unsigned cost = sh.sad[( ver + 2 ) * 5 + hor + 2];
          if( cost < minCost ) { minCost = cost; d0 = hor; d1 = ver; }
        }
        total0 = d0 * 16; total1 = d1 * 16;
        if( iabs( total0 ) != 32 && iabs( total1 ) != 32 )
        {
          const int ci = ( d1 + 2 ) * 5 + d0 + 2;
          const long long s0 = sh.sad[ci], s1 = sh.sad[ci - 1], s2 = sh.sad[ci - 5], s3 = sh.sad[ci + 1], s4 = sh.sad[ci + 5];
          long long num = ( s1 - s3 ) * 16, den = s1 + s3 - 2 * s0;
          if( den != 0 ) total0 += ( s1 != s0 && s3 != s0 ) ? dmvr_div_for_maxq7( num, den ) : ( s1 == s0 ? -8 : 8 );
          num = ( s2 - s4 ) * 16; den = s2 + s4 - 2 * s0;
          if( den != 0 ) total1 += ( s2 != s0 && s4 != s0 ) ? dmvr_div_for_maxq7( num, den ) : ( s2 == s0 ? -8 : 8 );
        }
        total0 = (int16_t) total0; total1 = (int16_t) total1;
      }
      sh.dmv[0] = total0; sh.dmv[1] = total1;
      sh.bioSub = ( minCost < (unsigned) ( 2 * w * h ) ) ? 0 : ( bio ? 1 : 0 );
      const int sub = ( ( it.y - cu.y ) / min( 16, (int) cu.h ) ) * ( ( cu.w + 15 ) >> 4 ) + ( ( it.x - cu.x ) >> 4 );
      dmvrOut[2 * ( cu.dmvr_off + sub )] = total0; dmvrOut[2 * ( cu.dmvr_off + sub ) + 1] = total1;
      if( pic.colMotion )
      {
        // collocated motion (VVR_TOOL_COL_MOTION): the 4x4 units at multiples of 8 luma samples inside the sub-block get the refined MVs, list 0 plus,
        // list 1 minus the delta (DecCu::TaskFinishMotionInfo, DecCu.cpp:186-213)
        for( int y2 = ( it.y + 7 ) & ~7; y2 < it.y + h; y2 += 8 ) for( int x2 = ( it.x + 7 ) & ~7; x2 < it.x + w; x2 += 8 )
        {
          vvr_motion& m = pic.colMotion[( y2 >> 3 ) * pic.colStride + ( x2 >> 3 )];
          m.mv[0][0] = cu.mv[0][0][0] + total0; m.mv[0][1] = cu.mv[0][0][1] + total1;
          m.mv[1][0] = cu.mv[1][0][0] - total0; m.mv[1][1] = cu.mv[1][0][1] - total1;
        }
      }
    }
    __syncthreads();
  }
  const bool bioSub = sh.bioSub != 0;
  // ---- stage 3: final prediction with the refined MVs (clipped against the SUB-block, :1752), same two filter stages as k_mc
  const int headroom = 14 - bd > 2 ? 14 - bd : 2;
  if( tid < 6 )
  {
    const int l = tid / 3, c = tid - 3 * l;
    if( c < ncomp )
    {
      const int sgn = l ? -1 : 1;
      const int mgx = cu.mv[l][0][0], mgy = cu.mv[l][0][1];
      const int rmx = clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, mgx + sgn * sh.dmv[0] ), rmy = clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, mgy + sgn * sh.dmv[1] );
      int cmx = rmx, cmy = rmy;
      const McBounds B = mc_bounds( pic, cu.x, cu.y );
      const int wrapOffF = mc_clip_mv_w( pic, B, it.x, it.y, w, cmx, cmy );
      const int cs = c ? 1 : 0, shf = 4 + cs, ntaps = c ? 4 : 8, half = ntaps / 2 - 1;
      const int dIntX = ( rmx >> shf ) - ( mgx >> shf ), dIntY = ( rmy >> shf ) - ( mgy >> shf );
      McSeg g; g.bx0 = B.x0 >> cs; g.by0 = B.y0 >> cs; g.bx1 = B.x1 >> cs; g.by1 = B.y1 >> cs;
      g.w = w >> cs; g.h = h >> cs;
      g.xFrac = cmx & ( ( 1 << shf ) - 1 ); g.yFrac = cmy & ( ( 1 << shf ) - 1 );
      g.ox = half; g.oy = half; g.ww = g.w + ntaps - 1; g.wh = g.h + ntaps - 1;
      if( dIntX || dIntY )
      {
        // padded local copy: the (w + ntaps - 1)^2 window at the start MV, replicated outwards (xPrefetchPad + paddingCore); the
        // window held in LDS is that copy displaced by the integer part of the refinement
        int pmx = mgx - ( half << shf ), pmy = mgy - ( half << shf );
        // xPrefetchPad (:1545-1556): ONE wrapClipMv, so the ordinary copy is read after a move by one period
        if( pic.hdr.wrap_offset ) g.wrapOff = mc_wrap_clip_mv( pic, it.x, it.y, w, pmx, pmy ) ? pic.hdr.wrap_offset >> cs : 0;
        else { mc_clip_mv( pic, B, it.x, it.y, pmx, pmy ); g.wrapOff = 0; }
        g.x0 = ( it.x >> cs ) + ( pmx >> shf ); g.y0 = ( it.y >> cs ) + ( pmy >> shf );
        g.cw = g.ww; g.chh = g.wh; g.padOff = 2; g.shX = 2 + dIntX; g.shY = 2 + dIntY;
      }
      else
      {
        g.x0 = ( it.x >> cs ) + ( cmx >> shf ) - half; g.y0 = ( it.y >> cs ) + ( cmy >> shf ) - half;
        g.padOff = 0; g.cw = g.ww; g.chh = g.wh; g.shX = g.shY = 0; g.wrapOff = wrapOffF >> cs;
      }
      sh.m.seg[l][c] = g;
      sh.m.refp[l][c] = refs.p[l * VVR_MAX_REFS + cu.ref_idx[l]][c];
    }
  }
  __syncthreads();
  // the tap rows of both stages and every window of the sub-block: one wait
  McTapRows T;
#pragma unroll
  for( int k = 0; k < 2; k++ )
  {
    const bool f4 = w == 4 && h == 4, altHpel = ( it.flags & MC_ITEM_HPEL ) != 0;
    T.h[k][0] = mc_tap_row( 0, sh.m.seg[k][0].xFrac, f4, altHpel ); T.v[k][0] = mc_tap_row( 0, sh.m.seg[k][0].yFrac, f4, altHpel );
    T.h[k][1] = ncomp == 3 ? mc_tap_row( 1, sh.m.seg[k][1].xFrac, false, false ) : 0; T.v[k][1] = ncomp == 3 ? mc_tap_row( 1, sh.m.seg[k][1].yFrac, false, false ) : 0;
  }
  McTapsPre pre;
  mc2_prefetch_taps<NT, 1>( pre, 2, ncomp, w, h, tid, T );
  mc_load_seg_all<NT>( sh.m, ncomp, reco.stride[0], reco.stride[1], reco.w[0], reco.w[1], tid );
  mc2_prefetch_taps<NT, 2>( pre, 2, ncomp, w, h, tid, T );      // (these arrive while stage 1 computes; held across the window loads they would cost the 16th sub-block per compute unit)
  __syncthreads();
  mc2_stage1<NT>( sh.m, 2, ncomp, w, h, headroom, tid, T, pre );
  mc2_stage2<NT>( sh.m, bioSub ? &sh.bs : nullptr, 2, ncomp, MCM_AVG, cu, 2, bd, headroom, reco, it.x, it.y, w, h, tid, fwdLut, T, nullptr, 0, 0, 0, pre );
  if( bioSub )
  {
    __syncthreads();
    mc_bdof_luma<NT>( sh.bs, sh.m.winL[0], sh.m.winL[1], MC2_WST_L, &sh.m.seg[0][0], 3, bd, reco, it.x, it.y, w, h, tid, fwdLut );
  }
}

// =====================================================================================================================
// k_mc_affine — affine motion compensation (+ PROF) of one <= 16x16 tile: 4x4 luma sub-blocks with their own MVs from the
// motion field, 4x4 chroma sub-blocks with the rounded mean of two luma sub-block MVs.
//   InterPrediction::xPredAffineBlk (InterPrediction.cpp:934-1288), applyPROFCore (:61), gradFilterCore<false> (:213),
//   roundAffineMv (Mv.cpp:57), isSubblockVectorSpreadOverLimit (:892); luma 4x4 blocks use the 6-tap table
//   (InterpolationFilter.cpp:1078-1085, 669-676).
// =====================================================================================================================
#define AF_WL 12            // row stride of a luma sub-block window (11 x 11)
#define AF_WC 8             // row stride of a chroma sub-block window (7 x 7)
struct AffSeg { int x0, y0, xFrac, yFrac, wrapOff; };     // window origin (block origin - 3 / - 1) in the reference plane, fractional MV, wrap-around period (0: clamped reads)
struct AffShared {
  __attribute__( ( aligned( 16 ) ) ) pel_t winL[2][16][12 * AF_WL];        // (row 11 is scratch: the second row of the last row pair of the horizontal stage)
  __attribute__( ( aligned( 16 ) ) ) pel_t winC[2][2][4][8 * AF_WC];
  __attribute__( ( aligned( 16 ) ) ) uint32_t tmpL[2][16][4][6];           // horizontal stage, 14 bit: [column][row pair]
  __attribute__( ( aligned( 16 ) ) ) uint32_t tmpC[2][2][4][4][4];
  pel_t  ext[2][16][36];          // PROF: 6x6 block of 14-bit luma samples (interior = prediction, ring = integer reference samples)
  pel_t  predL[2][16 * 16];       // per list: final (uni) or 14-bit (bi) luma after PROF
  AffSeg segL[2][16], segC[2][4];
  int    dMvH[2][16], dMvV[2][16], prof[2];
  const pel_t* refp[2][3];
};

__device__ __forceinline__ void aff_round_mv( int& mx, int& my, int sh ) { const int o = 1 << ( sh - 1 ); mx = ( mx + o - ( mx >= 0 ) ) >> sh; my = ( my + o - ( my >= 0 ) ) >> sh; }

__device__ __forceinline__ bool aff_spread_over_limit( int a, int b, int c, int d, int predType )
{
  const int s4 = 4 << 11, ft = 6;
  if( predType == 3 )
  {
    int rw = max( max( 0, 4 * a + s4 ), max( 4 * c, 4 * a + 4 * c + s4 ) ) - min( min( 0, 4 * a + s4 ), min( 4 * c, 4 * a + 4 * c + s4 ) );
    int rh = max( max( 0, 4 * b ), max( 4 * d + s4, 4 * b + 4 * d + s4 ) ) - min( min( 0, 4 * b ), min( 4 * d + s4, 4 * b + 4 * d + s4 ) );
    rw = ( rw >> 11 ) + ft + 3; rh = ( rh >> 11 ) + ft + 3;
    return rw * rh > ( ft + 9 ) * ( ft + 9 );
  }
  int rw = max( 0, 4 * a + s4 ) - min( 0, 4 * a + s4 ), rh = max( 0, 4 * b ) - min( 0, 4 * b );
  rw = ( rw >> 11 ) + ft + 3; rh = ( rh >> 11 ) + ft + 3;
  if( rw * rh > ( ft + 9 ) * ( ft + 5 ) ) return true;
  rw = max( 0, 4 * c ) - min( 0, 4 * c ); rh = max( 0, 4 * d + s4 ) - min( 0, 4 * d + s4 );
  rw = ( rw >> 11 ) + ft + 3; rh = ( rh >> 11 ) + ft + 3;
  return rw * rh > ( ft + 5 ) * ( ft + 9 );
}

// MV of the 4x4 luma sub-block (wx, wy) of an affine CU for list l, spanned from the CU's control points the way the motion field is filled on the
// host (PU::setAllAffineMv, UnitTools.cpp:2689-2810): the affine model evaluated at the sub-block's centre in 1/16 sample with 7 fractional bits
// more, rounded away from zero at .5, clipped to the 18-bit MV storage range; ONE vector for the whole CU (the model at the CU's centre) when the
// sub-block vectors would spread too far (isSubblockVectorSpreadOverLimit, InterPrediction.cpp:892).  VVR_TOOL_AFFINE_MV_ON_DEVICE.
// (the CU's fields as values: the control points of ONE list, already chosen - the record itself is read once per tile, see AffCu)
struct AffCpmv { int m[3][2]; };
__device__ __forceinline__ void aff_span_mv( const AffCpmv& P, int cuW, int cuH, bool sixP, int interDir, int wx, int wy, int& mx, int& my )
{
  const int lw = ilog2( cuW ), lh = ilog2( cuH );
  const int dHX = ( P.m[1][0] - P.m[0][0] ) * ( 1 << ( 7 - lw ) ), dHY = ( P.m[1][1] - P.m[0][1] ) * ( 1 << ( 7 - lw ) );
  int dVX, dVY;
  if( sixP ) { dVX = ( P.m[2][0] - P.m[0][0] ) * ( 1 << ( 7 - lh ) ); dVY = ( P.m[2][1] - P.m[0][1] ) * ( 1 << ( 7 - lh ) ); }
  else { dVX = -dHY; dVY = dHX; }
  const bool over = aff_spread_over_limit( dHX, dHY, dVX, dVY, interDir );
  const int px = over ? cuW >> 1 : 2 + 4 * wx, py = over ? cuH >> 1 : 2 + 4 * wy;
  mx = P.m[0][0] * 128 + dHX * px + dVX * py; my = P.m[0][1] * 128 + dHY * px + dVY * py;
  aff_round_mv( mx, my, 7 );
  mx = clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, mx ); my = clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, my );
}

// (from the record in global memory: k_mc_rpr)
__device__ __forceinline__ void aff_span_mv( const vvr_cu& cu, int l, int wx, int wy, int& mx, int& my )
{
  AffCpmv P;
#pragma unroll
  for( int a = 0; a < 3; a++ ) { P.m[a][0] = cu.mv[l][a][0]; P.m[a][1] = cu.mv[l][a][1]; }
  aff_span_mv( P, cu.w, cu.h, ( cu.flags & VVR_CU_AFFINE_6P ) != 0, cu.inter_dir, wx, wy, mx, my );
}

// one chroma sample of a 4x4 sub-block from its column of horizontally filtered row pairs (vertical 4-tap stage; the identity filter for a whole-sample vector:
// the four (xFrac, yFrac) cases of xPredAffineBlk :1224-1236 = xPredInterBlk's arithmetic)
__device__ __forceinline__ int aff_chroma_sample( const uint32_t* col /* 4 row pairs */, int yFrac, int py, bool bi, int bd, int headroom )
{
  const uint4 u = *reinterpret_cast<const uint4*>( col );
  const uint4 C = d_mcTaps[MCT_CHROMA + yFrac];
  const uint32_t D[4] = { u.x, u.y, u.z, u.w };
  const int m = py >> 1;
  uint32_t a = m ? D[1] : D[0], b = m ? D[2] : D[1];
  if( py & 1 ) { const uint32_t c = m ? D[3] : D[2]; a = __builtin_amdgcn_alignbit( b, a, 16 ); b = __builtin_amdgcn_alignbit( c, b, 16 ); }
  const int shift2 = 6 + headroom, offset2 = ( 1 << ( shift2 - 1 ) ) + ( IF_INTERNAL_OFFS << 6 );
  const int sum = mc_dot2( b, C.y, mc_dot2( a, C.x, bi ? 0 : offset2 ) );
  return bi ? sum >> 6 : clip_pel( sum >> shift2, bd );
}

template<int NT>
__global__ __launch_bounds__( NT ) void k_mc_affine( PicDev pic, RefSet refs, DevPlanes reco, const McItem* __restrict__ items, int numItems )
{
  __shared__ AffShared sh;
  const int item = mc_item_index();
  if( item >= numItems ) return;
  const McItem it = items[item];
  const int16_t* __restrict__ fwdLut = lmcs_fwd_at( pic, it.x, it.y );      // LMCS (where the tile's slice uses it): luma predictions are stored forward-mapped (lmcs_fwd_luma)
  // The CU record is read ONCE, as dwords all in flight together: the geometry below evaluates the affine model per lane and list, and
  // every field it touched through a reference into global memory was a load of its own with a wait of its own (twenty of them before the first window sample)
  struct { int x, y, w, h, flags, inter_dir, ref_idx[2], bcw_idx; int mv[2][3][2]; } cu;
  {
    static_assert( offsetof( vvr_cu, w ) == 4 && offsetof( vvr_cu, flags ) == 8 && offsetof( vvr_cu, inter_dir ) == 20 && offsetof( vvr_cu, ref_idx ) == 21 && offsetof( vvr_cu, bcw_idx ) == 23
                   && offsetof( vvr_cu, mv ) == 32 && sizeof( vvr_cu ) % 4 == 0, "the dwords of vvr_cu read below" );
    const uint32_t* __restrict__ cw = reinterpret_cast<const uint32_t*>( &pic.cu[__builtin_amdgcn_readfirstlane( it.cu )] );
    uint32_t hd[4]; int mvv[12];
    hd[0] = cw[0]; hd[1] = cw[1]; hd[2] = cw[2]; hd[3] = cw[5];
#pragma unroll
    for( int i = 0; i < 12; i++ ) mvv[i] = (int) cw[8 + i];
#pragma unroll
    for( int i = 0; i < 4; i++ ) hd[i] = (uint32_t) __builtin_amdgcn_readfirstlane( (int) hd[i] );
    cu.x = hd[0] & 0xffff; cu.y = hd[0] >> 16; cu.w = hd[1] & 0xff; cu.h = ( hd[1] >> 8 ) & 0xff; cu.flags = hd[2] & 0xffff;
    cu.inter_dir = hd[3] & 0xff; cu.ref_idx[0] = (int) (int8_t) ( hd[3] >> 8 ); cu.ref_idx[1] = (int) (int8_t) ( hd[3] >> 16 ); cu.bcw_idx = hd[3] >> 24;
#pragma unroll
    for( int i = 0; i < 12; i++ ) cu.mv[i / 6][( i / 2 ) % 3][i & 1] = mvv[i];      // (the same in every lane, but in vector registers: twelve more scalar ones cost the eighth resident tile)
  }
  const int bd = pic.hdr.bit_depth, ctu = 1 << pic.hdr.log2_ctu;
  const int tid = threadIdx.x;
  const int ncomp = pic.hdr.chroma_format ? 3 : 1;
  const int w = it.w, h = it.h;
  const int sbx = w >> 2, nsb = sbx * ( h >> 2 );            // luma sub-blocks in the tile
  const int cbx = w >> 3, ncb = cbx * ( h >> 3 );            // chroma sub-blocks (4x4 chroma samples = 8x8 luma)
  bool biPred = cu.ref_idx[0] >= 0 && cu.ref_idx[1] >= 0;
  // xCheckIdenticalMotion (:404-436): same reference picture and same control-point MVs in both lists -> list 0 only
  if( biPred && pic.hdr.ref_poc[0][cu.ref_idx[0]] == pic.hdr.ref_poc[1][cu.ref_idx[1]]
      && cu.mv[0][0][0] == cu.mv[1][0][0] && cu.mv[0][0][1] == cu.mv[1][0][1] && cu.mv[0][1][0] == cu.mv[1][1][0] && cu.mv[0][1][1] == cu.mv[1][1][1]
      && ( !( cu.flags & VVR_CU_AFFINE_6P ) || ( cu.mv[0][2][0] == cu.mv[1][2][0] && cu.mv[0][2][1] == cu.mv[1][2][1] ) ) && !pic.wp /* :408 */ ) biPred = false;
  const int l0 = cu.ref_idx[0] >= 0 ? 0 : 1, nl = biPred ? 2 : 1;
  const vvr_wp_params* __restrict__ wpT = wp_at( pic, cu.x, cu.y );
  const bool wpOn = wpT && cu.bcw_idx == 2;       // explicit weighted prediction: also a single list stays at 14 bit until the final stage
  const bool hi = biPred || wpOn;
  const int headroom = 14 - bd > 2 ? 14 - bd : 2;
  const McBounds AB = mc_bounds( pic, cu.x, cu.y );
  bool inside = true;                 // every window of the tile this lane laid out lies inside the picture (sub-picture): the windows can be loaded as dwords
  // ---- sub-block geometry
  {
    // (picture bounds, or the CU's sub-picture when that is treated as a picture: clipMvInSubpic against the CU, :1188-1193)
    const int horMax = ( AB.x1 + 1 + 8 - cu.x - 1 ) * 16, horMin = ( -ctu - 8 - ( cu.x - AB.x0 ) + 1 ) * 16;
    const int verMax = ( AB.y1 + 1 + 8 - cu.y - 1 ) * 16, verMin = ( -ctu - 8 - ( cu.y - AB.y0 ) + 1 ) * 16;
    const bool onDev = ( pic.hdr.tool_flags & VVR_TOOL_AFFINE_MV_ON_DEVICE ) != 0;
    const bool sixP = ( cu.flags & VVR_CU_AFFINE_6P ) != 0;
    for( int i = tid; i < nl * ( nsb + ncb ); i += NT )
    {
      const int k = i / ( nsb + ncb ), r = i - k * ( nsb + ncb ), l = biPred ? k : l0;
      AffCpmv P;        // the control points of this lane's list (values chosen per lane, never an index into the record)
#pragma unroll
      for( int a = 0; a < 3; a++ ) { P.m[a][0] = l ? cu.mv[1][a][0] : cu.mv[0][a][0]; P.m[a][1] = l ? cu.mv[1][a][1] : cu.mv[0][a][1]; }
      AffSeg g;
      if( r < nsb )
      {
        const int sx = r % sbx, sy = r / sbx;
        int smx, smy;
        if( onDev ) aff_span_mv( P, cu.w, cu.h, sixP, cu.inter_dir, ( ( it.x - cu.x ) >> 2 ) + sx, ( ( it.y - cu.y ) >> 2 ) + sy, smx, smy );
        else { const vvr_motion& m = pic.affMotion[it.mv[0][0] + 4 * sy + sx]; smx = m.mv[l][0]; smy = m.mv[l][1]; }
        int mx = smx, my = smy;
        // wrap-around: ONE wrapClipMv per sub-block, against the sub-block (:1177-1186)
        if( pic.hdr.wrap_offset ) g.wrapOff = mc_wrap_clip_mv( pic, it.x + 4 * sx, it.y + 4 * sy, 4, mx, my ) ? pic.hdr.wrap_offset : 0;
        else { mx = min( horMax, max( horMin, smx ) ); my = min( verMax, max( verMin, smy ) ); g.wrapOff = 0; }
        g.xFrac = mx & 15; g.yFrac = my & 15;
        g.x0 = it.x + 4 * sx + ( mx >> 4 ) - 3; g.y0 = it.y + 4 * sy + ( my >> 4 ) - 3;
        sh.segL[k][r] = g;
        inside = inside && !g.wrapOff && g.x0 >= AB.x0 && g.y0 >= AB.y0 && g.x0 + 10 <= AB.x1 && g.y0 + 10 <= AB.y1 && g.x0 - ( g.x0 & 1 ) + 12 <= reco.stride[0];
      }
      else
      {
        const int q = r - nsb, sx = q % cbx, sy = q / cbx;
        int mx, my;
        if( onDev )
        {
          int ax, ay, bx, by;
          aff_span_mv( P, cu.w, cu.h, sixP, cu.inter_dir, ( ( it.x - cu.x ) >> 2 ) + 2 * sx, ( ( it.y - cu.y ) >> 2 ) + 2 * sy, ax, ay );
          aff_span_mv( P, cu.w, cu.h, sixP, cu.inter_dir, ( ( it.x - cu.x ) >> 2 ) + 2 * sx + 1, ( ( it.y - cu.y ) >> 2 ) + 2 * sy + 1, bx, by );
          mx = ax + bx; my = ay + by;
        }
        else
        {
          const vvr_motion& m0 = pic.affMotion[it.mv[0][0] + 4 * ( 2 * sy ) + 2 * sx];
          const vvr_motion& m1 = pic.affMotion[it.mv[0][0] + 4 * ( 2 * sy + 1 ) + 2 * sx + 1];
          mx = m0.mv[l][0] + m1.mv[l][0]; my = m0.mv[l][1] + m1.mv[l][1];
        }
        aff_round_mv( mx, my, 1 );
        if( pic.hdr.wrap_offset ) g.wrapOff = mc_wrap_clip_mv( pic, it.x + 8 * sx, it.y + 8 * sy, 8, mx, my ) ? pic.hdr.wrap_offset >> 1 : 0;
        else { mx = min( horMax, max( horMin, mx ) ); my = min( verMax, max( verMin, my ) ); g.wrapOff = 0; }
        g.xFrac = mx & 31; g.yFrac = my & 31;
        g.x0 = ( it.x >> 1 ) + 4 * sx + ( mx >> 5 ) - 1; g.y0 = ( it.y >> 1 ) + 4 * sy + ( my >> 5 ) - 1;
        sh.segC[k][q] = g;
        inside = inside && !g.wrapOff && g.x0 >= ( AB.x0 >> 1 ) && g.y0 >= ( AB.y0 >> 1 ) && g.x0 + 6 <= ( AB.x1 >> 1 ) && g.y0 + 6 <= ( AB.y1 >> 1 ) && g.x0 - ( g.x0 & 1 ) + 8 <= reco.stride[1];
      }
    }
    if( tid < nl )
    {
      const int k = tid, l = biPred ? k : l0;
      for( int c = 0; c < ncomp; c++ ) sh.refp[k][c] = refs.p[l * VVR_MAX_REFS + ( l ? cu.ref_idx[1] : cu.ref_idx[0] )][c];
      // PROF switch and the per-position MV offsets of a 4x4 sub-block (:1015-1090)
      AffCpmv P;
#pragma unroll
      for( int a = 0; a < 3; a++ ) { P.m[a][0] = l ? cu.mv[1][a][0] : cu.mv[0][a][0]; P.m[a][1] = l ? cu.mv[1][a][1] : cu.mv[0][a][1]; }
      const int lw = ilog2( cu.w ), lh = ilog2( cu.h );
      const int dHX = ( P.m[1][0] - P.m[0][0] ) * ( 1 << ( 7 - lw ) ), dHY = ( P.m[1][1] - P.m[0][1] ) * ( 1 << ( 7 - lw ) );
      int dVX, dVY;
      if( sixP ) { dVX = ( P.m[2][0] - P.m[0][0] ) * ( 1 << ( 7 - lh ) ); dVY = ( P.m[2][1] - P.m[0][1] ) * ( 1 << ( 7 - lh ) ); }
      else { dVX = -dHY; dVY = dHX; }
      const bool eqRT = P.m[0][0] == P.m[1][0] && P.m[0][1] == P.m[1][1];
      const bool eqLB = P.m[0][0] == P.m[2][0] && P.m[0][1] == P.m[2][1];
      bool prof = ( pic.hdr.tool_flags & VVR_TOOL_PROF ) != 0;
      prof = prof && !( ( sixP && eqRT && eqLB ) || ( !sixP && eqRT ) ) && !aff_spread_over_limit( dHX, dHY, dVX, dVY, cu.inter_dir );
      sh.prof[k] = prof;
      if( prof )
      {
        const int qHX = dHX * 4, qHY = dHY * 4, qVX = dVX * 4, qVY = dVY * 4;
        const int h0 = ( ( dHX + dVX ) * 2 ) - ( ( qHX + qVX ) * 2 ), v0 = ( ( dHY + dVY ) * 2 ) - ( ( qHY + qVY ) * 2 );
        for( int y = 0; y < 4; y++ ) for( int x = 0; x < 4; x++ )
        {
          int a = h0 + x * qHX + y * qVX, b = v0 + x * qHY + y * qVY;
          aff_round_mv( a, b, 8 );
          sh.dMvH[k][y * 4 + x] = clip3( -31, 31, a ); sh.dMvV[k][y * 4 + x] = clip3( -31, 31, b );
        }
      }
    }
  }
  const bool fastWin = __syncthreads_and( inside ) != 0;
  // ---- windows: 11x11 per luma sub-block, 7x7 per chroma sub-block
  if( fastWin )
  {
    // every window inside: dword loads, 8 (luma) / 4 (chroma) lanes per window row, an odd first column realigned with one DPP move + v_alignbit (as mc3_load_*).
    // A bi-predicted 16x16 tile has 352 luma rows (11 passes of the workgroup) and 112 chroma rows (2 passes): the loads of six passes are issued before the first
    // of them is waited for - two memory round trips instead of thirteen (all at once would cost the eighth resident tile of a compute unit its registers)
    constexpr int ITL = ( 2 * 16 * 11 + NT / 8 - 1 ) / ( NT / 8 ), ITC = ( 2 * 2 * 4 * 7 + NT / 4 - 1 ) / ( NT / 4 ), HL = ( ITL + 1 ) / 2;
    const int q = tid & 7, rowsL = nl * nsb * 11, lgN = ilog2( nsb );            // (nsb is a power of two)
    const int q4 = tid & 3, rowsC = ncomp == 3 ? nl * 2 * ncb * 7 : 0, lgC = ilog2( max( ncb, 1 ) );
    uint32_t* const winLdw = reinterpret_cast<uint32_t*>( &sh.winL[0][0][0] );
    uint32_t* const winCdw = reinterpret_cast<uint32_t*>( &sh.winC[0][0][0][0] );
#pragma unroll
    for( int half = 0; half < 2; half++ )
    {
      uint32_t v[HL], vc[ITC]; int dw[HL], dc[ITC];
#pragma unroll
      for( int j = 0; j < HL; j++ )
      {
        v[j] = 0; dw[j] = -1;
        const int itn = half * HL + j, R = ( tid >> 3 ) + itn * ( NT / 8 );
        if( itn < ITL && R < rowsL )
        {
          const int blk = R / 11, yy = R - blk * 11, k = blk >> lgN, sb = blk & ( nsb - 1 );
          const int x0 = sh.segL[k][sb].x0, y0 = sh.segL[k][sb].y0, odd = x0 & 1;
          if( q < 6 ) v[j] = reinterpret_cast<const uint32_t*>( sh.refp[k][0] + (size_t) ( y0 + yy ) * reco.stride[0] + ( x0 - odd ) )[q];
          dw[j] = ( ( ( ( k * 16 + sb ) * 12 + yy ) * AF_WL ) >> 1 ) + q + ( odd << 30 );
        }
      }
      if( half == 1 )
      {
#pragma unroll
        for( int j = 0; j < ITC; j++ )
        {
          vc[j] = 0; dc[j] = -1;
          const int R = ( tid >> 2 ) + j * ( NT / 4 );
          if( R < rowsC )
          {
            const int blk = R / 7, yy = R - blk * 7, k = blk >> ( lgC + 1 ), qq = blk & ( 2 * ncb - 1 ), c = qq >> lgC, sb = qq & ( ncb - 1 );
            const int x0 = sh.segC[k][sb].x0, y0 = sh.segC[k][sb].y0, odd = x0 & 1;
            vc[j] = reinterpret_cast<const uint32_t*>( sh.refp[k][1 + c] + (size_t) ( y0 + yy ) * reco.stride[1] + ( x0 - odd ) )[q4];
            dc[j] = ( ( ( ( ( k * 2 + c ) * 4 + sb ) * 8 + yy ) * AF_WC ) >> 1 ) + q4 + ( odd << 30 );
          }
        }
      }
#pragma unroll
      for( int j = 0; j < HL; j++ )
      {
        const uint32_t wv = __builtin_amdgcn_alignbit( mc_dpp_next_lane( v[j] ), v[j], ( dw[j] >> 30 ) << 4 );      // (a pass that has no row: dw = -1, the shift does not matter)
        if( dw[j] >= 0 && q < 6 ) winLdw[dw[j] & 0x3fffffff] = wv;
      }
      if( half == 1 )
      {
#pragma unroll
        for( int j = 0; j < ITC; j++ )
        {
          const uint32_t wv = __builtin_amdgcn_alignbit( mc_dpp_next_lane( vc[j] ), vc[j], ( dc[j] >> 30 ) << 4 );      // (lane 3 takes its upper half from another row: column 7 of the window is never read)
          if( dc[j] >= 0 ) winCdw[dc[j] & 0x3fffffff] = wv;
        }
      }
    }
  }
  else
  {
    // clamped / wrapped coordinates, sample by sample
    const int nL = nl * nsb * 11 * AF_WL;
    for( int i = tid; i < nL; i += NT )
    {
      const int blk = i / ( 11 * AF_WL ), r = i - blk * ( 11 * AF_WL ), yy = r / AF_WL, xx = r - yy * AF_WL;
      if( xx >= 11 ) continue;
      const int k = blk / nsb, sb = blk - k * nsb;
      const AffSeg g = sh.segL[k][sb];
      const int sx = mc_ref_col( g.x0 + xx, AB.x0, AB.x1, reco.w[0], g.wrapOff ), sy = clip3( AB.y0, AB.y1, g.y0 + yy );
      sh.winL[k][sb][r] = sh.refp[k][0][(size_t) sy * reco.stride[0] + sx];
    }
    if( ncomp == 3 )
    {
      const int nC = nl * 2 * ncb * 7 * AF_WC;
      for( int i = tid; i < nC; i += NT )
      {
        const int blk = i / ( 7 * AF_WC ), r = i - blk * ( 7 * AF_WC ), yy = r / AF_WC, xx = r - yy * AF_WC;
        if( xx >= 7 ) continue;
        const int k = blk / ( 2 * ncb ), q = blk - k * 2 * ncb, c = q / ncb, sb = q - c * ncb;
        const AffSeg g = sh.segC[k][sb];
        const int sx = mc_ref_col( g.x0 + xx, AB.x0 >> 1, AB.x1 >> 1, reco.w[1], g.wrapOff ), sy = clip3( AB.y0 >> 1, AB.y1 >> 1, g.y0 + yy );
        sh.winC[k][c][sb][r] = sh.refp[k][1 + c][(size_t) sy * reco.stride[1] + sx];
      }
    }
  }
  __syncthreads();
  // ---- the two filter stages of every sub-block, in the form of k_mc (round 6): horizontal filter of two window rows x four columns per work item to 14-bit
  // intermediates [column][row pair], vertical filter of a column of four rows per work item and list - v_dot2_i32_i16 on sample pairs, the identity filter where
  // the vector has no fractional part in a direction (exact: 64 s >> 6; xPredAffineBlk :1224-1236 = xPredInterBlk's four cases), taps from d_mcTaps per lane
  // (every sub-block has its own fraction; luma: the table of 4x4 blocks)
  {
    const int shift1 = 6 - headroom, offset1 = __builtin_amdgcn_readfirstlane( -IF_INTERNAL_OFFS * ( 1 << shift1 ) );
    const int itemsL = nl * nsb * 6;
    for( int idx = tid; idx < itemsL; idx += NT )
    {
      const int blk = idx / 6, rp = idx - blk * 6, k = blk / nsb, sb = blk - k * nsb;
      const uint4 C = d_mcTaps[MCT_4X4 + sh.segL[k][sb].xFrac];
      const uint2* s0 = reinterpret_cast<const uint2*>( &sh.winL[k][sb][2 * rp * AF_WL] );
      int a[4], b[4];
#pragma unroll
      for( int rr = 0; rr < 2; rr++ )
      {
        const uint2 u0 = s0[3 * rr], u1 = s0[3 * rr + 1], u2 = s0[3 * rr + 2];
        const uint32_t D[6] = { u0.x, u0.y, u1.x, u1.y, u2.x, u2.y };
        uint32_t S[5];
#pragma unroll
        for( int e = 0; e < 5; e++ ) S[e] = __builtin_amdgcn_alignbit( D[e + 1], D[e], 16 );
        int o[4];
        o[0] = mc_dot2_init( D[0], C.x, offset1 ); o[1] = mc_dot2_init( S[0], C.x, offset1 ); o[2] = mc_dot2_init( D[1], C.x, offset1 ); o[3] = mc_dot2_init( S[1], C.x, offset1 );
        o[0] = mc_dot2( D[1], C.y, o[0] ); o[1] = mc_dot2( S[1], C.y, o[1] ); o[2] = mc_dot2( D[2], C.y, o[2] ); o[3] = mc_dot2( S[2], C.y, o[3] );
        o[0] = mc_dot2( D[2], C.z, o[0] ); o[1] = mc_dot2( S[2], C.z, o[1] ); o[2] = mc_dot2( D[3], C.z, o[2] ); o[3] = mc_dot2( S[3], C.z, o[3] );
        o[0] = mc_dot2( D[3], C.w, o[0] ); o[1] = mc_dot2( S[3], C.w, o[1] ); o[2] = mc_dot2( D[4], C.w, o[2] ); o[3] = mc_dot2( S[4], C.w, o[3] );
#pragma unroll
        for( int e = 0; e < 4; e++ ) { if( rr ) b[e] = o[e] >> shift1; else a[e] = o[e] >> shift1; }
      }
#pragma unroll
      for( int e = 0; e < 4; e++ ) sh.tmpL[k][sb][e][rp] = __builtin_amdgcn_perm( (uint32_t) b[e], (uint32_t) a[e], 0x05040100u );
    }
    if( ncomp == 3 )
    {
      const int itemsC = nl * 2 * ncb * 4;
      for( int idx = tid; idx < itemsC; idx += NT )
      {
        const int blk = idx >> 2, rp = idx & 3, k = blk / ( 2 * ncb ), qq = blk - k * 2 * ncb, c = qq / ncb, sb = qq - c * ncb;
        const uint4 C = d_mcTaps[MCT_CHROMA + sh.segC[k][sb].xFrac];
        int a[4], b[4];
#pragma unroll
        for( int rr = 0; rr < 2; rr++ )
        {
          const uint4 u = *reinterpret_cast<const uint4*>( &sh.winC[k][c][sb][( 2 * rp + rr ) * AF_WC] );
          const uint32_t D[4] = { u.x, u.y, u.z, u.w };
          uint32_t S[3];
#pragma unroll
          for( int e = 0; e < 3; e++ ) S[e] = __builtin_amdgcn_alignbit( D[e + 1], D[e], 16 );
          int o[4];
          o[0] = mc_dot2_init( D[0], C.x, offset1 ); o[1] = mc_dot2_init( S[0], C.x, offset1 ); o[2] = mc_dot2_init( D[1], C.x, offset1 ); o[3] = mc_dot2_init( S[1], C.x, offset1 );
          o[0] = mc_dot2( D[1], C.y, o[0] ); o[1] = mc_dot2( S[1], C.y, o[1] ); o[2] = mc_dot2( D[2], C.y, o[2] ); o[3] = mc_dot2( S[2], C.y, o[3] );
#pragma unroll
          for( int e = 0; e < 4; e++ ) { if( rr ) b[e] = o[e] >> shift1; else a[e] = o[e] >> shift1; }
        }
#pragma unroll
        for( int e = 0; e < 4; e++ ) sh.tmpC[k][c][sb][e][rp] = __builtin_amdgcn_perm( (uint32_t) b[e], (uint32_t) a[e], 0x05040100u );
      }
    }
  }
  __syncthreads();
  // ---- luma: vertical stage = the prediction (14-bit when bi-predicted or refined by PROF; else the final sample), a column of a sub-block per work item and list
  {
    const int shift2 = 6 + headroom, offset2 = ( 1 << ( shift2 - 1 ) ) + ( IF_INTERNAL_OFFS << 6 );
    for( int idx = tid; idx < nl * nsb * 4; idx += NT )
    {
      const int k = idx / ( nsb * 4 ), r = idx - k * nsb * 4, sb = r >> 2, x = r & 3;
      const bool prof = sh.prof[k] != 0, fin = !( hi || prof );
      const uint4 C = d_mcTaps[MCT_4X4 + sh.segL[k][sb].yFrac];
      const uint2* tp = reinterpret_cast<const uint2*>( sh.tmpL[k][sb][x] );
      const uint2 u0 = tp[0], u1 = tp[1], u2 = tp[2];
      const uint32_t D[6] = { u0.x, u0.y, u1.x, u1.y, u2.x, u2.y };
      uint32_t S[5];
#pragma unroll
      for( int e = 0; e < 5; e++ ) S[e] = __builtin_amdgcn_alignbit( D[e + 1], D[e], 16 );
      const int init = fin ? offset2 : 0;
      int o[4] = { init, init, init, init };
      o[0] = mc_dot2( D[0], C.x, o[0] ); o[1] = mc_dot2( S[0], C.x, o[1] ); o[2] = mc_dot2( D[1], C.x, o[2] ); o[3] = mc_dot2( S[1], C.x, o[3] );
      o[0] = mc_dot2( D[1], C.y, o[0] ); o[1] = mc_dot2( S[1], C.y, o[1] ); o[2] = mc_dot2( D[2], C.y, o[2] ); o[3] = mc_dot2( S[2], C.y, o[3] );
      o[0] = mc_dot2( D[2], C.z, o[0] ); o[1] = mc_dot2( S[2], C.z, o[1] ); o[2] = mc_dot2( D[3], C.z, o[2] ); o[3] = mc_dot2( S[3], C.z, o[3] );
      o[0] = mc_dot2( D[3], C.w, o[0] ); o[1] = mc_dot2( S[3], C.w, o[1] ); o[2] = mc_dot2( D[4], C.w, o[2] ); o[3] = mc_dot2( S[4], C.w, o[3] );
#pragma unroll
      for( int y = 0; y < 4; y++ )
      {
        const int v = fin ? clip_pel( o[y] >> shift2, bd ) : o[y] >> 6;
        if( prof ) sh.ext[k][sb][( 1 + y ) * 6 + 1 + x] = (pel_t) v;
        else sh.predL[k][sb * 16 + y * 4 + x] = (pel_t) v;
      }
    }
  }
  for( int i = tid; i < nl * nsb * 20; i += NT )
  {
    const int blk = i / 20, r = i - blk * 20, k = blk / nsb, sb = blk - k * nsb;
    if( !sh.prof[k] ) continue;
    int ei, ej;                                  // ring of the 6x6 block
    if( r < 6 ) { ei = r; ej = 0; } else if( r < 12 ) { ei = r - 6; ej = 5; } else { ei = ( r & 1 ) ? 5 : 0; ej = 1 + ( ( r - 12 ) >> 1 ); }
    const AffSeg g = sh.segL[k][sb];
    const int sref = sh.winL[k][sb][( 3 + ej - 1 + ( g.yFrac >> 3 ) ) * AF_WL + 3 + ei - 1 + ( g.xFrac >> 3 )];
    sh.ext[k][sb][ej * 6 + ei] = (pel_t) ( (int16_t) ( sref << headroom ) - (int16_t) IF_INTERNAL_OFFS );
  }
  __syncthreads();
  for( int i = tid; i < nl * w * h; i += NT )
  {
    const int k = i / ( w * h ), r = i - k * w * h, sb = r >> 4, px = r & 3, py = ( r >> 2 ) & 3;
    if( !sh.prof[k] ) continue;
    const pel_t* sp = &sh.ext[k][sb][( 1 + py ) * 6 + 1 + px];
    const int gY = (int16_t) ( ( sp[6] >> 6 ) - ( sp[-6] >> 6 ) ), gX = (int16_t) ( ( sp[1] >> 6 ) - ( sp[-1] >> 6 ) );
    const int dILimit = 1 << max( bd + 1, 13 );
    int dI = sh.dMvH[k][py * 4 + px] * gX + sh.dMvV[k][py * 4 + px] * gY;
    dI = clip3( -dILimit, dILimit - 1, dI );
    int v = (int16_t) ( sp[0] + dI );
    if( !hi ) { v = (int16_t) ( ( v + ( 1 << ( headroom - 1 ) ) + IF_INTERNAL_OFFS ) >> headroom ); v = clip_pel( v, bd ); }
    sh.predL[k][sb * 16 + py * 4 + px] = (pel_t) v;
  }
  __syncthreads();
  // ---- output: uni-directional result, bi-predictive average or BCW (xWeightedAverage :1349; no BDOF with affine)
  for( int c = 0; c < ncomp; c++ )
  {
    const int cs = c ? 1 : 0, cw = w >> cs, chh = h >> cs, sbw = cw >> 2;
    for( int i = tid; i < cw * chh; i += NT )
    {
      const int sb = i >> 4, px = i & 3, py = ( i >> 2 ) & 3;
      int p0, p1 = 0;
      if( c == 0 ) { p0 = sh.predL[0][sb * 16 + py * 4 + px]; if( biPred ) p1 = sh.predL[1][sb * 16 + py * 4 + px]; }
      else
      {
        p0 = aff_chroma_sample( sh.tmpC[0][c - 1][sb][px], sh.segC[0][sb].yFrac, py, hi, bd, headroom );
        if( biPred ) p1 = aff_chroma_sample( sh.tmpC[1][c - 1][sb][px], sh.segC[1][sb].yFrac, py, true, bd, headroom );
      }
      int out = p0;
      if( wpOn ) out = biPred ? wp_bi( wpT, cu.ref_idx[0], cu.ref_idx[1], c, p0, p1, bd, headroom ) : wp_uni( wpT, l0, l0 ? cu.ref_idx[1] : cu.ref_idx[0], c, p0, bd, headroom );
      else if( biPred )
      {
        if( cu.bcw_idx != 2 )
        {
          const int w1 = d_bcw_weights[cu.bcw_idx], w0 = 8 - w1, shift = headroom + 3, offset = ( 1 << ( shift - 1 ) ) + ( IF_INTERNAL_OFFS << 3 );
          out = clip_pel( ( p0 * w0 + p1 * w1 + offset ) >> shift, bd );
        }
        else
        {
          const int shift = headroom + 1, offset = ( 1 << ( shift - 1 ) ) + 2 * IF_INTERNAL_OFFS;
          out = clip_pel( ( p0 + p1 + offset ) >> shift, bd );
        }
      }
      const int x = ( it.x >> cs ) + 4 * ( sb % sbw ) + px, y = ( it.y >> cs ) + 4 * ( sb / sbw ) + py;
      reco.p[c][(size_t) y * reco.stride[c] + x] = (pel_t) lmcs_fwd_luma( fwdLut, c, out );
    }
  }
}

// =====================================================================================================================
// k_mc_rpr - the tiles of CUs that predict from a SCALED reference picture (reference picture resampling, vvr_picture.rpr).
//   InterPrediction::xPredInterBlkRPR (InterPrediction.cpp:2081-2217): the position in the reference picture advances by the scaling ratio per sample
//   (1/1024 sample steps from the block's origin), every column has its own integer position and horizontal phase, every row its own vertical ones;
//   ratios above 1.25 / 1.75 select the low-pass filter sets (regular CUs: m_lumaFilterRPR1/2, m_chromaFilterRPR1/2; affine sub-blocks:
//   m_affineLumaFilterRPR1/2), the MV is taken as it is (no clipMv, :650), and such a CU has no BDOF, DMVR (:1431-1435) or PROF (:1029).
//   The reference filters column by column into a buffer and row by row out of it; a sample is
//     sum_t fV[yFrac(row)][t] * (int16) ( ( sum_s fH[xFrac(col)][s] * ref( xInt(col) - (N/2-1) + s, yInt(row) - (N/2-1) + t ) + offset1 ) >> shift1 )
//   with clamped reads (= its border-extended picture; the rows it replicates below the margin, :2188-2196, are rows of the margin).
// One workgroup of 256 per <= 16x16 tile.  Plain, SbTMVP and GPM tiles: the source window of the tile in the scaled picture (its extent follows from
// the ratio: at most 39 x 39 luma samples at 2x) is staged in LDS with clamped coordinates, a horizontal pass leaves one 16-bit row of intermediates
// per window row, the vertical pass picks its rows with every output row's own offset and phase (rpr_tile).  Affine tiles - a block of its own per
// 4x4 sub-block - are evaluated sample by sample straight from the reference plane (rpr_sample): rare among the rare.
// The other list of such a CU may be an ordinary reference picture: its prediction is the regular arithmetic
// (xPredInterBlk in the separable 2-D form, which the 1-D and copy cases equal number for number: phase 0 of the regular filters is { .., 64, .. }
// and the roundings nest), with PROF where the reference applies it to that list of an affine CU.  Plain, SbTMVP, GPM, CIIP (inter part) and affine
// tiles; the combination (rounding, average, BCW, explicit weights, GPM blend) is the one of k_mc / k_mc_affine.
// =====================================================================================================================
#define MC_RPR_ONE ( 1 << 14 )
__device__ __forceinline__ const int16_t* rpr_taps( int comp, int filter, int frac, bool altHpel )
{
  if( comp ) return filter == 3 ? d_chroma_filter_rpr1[frac] : filter == 4 ? d_chroma_filter_rpr2[frac] : d_chroma_filter[frac];
  if( filter == 0 ) return ( frac == 8 && altHpel ) ? d_luma_alt_hpel : d_luma_filter[frac];
  return filter == 2 ? d_luma_filter_4x4[frac] : filter == 3 ? d_luma_filter_rpr1[frac] : filter == 4 ? d_luma_filter_rpr2[frac] : filter == 5 ? d_affine_luma_filter_rpr1[frac] : d_affine_luma_filter_rpr2[frac];
}
// what a block predicted from a scaled picture shares: filter sets, the position of its first sample in the reference picture (1/1024 sample), the steps
struct RprGeo { int xFilter, yFilter, stepX, stepY, x0, y0, rw, rh; };
__device__ __forceinline__ RprGeo rpr_geometry( const vvr_rpr_ref& rr, int winL, int winT, int comp, int bx, int by, int mvx, int mvy, int filterIndex )
{
  RprGeo g;
  const int cs = comp ? 1 : 0;
  const int thr1 = MC_RPR_ONE * 5 / 4, thr2 = MC_RPR_ONE * 7 / 4;
  const int rx = rr.ratio[0], ry = rr.ratio[1];
  g.xFilter = filterIndex; g.yFilter = filterIndex;
  if( rx > thr2 ) g.xFilter = 4; else if( rx > thr1 ) g.xFilter = 3;
  if( ry > thr2 ) g.yFilter = 4; else if( ry > thr1 ) g.yFilter = 3;
  if( !comp && filterIndex == 2 ) { if( rx > thr1 ) g.xFilter += 2; if( ry > thr1 ) g.yFilter += 2; }
  const int posShift = 10;
  g.stepX = ( rx + 8 ) >> 4; g.stepY = ( ry + 8 ) >> 4;
  const long long posX = ( ( bx << cs ) - winL ) >> cs, posY = ( ( by << cs ) - winT ) >> cs;
  const int addX = comp ? ( 1 - rr.hor_collocated_chroma ) * 8 * ( rx - MC_RPR_ONE ) : 0;
  const int addY = comp ? ( 1 - rr.ver_collocated_chroma ) * 8 * ( ry - MC_RPR_ONE ) : 0;
  long long x0 = ( posX * ( 1 << ( 4 + cs ) ) + mvx ) * (long long) rx + addX;
  x0 = ( x0 >= 0 ? 1 : -1 ) * ( ( ( x0 >= 0 ? x0 : -x0 ) + ( 1ll << ( 7 + cs ) ) ) >> ( 8 + cs ) ) + (long long) rr.win_left * ( 1 << ( posShift - cs ) );
  long long y0 = ( posY * ( 1 << ( 4 + cs ) ) + mvy ) * (long long) ry + addY;
  y0 = ( y0 >= 0 ? 1 : -1 ) * ( ( ( y0 >= 0 ? y0 : -y0 ) + ( 1ll << ( 7 + cs ) ) ) >> ( 8 + cs ) ) + (long long) rr.win_top * ( 1 << ( posShift - cs ) );
  g.x0 = (int) x0; g.y0 = (int) y0;
  g.rw = rr.width >> cs; g.rh = rr.height >> cs;
  return g;
}
// integer position and phase of column / row `i` of such a block (dir 0: columns)
__device__ __forceinline__ void rpr_position( const RprGeo& g, int comp, int dir, int i, int& pInt, int& pFrac )
{
  const int posShift = 10, shiftHor = 4 + ( comp ? 1 : 0 ), off = 1 << ( posShift - shiftHor - 1 );
  const int p = ( dir ? g.y0 + i * g.stepY : g.x0 + i * g.stepX ) + off;
  pInt = clip3( -4, ( dir ? g.rh : g.rw ) + 4, p >> posShift );
  pFrac = ( p >> ( posShift - shiftHor ) ) & ( ( 1 << shiftHor ) - 1 );
}
// sample (col, row) of the block with origin (bx, by) (component samples) predicted from the scaled picture `ref` (rw x rh component samples):
// 14-bit intermediate (bi) or rounded and clipped.  filterIndex: 0 regular CU, 2 affine sub-block.
__device__ int rpr_sample( const pel_t* __restrict__ ref, int stride, const vvr_rpr_ref& rr, int winL, int winT, int comp, int bx, int by, int col, int row,
                           int mvx, int mvy, bool bi, bool altHpel, int filterIndex, int bd )
{
  const RprGeo g = rpr_geometry( rr, winL, winT, comp, bx, by, mvx, mvy, filterIndex );
  const int ntaps = comp ? 4 : 8, half = ntaps / 2 - 1;
  const int headroom = 14 - bd > 2 ? 14 - bd : 2;
  const int shift1 = 6 - headroom, offset1 = -IF_INTERNAL_OFFS * ( 1 << shift1 );
  int xInt, xFrac, yInt, yFrac;
  rpr_position( g, comp, 0, col, xInt, xFrac ); rpr_position( g, comp, 1, row, yInt, yFrac );
  const int16_t* cv = rpr_taps( comp, g.yFilter, yFrac, altHpel && rr.ratio[1] == MC_RPR_ONE );
  const int16_t* ch = rpr_taps( comp, g.xFilter, xFrac, altHpel && rr.ratio[0] == MC_RPR_ONE );
  int sum2 = 0;
  for( int t = 0; t < ntaps; t++ )
  {
    const pel_t* line = ref + (size_t) clip3( 0, g.rh - 1, yInt - half + t ) * stride;
    int sum = 0;
    for( int u = 0; u < ntaps; u++ ) sum += line[clip3( 0, g.rw - 1, xInt - half + u )] * ch[u];
    sum2 += (int16_t) ( ( sum + offset1 ) >> shift1 ) * cv[t];
  }
  if( bi ) return (int16_t) ( sum2 >> 6 );
  const int shift2 = 6 + headroom, offset2 = ( 1 << ( shift2 - 1 ) ) + ( IF_INTERNAL_OFFS << 6 );
  return clip_pel( (int16_t) ( ( sum2 + offset2 ) >> shift2 ), bd );
}
// The same for a whole w x h part (columns col0 .., rows row0 .. of the block) by the 256 work items of a workgroup, through LDS: the source window
// of the part - its extent follows from the ratio, at most 39 x 39 samples for 16 x 16 at 2x - is staged once with clamped coordinates, the horizontal
// pass writes one 16-bit row of intermediates per window row, the vertical pass reads them back with every row's own offset and phase.  out[row * w + col].
#define RPR_WIN 40
struct RprShared {
  pel_t   win[RPR_WIN * RPR_WIN];
  int16_t tmp[RPR_WIN * 16];
  int16_t pi[2][16], pf[2][16];          // [columns, rows]: integer position, phase
  int16_t out[2][3][16 * 16];            // [list][component]: 14-bit (or final) prediction of the tile
};
__device__ void rpr_tile( RprShared& sh, const pel_t* __restrict__ ref, int stride, const vvr_rpr_ref& rr, int winL, int winT, int comp, int bx, int by, int col0, int row0, int w, int h,
                          int mvx, int mvy, bool bi, bool altHpel, int bd, int16_t* __restrict__ out )
{
  const int tid = threadIdx.x;
  const RprGeo g = rpr_geometry( rr, winL, winT, comp, bx, by, mvx, mvy, 0 );
  const int ntaps = comp ? 4 : 8, half = ntaps / 2 - 1;
  const int headroom = 14 - bd > 2 ? 14 - bd : 2;
  const int shift1 = 6 - headroom, offset1 = -IF_INTERNAL_OFFS * ( 1 << shift1 );
  if( tid < 32 )
  {
    const int dir = tid >> 4, i = tid & 15;
    if( i < ( dir ? h : w ) ) { int pI, pF; rpr_position( g, comp, dir, ( dir ? row0 : col0 ) + i, pI, pF ); sh.pi[dir][i] = (int16_t) pI; sh.pf[dir][i] = (int16_t) pF; }
  }
  __syncthreads();
  // (positions do not decrease with the column / row: the steps are positive and the clip is monotonous)
  const int wx0 = sh.pi[0][0] - half, wy0 = sh.pi[1][0] - half, ww = sh.pi[0][w - 1] - sh.pi[0][0] + ntaps, wh = sh.pi[1][h - 1] - sh.pi[1][0] + ntaps;
  if( ww > RPR_WIN || wh > RPR_WIN )
  {   // (cannot happen with ratios of at most 2; kept for a table that says otherwise)
    for( int i = tid; i < w * h; i += 256 ) out[i] = (int16_t) rpr_sample( ref, stride, rr, winL, winT, comp, bx, by, col0 + i % w, row0 + i / w, mvx, mvy, bi, altHpel, 0, bd );
    __syncthreads();
    return;
  }
  for( int i = tid; i < ww * wh; i += 256 )
  {
    const int y = i / ww, x = i - y * ww;
    sh.win[y * RPR_WIN + x] = ref[(size_t) clip3( 0, g.rh - 1, wy0 + y ) * stride + clip3( 0, g.rw - 1, wx0 + x )];
  }
  __syncthreads();
  const bool altX = altHpel && rr.ratio[0] == MC_RPR_ONE, altY = altHpel && rr.ratio[1] == MC_RPR_ONE;
  for( int i = tid; i < wh * w; i += 256 )
  {
    const int y = i / w, col = i - y * w;
    const int16_t* ch = rpr_taps( comp, g.xFilter, sh.pf[0][col], altX );
    const pel_t* src = &sh.win[y * RPR_WIN + sh.pi[0][col] - sh.pi[0][0]];
    int sum = 0;
    for( int u = 0; u < ntaps; u++ ) sum += src[u] * ch[u];
    sh.tmp[y * 16 + col] = (int16_t) ( ( sum + offset1 ) >> shift1 );
  }
  __syncthreads();
  const int shift2 = 6 + headroom, offset2 = ( 1 << ( shift2 - 1 ) ) + ( IF_INTERNAL_OFFS << 6 );
  for( int i = tid; i < w * h; i += 256 )
  {
    const int row = i / w, col = i - row * w;
    const int16_t* cv = rpr_taps( comp, g.yFilter, sh.pf[1][row], altY );
    const int16_t* src = &sh.tmp[( sh.pi[1][row] - sh.pi[1][0] ) * 16 + col];
    int sum2 = 0;
    for( int t = 0; t < ntaps; t++ ) sum2 += src[t * 16] * cv[t];
    out[i] = bi ? (int16_t) ( sum2 >> 6 ) : (int16_t) clip_pel( (int16_t) ( ( sum2 + offset2 ) >> shift2 ), bd );
  }
  __syncthreads();
}
// the sample at (px, py) of the component plane predicted with the (clipped) MV from an ordinary reference picture of the current picture's size
__device__ int reg_sample( const pel_t* __restrict__ ref, int stride, int pw, int ph, int comp, int px, int py, int mvx, int mvy, bool bi, bool altHpel, bool sixTap, int bd )
{
  const int sh = 4 + ( comp ? 1 : 0 ), ntaps = comp ? 4 : 8, half = ntaps / 2 - 1;
  const int xFrac = mvx & ( ( 1 << sh ) - 1 ), yFrac = mvy & ( ( 1 << sh ) - 1 ), x0 = px + ( mvx >> sh ), y0 = py + ( mvy >> sh );
  const int16_t* ch = comp ? d_chroma_filter[xFrac] : ( xFrac == 8 && altHpel ) ? d_luma_alt_hpel : sixTap ? d_luma_filter_4x4[xFrac] : d_luma_filter[xFrac];
  const int16_t* cv = comp ? d_chroma_filter[yFrac] : ( yFrac == 8 && altHpel ) ? d_luma_alt_hpel : sixTap ? d_luma_filter_4x4[yFrac] : d_luma_filter[yFrac];
  const int headroom = 14 - bd > 2 ? 14 - bd : 2;
  const int shift1 = 6 - headroom, offset1 = -IF_INTERNAL_OFFS * ( 1 << shift1 );
  int sum2 = 0;
  for( int t = 0; t < ntaps; t++ )
  {
    const pel_t* line = ref + (size_t) clip3( 0, ph - 1, y0 - half + t ) * stride;
    int sum = 0;
    for( int u = 0; u < ntaps; u++ ) sum += line[clip3( 0, pw - 1, x0 - half + u )] * ch[u];
    sum2 += (int16_t) ( ( sum + offset1 ) >> shift1 ) * cv[t];
  }
  if( bi ) return (int16_t) ( sum2 >> 6 );
  const int shift2 = 6 + headroom, offset2 = ( 1 << ( shift2 - 1 ) ) + ( IF_INTERNAL_OFFS << 6 );
  return clip_pel( (int16_t) ( ( sum2 + offset2 ) >> shift2 ), bd );
}
// sample (px, py) of the 4x4 luma sub-block at (sbx, sby) of an affine CU refined by PROF (applyPROFCore, InterPrediction.cpp:61; the 6x6 block of
// xPredAffineBlk :1224-1290: the prediction inside, integer reference samples around it), from an ordinary reference picture
__device__ int aff_prof_sample( const pel_t* __restrict__ ref, int stride, int pw, int ph, int sbx, int sby, int px, int py, int mx, int my, int dMvH, int dMvV, bool bi, int bd )
{
  const int headroom = 14 - bd > 2 ? 14 - bd : 2;
  auto ext = [&]( int i, int j ) -> int
  {
    if( i >= 1 && i <= 4 && j >= 1 && j <= 4 ) return reg_sample( ref, stride, pw, ph, 0, sbx + i - 1, sby + j - 1, mx, my, true, false, true, bd );
    const int x = clip3( 0, pw - 1, sbx + ( mx >> 4 ) + i - 1 + ( ( mx & 15 ) >> 3 ) ), y = clip3( 0, ph - 1, sby + ( my >> 4 ) + j - 1 + ( ( my & 15 ) >> 3 ) );
    return (int16_t) ( (int16_t) ( ref[(size_t) y * stride + x] << headroom ) - (int16_t) IF_INTERNAL_OFFS );
  };
  const int gX = (int16_t) ( ( ext( 2 + px, 1 + py ) >> 6 ) - ( ext( px, 1 + py ) >> 6 ) ), gY = (int16_t) ( ( ext( 1 + px, 2 + py ) >> 6 ) - ( ext( 1 + px, py ) >> 6 ) );
  const int dILimit = 1 << max( bd + 1, 13 );
  const int dI = clip3( -dILimit, dILimit - 1, dMvH * gX + dMvV * gY );
  int v = (int16_t) ( ext( 1 + px, 1 + py ) + dI );
  if( !bi ) { v = (int16_t) ( ( v + ( 1 << ( headroom - 1 ) ) + IF_INTERNAL_OFFS ) >> headroom ); v = clip_pel( v, bd ); }
  return v;
}

__global__ __launch_bounds__( 256 ) void k_mc_rpr( PicDev pic, RefSet refs, DevPlanes reco, const McItem* __restrict__ items, int numItems )
{
  __shared__ RprShared sh;
  const int item = blockIdx.x;
  if( item >= numItems ) return;
  const McItem it = items[item];
  const vvr_cu& cu = pic.cu[it.cu];
  const vvr_rpr_params& R = *pic.rpr;
  const int16_t* __restrict__ fwdLut = lmcs_fwd_at( pic, it.x, it.y );
  const int bd = pic.hdr.bit_depth, ctu = 1 << pic.hdr.log2_ctu;
  const int ncomp = pic.hdr.chroma_format ? 3 : 1;
  const int headroom = 14 - bd > 2 ? 14 - bd : 2;
  const bool aff = ( it.flags & MC_ITEM_AFFINE ) != 0, geo = ( it.flags & MC_ITEM_GEO ) != 0;
  const bool altHpel = ( it.flags & MC_ITEM_HPEL ) != 0;
  const vvr_wp_params* __restrict__ wpT = wp_at( pic, it.x, it.y );
  // (two scalars and selects, not an array indexed with the list: a dynamically indexed local array - and the McItem record with it - lives in scratch memory)
  const int mRef0 = aff ? cu.ref_idx[0] : it.ref[0], mRef1 = aff ? cu.ref_idx[1] : it.ref[1];
  bool uni;
  if( aff )
  {
    bool biPred = mRef0 >= 0 && mRef1 >= 0;
    // xCheckIdenticalMotion (:404-436)
    if( biPred && pic.hdr.ref_poc[0][mRef0] == pic.hdr.ref_poc[1][mRef1]
        && cu.mv[0][0][0] == cu.mv[1][0][0] && cu.mv[0][0][1] == cu.mv[1][0][1] && cu.mv[0][1][0] == cu.mv[1][1][0] && cu.mv[0][1][1] == cu.mv[1][1][1]
        && ( !( cu.flags & VVR_CU_AFFINE_6P ) || ( cu.mv[0][2][0] == cu.mv[1][2][0] && cu.mv[0][2][1] == cu.mv[1][2][1] ) ) && !pic.wp ) biPred = false;
    uni = !biPred;
  }
  else uni = ( it.flags & MC_ITEM_UNI ) != 0;
  const int l0 = uni ? ( mRef0 >= 0 ? 0 : 1 ) : 0, nl = uni ? 1 : 2;
  const int bcw = aff ? cu.bcw_idx : it.bcw;
  const bool wpOn = wpT && !geo && bcw == 2;
  const bool hi = !uni || wpOn;
  // affine: clip range of the sub-block MVs (:1194-1197), PROF per list (:1015-1029) with the MV offset of this work item's position in a sub-block
  const int horMax = ( pic.hdr.width + 8 - cu.x - 1 ) * 16, horMin = ( -ctu - 8 - cu.x + 1 ) * 16;
  const int verMax = ( pic.hdr.height + 8 - cu.y - 1 ) * 16, verMin = ( -ctu - 8 - cu.y + 1 ) * 16;
  const int wL = it.w, hL = it.h, wC = wL >> 1, hC = hL >> 1;
  const int total = wL * hL + ( ncomp == 3 ? 2 * wC * hC : 0 );
  // plain, SbTMVP and GPM tiles: the prediction of every (list, component) first - from a scaled picture through LDS (rpr_tile), from an ordinary one
  // sample by sample -, combined below; affine tiles (a block of their own per 4x4 sub-block) are evaluated sample by sample in the loop below
  if( !aff )
  {
    for( int k = 0; k < nl; k++ )
    {
      const int l = geo ? ( cu.geo_dir_ref[k] >> 4 ) - 1 : uni ? l0 : k;
      const int ri = geo ? ( cu.geo_dir_ref[k] & 15 ) : ( l ? mRef1 : mRef0 );
      const vvr_rpr_ref& rr = R.ref[l][ri];
      const bool bi = hi || geo;
      int mvx = geo ? cu.geo_mv[k][0] : ( l ? it.mv[1][0] : it.mv[0][0] ), mvy = geo ? cu.geo_mv[k][1] : ( l ? it.mv[1][1] : it.mv[0][1] );
      if( !rr.scaled ) { const McBounds B = { 0, 0, (int) pic.hdr.width - 1, (int) pic.hdr.height - 1 }; mc_clip_mv( pic, B, it.clipX, it.clipY, mvx, mvy ); }
      for( int c = 0; c < ncomp; c++ )
      {
        const int cs = c ? 1 : 0, cw = wL >> cs, chh = hL >> cs;
        const pel_t* __restrict__ plane = refs.p[l * VVR_MAX_REFS + ri][c];
        if( rr.scaled ) rpr_tile( sh, plane, reco.stride[c], rr, R.win_left, R.win_top, c, it.clipX >> cs, it.clipY >> cs, ( it.x - it.clipX ) >> cs, ( it.y - it.clipY ) >> cs, cw, chh, mvx, mvy, bi, altHpel, bd, sh.out[k][c] );
        else for( int i = threadIdx.x; i < cw * chh; i += 256 ) sh.out[k][c][i] = (int16_t) reg_sample( plane, reco.stride[c], reco.w[c], reco.h[c], c, ( it.x >> cs ) + i % cw, ( it.y >> cs ) + i / cw, mvx, mvy, bi, altHpel, false, bd );
      }
    }
    __syncthreads();
  }
  for( int sidx = threadIdx.x; sidx < total; sidx += 256 )
  {
    int c, x, y;
    if( sidx < wL * hL ) { c = 0; y = sidx / wL; x = sidx - y * wL; }
    else { const int q = sidx - wL * hL; c = 1 + ( q >= wC * hC ); const int r = q - ( c - 1 ) * wC * hC; y = r / wC; x = r - y * wC; }
    const int cs = c ? 1 : 0;
    const int ax = ( it.x >> cs ) + x, ay = ( it.y >> cs ) + y;       // position in the component plane
    int p[2] = { 0, 0 };
    for( int k = 0; k < nl; k++ )
    {
      const int l = geo ? ( cu.geo_dir_ref[k] >> 4 ) - 1 : uni ? l0 : k;
      const int ri = geo ? ( cu.geo_dir_ref[k] & 15 ) : ( l ? mRef1 : mRef0 );
      const vvr_rpr_ref& rr = R.ref[l][ri];
      const pel_t* __restrict__ plane = refs.p[l * VVR_MAX_REFS + ri][c];
      const bool bi = hi || geo;
      if( !aff ) p[k] = sh.out[k][c][y * ( c ? wC : wL ) + x];
      else
      {
        // the MV of the 4x4 sub-block (luma), or of the 4x4 chroma block from the luma sub-blocks (0,0) and (1,1) of its 2x2 group (:1156-1176)
        const int sx = x >> 2, sy = y >> 2;
        const bool onDev = ( pic.hdr.tool_flags & VVR_TOOL_AFFINE_MV_ON_DEVICE ) != 0;
        int mx, my;
        if( c == 0 )
        {
          if( onDev ) aff_span_mv( cu, l, ( ( it.x - cu.x ) >> 2 ) + sx, ( ( it.y - cu.y ) >> 2 ) + sy, mx, my );
          else { const vvr_motion& m = pic.affMotion[it.mv[0][0] + 4 * sy + sx]; mx = m.mv[l][0]; my = m.mv[l][1]; }
        }
        else
        {
          if( onDev )
          {
            int a0, a1, b0, b1;
            aff_span_mv( cu, l, ( ( it.x - cu.x ) >> 2 ) + 2 * sx, ( ( it.y - cu.y ) >> 2 ) + 2 * sy, a0, a1 );
            aff_span_mv( cu, l, ( ( it.x - cu.x ) >> 2 ) + 2 * sx + 1, ( ( it.y - cu.y ) >> 2 ) + 2 * sy + 1, b0, b1 );
            mx = a0 + b0; my = a1 + b1;
          }
          else
          {
            const vvr_motion& m0 = pic.affMotion[it.mv[0][0] + 4 * ( 2 * sy ) + 2 * sx];
            const vvr_motion& m1 = pic.affMotion[it.mv[0][0] + 4 * ( 2 * sy + 1 ) + 2 * sx + 1];
            mx = m0.mv[l][0] + m1.mv[l][0]; my = m0.mv[l][1] + m1.mv[l][1];
          }
          aff_round_mv( mx, my, 1 );
        }
        mx = min( horMax, max( horMin, mx ) ); my = min( verMax, max( verMin, my ) );
        const int bx = ( it.x >> cs ) + 4 * sx, by = ( it.y >> cs ) + 4 * sy;
        if( rr.scaled ) p[k] = rpr_sample( plane, reco.stride[c], rr, R.win_left, R.win_top, c, bx, by, x & 3, y & 3, mx, my, bi, false, 2, bd );
        else if( c ) p[k] = reg_sample( plane, reco.stride[c], reco.w[c], reco.h[c], c, ax, ay, mx, my, bi, false, false, bd );
        else
        {
          const int lw = ilog2( cu.w ), lh = ilog2( cu.h );
          const int dHX = ( cu.mv[l][1][0] - cu.mv[l][0][0] ) * ( 1 << ( 7 - lw ) ), dHY = ( cu.mv[l][1][1] - cu.mv[l][0][1] ) * ( 1 << ( 7 - lw ) );
          int dVX, dVY;
          const bool sixP = ( cu.flags & VVR_CU_AFFINE_6P ) != 0;
          if( sixP ) { dVX = ( cu.mv[l][2][0] - cu.mv[l][0][0] ) * ( 1 << ( 7 - lh ) ); dVY = ( cu.mv[l][2][1] - cu.mv[l][0][1] ) * ( 1 << ( 7 - lh ) ); }
          else { dVX = -dHY; dVY = dHX; }
          const bool eqRT = cu.mv[l][0][0] == cu.mv[l][1][0] && cu.mv[l][0][1] == cu.mv[l][1][1];
          const bool eqLB = cu.mv[l][0][0] == cu.mv[l][2][0] && cu.mv[l][0][1] == cu.mv[l][2][1];
          const bool prof = ( pic.hdr.tool_flags & VVR_TOOL_PROF ) && !( ( sixP && eqRT && eqLB ) || ( !sixP && eqRT ) ) && !aff_spread_over_limit( dHX, dHY, dVX, dVY, cu.inter_dir );
          if( prof )
          {
            const int qHX = dHX * 4, qHY = dHY * 4, qVX = dVX * 4, qVY = dVY * 4;
            int a = ( ( dHX + dVX ) * 2 ) - ( ( qHX + qVX ) * 2 ) + ( x & 3 ) * qHX + ( y & 3 ) * qVX, b = ( ( dHY + dVY ) * 2 ) - ( ( qHY + qVY ) * 2 ) + ( x & 3 ) * qHY + ( y & 3 ) * qVY;
            aff_round_mv( a, b, 8 );
            p[k] = aff_prof_sample( plane, reco.stride[0], reco.w[0], reco.h[0], bx, by, x & 3, y & 3, mx, my, clip3( -31, 31, a ), clip3( -31, 31, b ), bi, bd );
          }
          else p[k] = reg_sample( plane, reco.stride[0], reco.w[0], reco.h[0], 0, ax, ay, mx, my, bi, false, true, bd );
        }
      }
    }
    int out = p[0];
    if( geo )
    {
      const int MS = 112;
      const int angle = d_geo_params[cu.geo_split_dir][0];
      const int wIdx = ilog2( cu.w ) - 3, hIdx = ilog2( cu.h ) - 3;
      const int ox = d_geo_weight_offset[cu.geo_split_dir][hIdx][wIdx][0], oy = d_geo_weight_offset[cu.geo_split_dir][hIdx][wIdx][1];
      const int8_t* gW = d_geo_weights[d_geo_angle2mask[angle]];
      const int mir = d_geo_angle2mirror[angle];
      const int lx = ( ax << cs ) - cu.x, ly = ( ay << cs ) - cu.y;
      const int wt = mir == 2 ? gW[( MS - 1 - oy - ly ) * MS + ox + lx] : mir == 1 ? gW[( oy + ly ) * MS + ( MS - 1 - ox ) - lx] : gW[( oy + ly ) * MS + ox + lx];
      const int shift = headroom + 3, offset = ( 1 << ( shift - 1 ) ) + ( IF_INTERNAL_OFFS << 3 );
      out = clip_pel( ( wt * p[0] + ( 8 - wt ) * p[1] + offset ) >> shift, bd );
    }
    else if( wpOn ) out = uni ? wp_uni( wpT, l0, l0 ? mRef1 : mRef0, c, p[0], bd, headroom ) : wp_bi( wpT, mRef0, mRef1, c, p[0], p[1], bd, headroom );
    else if( !uni )
    {
      if( bcw != 2 )
      {
        const int w1 = d_bcw_weights[bcw], w0 = 8 - w1, shift = headroom + 3, offset = ( 1 << ( shift - 1 ) ) + ( IF_INTERNAL_OFFS << 3 );
        out = clip_pel( ( p[0] * w0 + p[1] * w1 + offset ) >> shift, bd );
      }
      else
      {
        const int shift = headroom + 1, offset = ( 1 << ( shift - 1 ) ) + 2 * IF_INTERNAL_OFFS;
        out = clip_pel( ( p[0] + p[1] + offset ) >> shift, bd );
      }
    }
    reco.p[c][(size_t) ay * reco.stride[c] + ax] = (pel_t) lmcs_fwd_luma( fwdLut, c, out );
  }
}

void launch_mc_rpr( hipStream_t s, const PicDev& pic, const RefSet& refs, DevPlanes reco, const McItem* items, int numItems )
{
  if( !numItems ) return;
  hipLaunchKernelGGL( k_mc_rpr, dim3( numItems ), dim3( 256 ), 0, s, pic, refs, reco, items, numItems );
}

void launch_mc( hipStream_t s, const PicDev& pic, const RefSet& refs, DevPlanes reco, const McItem* items, int numItems, const McItem* items2, int numItems2, int bdof )
{
  if( !( numItems + numItems2 ) ) return;
  // (one wavefront per tile; two were measured: 123 instead of 106 us per 4K B picture - the stages of a 16x16 tile are 136 and 48 work items)
  if( bdof ) hipLaunchKernelGGL( ( k_mc<64, true> ),  dim3( numItems + numItems2 ), dim3( 64 ), 0, s, pic, refs, reco, items, numItems, items2, numItems2 );
  else       hipLaunchKernelGGL( ( k_mc<64, false> ), dim3( numItems + numItems2 ), dim3( 64 ), 0, s, pic, refs, reco, items, numItems, items2, numItems2 );
}

// =====================================================================================================================
// prep_expand_mc (a section of k_prep) - the motion-compensation tiles of the CUs whose tiles are a function of the CU record alone (plain, BDOF, DMVR): the records of a
// CU's <= 16x16 tiles, written where the host reserved room for them (the host only counted them).  Same records as
// PrepScratch::buildWorkLists writes for the tiles it still writes itself.
// =====================================================================================================================
__device__ __forceinline__ void prep_expand_mc( int bid, const vvr_cu* __restrict__ cus, const McCuRef* __restrict__ refs, int numRefs, McItem* __restrict__ plain, McItem* __restrict__ bdof, McItem* __restrict__ dmvr )
{
  // one wavefront per CU, one lane per tile (a 128x128 CU has 64 of them)
  const int i = bid * 4 + ( threadIdx.x >> 6 ), t = threadIdx.x & 63;
  if( i >= numRefs ) return;
  const McCuRef r = refs[i];
  const vvr_cu& cu = cus[r.cu];
  const int cw = cu.w, ch = cu.h, tilesX = ( cw + 15 ) >> 4, nt = tilesX * ( ( ch + 15 ) >> 4 );
  if( t >= nt ) return;
  const int cls = (int) ( r.first >> 30 );
  McItem it;
  it.flags = 0; it.cu = r.cu;
  it.mv[0][0] = it.mv[0][1] = it.mv[1][0] = it.mv[1][1] = 0; it.ref[0] = it.ref[1] = 0; it.bcw = 0; it.clipW4 = 0; it.clipX = it.clipY = 0;
  {
    // everything k_mc needs about the motion of the tile (k_mc_dmvr: the start vectors and what they are clipped against)
    it.ref[0] = cu.ref_idx[0]; it.ref[1] = cu.ref_idx[1];
    it.mv[0][0] = cu.mv[0][0][0]; it.mv[0][1] = cu.mv[0][0][1]; it.mv[1][0] = cu.mv[1][0][0]; it.mv[1][1] = cu.mv[1][0][1];
    it.clipX = cu.x; it.clipY = cu.y;
    it.bcw = cu.bcw_idx;
    it.flags = (uint16_t) ( ( cu.mc_mode == VVR_MC_UNI ? MC_ITEM_UNI : 0 ) | ( cu.imv == 3 ? MC_ITEM_HPEL : 0 ) | ( cu.mc_mode == VVR_MC_GEO ? MC_ITEM_GEO : 0 ) );
    if( cls == 2 ) it.clipW4 = (uint8_t) ( cw >> 2 );       // (the start vectors of a DMVR CU are clipped against the CU: its width under reference wrap-around)
  }
  const int ty = t / tilesX, tx = t - ty * tilesX, x = tx << 4, y = ty << 4;          // (rows of tiles, as the host writes them)
  it.x = (uint16_t) ( cu.x + x ); it.y = (uint16_t) ( cu.y + y ); it.w = (uint8_t) min( 16, cw - x ); it.h = (uint8_t) min( 16, ch - y );
  ( cls == 0 ? plain : cls == 1 ? bdof : dmvr )[( r.first & 0x3fffffffu ) + t] = it;
}

void launch_mc_affine( hipStream_t s, const PicDev& pic, const RefSet& refs, DevPlanes reco, const McItem* items, int numItems )
{
  if( !numItems ) return;
  // four wavefronts per tile: every stage is a loop over independent elements (1056 window samples, 704 first-stage samples per list of a
  // 16x16 tile), and the 18 KB of LDS per tile allow 8 tiles per CU whatever the workgroup size - with one wavefront per tile that is 2
  // wavefronts per SIMD, each walking through 66 dependent rounds of scattered 2-byte loads
  hipLaunchKernelGGL( k_mc_affine<256>, dim3( numItems ), dim3( 256 ), 0, s, pic, refs, reco, items, numItems );
}

void launch_mc_dmvr( hipStream_t s, const PicDev& pic, const RefSet& refs, DevPlanes reco, const McItem* items, int numItems, int32_t* dmvrOut )
{
  if( !numItems ) return;
  // two wavefronts per sub-block: 52.5 us per 4K B picture alone against 59.8 with one and 58.3 with four (the search's serial decisions and the
  // barriers between the stages weigh more with four; profiles/round3_lanes_and_host_threads.txt)
  hipLaunchKernelGGL( k_mc_dmvr<128>, dim3( numItems ), dim3( 128 ), 0, s, pic, refs, reco, items, numItems, dmvrOut );
}

// =====================================================================================================================
// k_itrans — dequantisation + LFNST + 2-D inverse transform + residual add / store, one transform block per workgroup.
//   Quant::dequant (Quant.cpp:295), invLfnstNxNCore/xInvLfnst (TrQuant.cpp:79,201), TrQuant::xIT (:410),
//   fastInvCore_ (TrQuant_EMT.cpp:389), cpyResiClip (:366), xITransformSkip (TrQuant.cpp:489), invTransformICT (:320),
//   AreaBuf::reconstruct (Buffer.cpp:482).
// Coefficients are dequantised straight into LDS; both 1-D passes run out of LDS.
// =====================================================================================================================
__device__ __forceinline__ const int16_t* tr_matrix( int type, int n )
{
  if( type == 0 ) { switch( n ) { case 2: return d_dct2_2; case 4: return d_dct2_4; case 8: return d_dct2_8; case 16: return d_dct2_16; case 32: return d_dct2_32; default: return d_dct2_64; } }
  if( type == 1 ) { switch( n ) { case 4: return d_dct8_4; case 8: return d_dct8_8; case 16: return d_dct8_16; default: return d_dct8_32; } }
  switch( n ) { case 4: return d_dst7_4; case 8: return d_dst7_8; case 16: return d_dst7_16; default: return d_dst7_32; }
}

__device__ __forceinline__ int wide_angle_mode( int w, int h, int mode )   // PU::getWideAngIntraMode (UnitTools.cpp:617)
{
  const int modeShift[6] = { 0, 6, 10, 12, 14, 15 };
  if( mode < 2 ) return mode;
  const int d = iabs( ilog2( w ) - ilog2( h ) );
  if( w > h && mode < 2 + modeShift[d] ) mode += 65;
  else if( h > w && mode > 66 - modeShift[d] ) mode -= 67;
  return mode;
}

// ---------------------------------------------------------------------------------------------------------------------
// LMCS chroma residual scaling: Reshape::calculateChromaAdjVpduNei (Reshape.cpp:192-274) — the factor of a VPDU is looked up from
// the mean of the reconstructed (mapped-domain) luma samples left of and above the CU at the VPDU origin; AreaBuf::scaleSignal
// (Buffer.cpp:412) applies it.  The neighbourhood descriptor comes from the host glue (pic.csVpdu); `acc` is an LDS word.
// Must be called by all threads of the workgroup.
// ---------------------------------------------------------------------------------------------------------------------
// what k_intra reads of a picture: a kernel that keeps a long list of uniform values per block cannot afford the whole PicDev in scalar
// registers (it spilled them into vector lanes on its serial path)
struct IntraPic {
  pel_t*       plane[3];
  const pel_t* resi[3];
  int          stride[3], rstride[3], w[3], h[3];
  const uint32_t* csVpdu;            // as PicDev
  const vvr_lmcs_params* lmcs;
  int vpdusX, vpduLog2, ctusX, log2Ctu, bitDepth, width, height, colloc;
};
// Loads of samples that another workgroup of the SAME launch may have written (k_intra_leaf): device scope (sc1) - they are served by L2 / memory, never
// by this CU's vector L1, which another CU's stores do not refresh; the writer stores with sc1 (write-through) as well (MI355X_MICROARCH.md, inter-workgroup
// visibility: "sc1 stores AND sc1 loads").  SC1 = false: the plain load every other kernel uses.
// eight samples (16 bytes, 8-byte aligned), device scope when SC1: two 8-byte loads that bypass the vector L1 (what another workgroup of the launch stored write-through)
template<bool SC1> __device__ __forceinline__ uint4 ld_pel8( const pel_t* p )
{
  if( !SC1 ) return *reinterpret_cast<const uint4*>( p );
  const unsigned long long a = __hip_atomic_load( reinterpret_cast<const unsigned long long*>( p ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
  const unsigned long long b = __hip_atomic_load( reinterpret_cast<const unsigned long long*>( p ) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
  return make_uint4( (uint32_t) a, (uint32_t) ( a >> 32 ), (uint32_t) b, (uint32_t) ( b >> 32 ) );
}
template<bool SC1> __device__ __forceinline__ int ld_pel( const pel_t* p )
{
  if constexpr( SC1 ) return (int) (int16_t) __hip_atomic_load( reinterpret_cast<uint16_t*>( const_cast<pel_t*>( p ) ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
  else return (int) *p;
}
template<bool SC1> __device__ __forceinline__ uint32_t ld_pel2( const pel_t* p )      // two neighbouring samples (4-byte aligned)
{
  if constexpr( SC1 ) return __hip_atomic_load( reinterpret_cast<uint32_t*>( const_cast<pel_t*>( p ) ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
  else return *reinterpret_cast<const uint32_t*>( p );
}
template<bool SC1 = false>
__device__ __forceinline__ int lmcs_cscale_factor_wave( const IntraPic& pic, int lumaX, int lumaY, int lane )
{
  // called by one whole wavefront; every lane returns the factor.  All table reads are issued up front, one entry per lane, so the
  // function costs two memory round trips (VPDU record, then luma samples + tables) instead of one per pivot of the search
  const uint32_t d = pic.csVpdu[( lumaY >> pic.vpduLog2 ) * pic.vpdusX + ( lumaX >> pic.vpduLog2 )];
  const int pivotL = pic.lmcs->pivot[min( lane + 1, 16 )];                 // pivot[idx + 1] for idx = lane
  const int scaleL = pic.lmcs->chroma_scale[min( lane, 15 )];
  const int minBin = pic.lmcs->min_bin, maxBin = pic.lmcs->max_bin;
  const int xPos = d & 0x1fff, yPos = ( d >> 13 ) & 0x1fff;
  const bool hasLeft = ( d >> 26 ) & 1, hasAbove = ( d >> 27 ) & 1;
  const int n = 1 << pic.vpduLog2, nLog = pic.vpduLog2;
  const pel_t* __restrict__ Y = pic.plane[0]; const int st = pic.stride[0];
  int part = 0;
  for( int t = lane; t < 2 * n; t += 64 )
  {
    const int side = t >= n, i = t - side * n;
    if( !side && hasLeft )  part += ld_pel<SC1>( &Y[(size_t) ( yPos + min( i, pic.height - yPos - 1 ) ) * st + xPos - 1] );
    if( side && hasAbove )  part += ld_pel<SC1>( &Y[(size_t) ( yPos - 1 ) * st + xPos + min( i, pic.width - xPos - 1 )] );
  }
#pragma unroll
  for( int off = 32; off >= 1; off >>= 1 ) part += __shfl_xor( part, off, 64 );
  const int recLuma = part;
  int lumaValue;
  if( hasLeft && hasAbove ) lumaValue = ( recLuma + ( 1 << nLog ) ) >> ( nLog + 1 );
  else if( hasLeft || hasAbove ) lumaValue = ( recLuma + ( 1 << ( nLog - 1 ) ) ) >> nLog;
  else lumaValue = 1 << ( pic.bitDepth - 1 );
  // first idx in [minBin, maxBin] with lumaValue < pivot[idx + 1], else maxBin + 1 (Reshape::getPWLIdxInv, :280); table entry min( idx, 15 )
  const unsigned long long hit = __ballot( lane >= minBin && lane <= maxBin && lumaValue < pivotL );
  const int idx = hit ? __builtin_ctzll( hit ) : maxBin + 1;
  return __shfl( scaleL, min( idx, 15 ), 64 );
}
__device__ __forceinline__ int lmcs_scale_resi( int r, int scale, int bd )
{
  const int maxAbs = ( 1 << bd ) - 1;
  int v = clip3( -maxAbs - 1, maxAbs, r );
  const int sign = v >= 0 ? 1 : -1, a = sign * v;
  v = sign * ( ( a * scale + ( 1 << 10 ) ) >> 11 );
  return clip3( -32768, 32767, v );
}

// Explicit scaling lists: the matrix entry at (x, y) of a (1 << lw) x (1 << lh) block, i.e. what the reference's expanded tables hold
// (Quant::setScalingListDec / processScalingListDec, Quant.cpp:386-570; list ids g_scalingListId, Rom.cpp:504)
__device__ __forceinline__ int scaling_entry( const vvr_scaling_list* __restrict__ sl, int listType, int lw, int lh, int x, int y )
{
  const int large = max( lw, lh );
  // g_scalingListId[large][listType]
  int id;
  if( large == 1 ) id = listType == 5 ? 1 : 0;
  else if( large < 6 ) id = 2 + 6 * ( large - 2 ) + listType;
  else id = listType == 0 ? 26 : listType == 3 ? 27 : 20 + listType;
  if( lw == lh )
  {
    const int sl2 = min( lw, 3 ), rl2 = lw - sl2;
    if( rl2 > 0 && x == 0 && y == 0 ) return sl->dc[id];
    return sl->coef[id][( ( y >> rl2 ) << sl2 ) + ( x >> rl2 )];
  }
  const int sl2 = large >= 3 ? 3 : 2;
  if( large > 3 && x == 0 && y == 0 ) return sl->dc[id];
  if( lh > lw ) { const int rWH = lh - lw, rH = lh - sl2; return sl->coef[id][( ( y >> rH ) << sl2 ) + ( ( x << rWH ) >> rH )]; }
  const int rWH = lw - lh, rW = lw - sl2;
  return sl->coef[id][( ( ( y << rWH ) >> rW ) << sl2 ) + ( x >> rW )];
}

// NT threads per transform block: 64 for the <= 16x16 class (one wavefront per block: four times as many blocks resident, no
// cross-wave barrier), 256 for the larger classes.
// Round 6: three memory round trips per block - the item; its TU record; then EVERYTHING else at once: the CU record, the coded levels, the basis rows of both
// passes and, for a block that is added onto the prediction, the prediction samples its lanes are going to change - instead of seven one after the other (item, TU,
// CU, levels, basis rows, and at the very end a read of the prediction for every group of stores).  What the loads depend on of the CU - a chroma block that takes
// its geometry from an ISP CU, BDPCM, LFNST - the host says in the item (TbItem::pad), so that only such a block waits for the CU record before it asks for the rest.
template<int MAXN, int NT>
__global__ __launch_bounds__( NT ) __attribute__( ( amdgpu_waves_per_eu( MAXN > 32 ? 3 : 8, 8 ) ) ) void k_itrans( PicDev pic, DevPlanes reco, DevPlanes resi, const TbItem* __restrict__ items, int numItems )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) int32_t dq[MAXN * MAXN];      // (16-byte reads in the two passes)
  __shared__ __attribute__( ( aligned( 16 ) ) ) int32_t tmp[MAXN * MAXN];
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t mvS[MAXN * MAXN], mhS[MAXN * MAXN];
  __shared__ int32_t lf_in[16], lf_out[48];
  const int item = blockIdx.x;
  if( item >= numItems ) return;
  const TbItem it = items[item];
  const vvr_tu& tu = pic.tu[it.tu];
  const int comp = it.comp, bd = pic.hdr.bit_depth, tid = threadIdx.x;
  const int csh = comp ? 1 : 0;
  // ---- the TU record: every field the block needs, asked for together
  const int tuX = tu.x, tuY = tu.y, tuW = tu.w, tuH = tu.h, mts = tu.mts_idx[comp], trt = tu.tr_type[comp];
  const int scanX = tu.max_scan_x[comp], scanY = tu.max_scan_y[comp], tuQp = tu.qp[comp];
  const int16_t* __restrict__ lev = pic.coef + tu.coef_off[comp];
  const vvr_cu& cu = pic.cu[tu.cu];
  int bw = tuW >> csh, bh = tuH >> csh, bx = tuX >> csh, by = tuY >> csh;
  if( it.pad & TB_P_CUGEOM ) { bw = cu.w >> 1; bh = cu.h >> 1; bx = cu.x >> 1; by = cu.y >> 1; }      // (chroma of an ISP CU)
  const int lw = ilog2( bw ), lh = ilog2( bh ), n = bw * bh;
  const bool isTS = mts == VVR_MTS_SKIP;
  const bool bdpcmOn = ( it.pad & TB_P_BDPCM ) != 0;
  const bool lfnstBit = ( it.pad & TB_P_LFNST ) != 0;                // cu.lfnst_idx > 0 && ( cu.tree != VVR_TREE_JOINT || comp == 0 )
  const bool lfnstOn = ( pic.hdr.tool_flags & VVR_TOOL_LFNST ) && lfnstBit && !isTS;
  // the corner that carries coefficients when the passes start: the coded one, the whole block under BDPCM, at least the LFNST output region
  const int cw = scanX + 1, codedRows = scanY + 1;
  int maxX = scanX, maxY = scanY;
  if( bdpcmOn ) { maxX = bw - 1; maxY = bh - 1; }
  if( lfnstOn ) { maxX = max( maxX, min( bw - 1, 7 ) ); maxY = max( maxY, min( bh - 1, 7 ) ); }
  const int trHor = trt & 3, trVer = trt >> 2;
  const int shift1 = 7, shift2 = 20 - bd;
  const bool dcOnly = !isTS && maxX == 0 && maxY == 0 && trHor == 0 && trVer == 0;
  const bool oneD = !isTS && ( bw == 1 || bh == 1 );        // ISP partitions of 4xN / Nx4 CUs: one pass, shift_2nd + 1 (TrQuant.cpp:466-482)
  const bool twoD = !isTS && !dcOnly && !oneD;
  const bool fourRows = bh >= 4 && !oneD;                   // four neighbouring rows of a column per work item in the output stage
  // basis rows the passes touch (zero-out of the high frequencies: TrQuant_EMT.cpp:389)
  int redW = 0, cutH = 0, cntV = 0, cntH = 0;
  const int16_t* __restrict__ Mv = nullptr; const int16_t* __restrict__ Mh = nullptr;
  if( oneD )
  {
    const int n1 = bw == 1 ? bh : bw, tr = bw == 1 ? trVer : trHor, maxPos = bw == 1 ? maxY : maxX;
    const int skip = max( ( tr != 0 && n1 == 32 ) ? 16 : n1 > 32 ? n1 - 32 : 0, n1 - maxPos - 1 );
    redW = n1 - skip;                                        // rows of the basis that take part
    Mh = tr_matrix( tr, n1 ); cntH = redW * n1;
  }
  else if( twoD )
  {
    const int skipW = max( ( trHor != 0 && bw == 32 ) ? 16 : bw > 32 ? bw - 32 : 0, bw - maxX - 1 );
    const int skipH = max( ( trVer != 0 && bh == 32 ) ? 16 : bh > 32 ? bh - 32 : 0, bh - maxY - 1 );
    cutH = bh - skipH; redW = bw - skipW;
    Mv = tr_matrix( trVer, bh ); Mh = tr_matrix( trHor, bw ); cntV = cutH * bh; cntH = redW * bw;
  }
  // ---- the loads: levels (one per lane and step over the whole block: zero outside the coded corner), basis rows as dwords, prediction samples
  constexpr int ITER_C = MAXN * MAXN / NT, ITER_M = ( MAXN > 32 ? 32 : MAXN ) * MAXN / 2 / NT, ITER_E = MAXN * MAXN / 4 / NT;
  int lv[ITER_C];
  if( !bdpcmOn )
  {
#pragma unroll
    for( int k = 0; k < ITER_C; k++ )
    {
      const int i = tid + k * NT, y = i >> lw, x = i & ( bw - 1 );
      lv[k] = 0;
      if( i < n && x < cw && y < codedRows ) lv[k] = lev[y * cw + x];
    }
  }
  uint32_t mvR[ITER_M], mhR[ITER_M];
#pragma unroll
  for( int k = 0; k < ITER_M; k++ )
  {
    const int j = tid + k * NT;
    mvR[k] = 0; mhR[k] = 0;
    if( 2 * j < cntV ) mvR[k] = reinterpret_cast<const uint32_t*>( Mv )[j];
    if( 2 * j < cntH ) mhR[k] = reinterpret_cast<const uint32_t*>( Mh )[j];
  }
  const int ict = it.ict ? (int) it.ict - 4 : 0;
  const bool havePred = it.mode == TB_ADD && !ict && fourRows;
  int predv[ITER_E][4];
  {
    const pel_t* __restrict__ P = reco.p[comp]; const int stride = reco.stride[comp];
#pragma unroll
    for( int e = 0; e < ITER_E; e++ )
    {
      const int i = tid + e * NT, x = i & ( bw - 1 ), y0 = ( i >> lw ) << 2;
#pragma unroll
      for( int r = 0; r < 4; r++ ) { predv[e][r] = 0; if( havePred && i < ( bh >> 2 ) * bw ) predv[e][r] = P[(size_t) ( by + y0 + r ) * stride + bx + x]; }
    }
  }
  // ---- basis rows into LDS
#pragma unroll
  for( int k = 0; k < ITER_M; k++ )
  {
    const int j = tid + k * NT;
    if( 2 * j < cntV ) reinterpret_cast<uint32_t*>( mvS )[j] = mvR[k];
    if( 2 * j < cntH ) reinterpret_cast<uint32_t*>( mhS )[j] = mhR[k];
  }
  // ---- dequantisation
  {
    const uint32_t sliceFlags = flags_at( pic, tuX, tuY );      // dependent quantisation and the scaling lists are switches of the block's slice (Quant.cpp:306,336)
    const bool depQuant = ( sliceFlags & VVR_TOOL_DEP_QUANT ) && !isTS;
    int qp = tuQp;
    if( isTS ) qp = max( qp, (int) pic.hdr.min_qp_ts );
    const int per = depQuant ? ( qp + 1 ) / 6 : qp / 6;
    const int rem = depQuant ? ( qp + 1 - 6 * per ) : qp - 6 * per;
    const bool needSqrt = !isTS && ( ( lw + lh ) & 1 );
    const int trShift = 15 - bd - ( ( lw + lh ) >> 1 ) - ( needSqrt ? 1 : 0 );
    // explicit scaling list (getUseScalingList, Quant.h:103): not for transform skip, optionally not for LFNST blocks
    const bool useSL = pic.scaling && ( sliceFlags & VVR_TOOL_SCALING_LIST ) && !isTS && !( lfnstBit && ( pic.hdr.tool_flags & VVR_TOOL_SCALING_LIST_NO_LFNST ) );
    const int rightShift = 6 + ( depQuant ? 1 : 0 ) - ( ( isTS ? 0 : trShift ) + per ) + ( useSL ? 4 : 0 );
    const int scaleQP = d_inv_quant_scales[needSqrt ? 1 : 0][rem];
    int targetBits = 32 + rightShift - 7; if( targetBits > 16 ) targetBits = 16;
    const int inMax = ( 1 << ( targetBits - 1 ) ) - 1, inMin = -inMax - 1;
    if( bdpcmOn )
    {
      // invResDPCM (Quant.cpp:239): running sums along rows (mode 1) / columns (mode 2); one thread per line
      const int bdpcm = comp ? cu.bdpcm[1] : cu.bdpcm[0];
      const int lines = bdpcm == 1 ? bh : bw, len = bdpcm == 1 ? bw : bh;
      for( int l = tid; l < lines; l += NT )
      {
        int acc = 0;
        for( int k = 0; k < len; k++ )
        {
          const int idx = bdpcm == 1 ? l * bw + k : k * bw + l;
          const int v = lev[idx];
          acc = k == 0 ? v : clip3( -32768, 32767, acc + v );
          dq[idx] = acc;
        }
      }
      __syncthreads();
      for( int i = tid; i < n; i += NT )
      {
        const int level = dq[i];
        if( level )
        {
          const long long c = clip3( inMin, inMax, level );
          const long long v = rightShift > 0 ? ( c * scaleQP + ( 1ll << ( rightShift - 1 ) ) ) >> rightShift : ( c * scaleQP ) * ( 1ll << -rightShift );
          dq[i] = clip3( -32768, 32767, (int) v );
        }
      }
    }
    else
    {
      const int listType = useSL ? ( cu.pred_mode == VVR_PRED_INTRA ? 0 : 3 ) + comp : 0;
#pragma unroll
      for( int k = 0; k < ITER_C; k++ )
      {
        const int i = tid + k * NT;
        if( i < n )
        {
          const int level = lv[k];
          int out = 0;
          if( level )
          {
            const long long c = clip3( inMin, inMax, level );
            const int scale = useSL ? scaling_entry( pic.scaling, listType, lw, lh, i & ( bw - 1 ), i >> lw ) * scaleQP : scaleQP;
            const long long v = rightShift > 0 ? ( c * scale + ( 1ll << ( rightShift - 1 ) ) ) >> rightShift : ( c * scale ) * ( 1ll << -rightShift );
            out = clip3( -32768, 32767, (int) v );
          }
          dq[i] = out;
        }
      }
    }
  }
  __syncthreads();
  // ---- LFNST
  if( lfnstOn )
  {
    const bool whge3 = bw >= 8 && bh >= 8;
    int mode;
    if( ( cu.flags & VVR_CU_MIP ) && comp == 0 ) mode = 0;
    else if( comp && cu.intra_dir[1] >= 67 ) mode = cu.lfnst_intra_mode;
    else mode = cu.intra_dir[comp ? 1 : 0];
    mode = wide_angle_mode( ( cu.isp_mode && !comp ) ? cu.w : bw, ( cu.isp_mode && !comp ) ? cu.h : bh, mode );
    const int lm = mode < 0 ? mode + 14 + 67 : mode >= 67 ? mode + 14 : mode;
    const bool transpose = ( lm >= 67 && lm >= 67 + 14 ) || ( lm < 67 && lm > 34 );
    const int sb = whge3 ? 8 : 4;
    const int zeroOut = ( ( bw == 4 && bh == 4 ) || ( bw == 8 && bh == 8 ) ) ? 8 : 16;
    if( tid < 16 ) { const uint8_t* xy = whge3 ? d_lfnst_scan8x8_xy[tid] : d_lfnst_scan4x4_xy[tid]; lf_in[tid] = dq[xy[1] * bw + xy[0]]; }
    __syncthreads();
    const int set = d_lfnst_lut[lm], idx = cu.lfnst_idx - 1, trSize = sb == 8 ? 48 : 16;
    if( tid < trSize )
    {
      int r = 0;
      for( int i = 0; i < zeroOut; i++ ) r += lf_in[i] * ( sb == 8 ? d_lfnst8x8[set][idx][tid][i] : d_lfnst4x4[set][idx][tid][i] );
      lf_out[tid] = clip3( -32768, 32767, ( r + 64 ) >> 7 );
    }
    __syncthreads();
    if( tid < sb * sb )
    {
      const int y = tid / sb, x = tid % sb;
      if( sb == 4 ) dq[y * bw + x] = transpose ? lf_out[x * 4 + y] : lf_out[y * 4 + x];
      else if( transpose )
      {
        if( x < 4 ) dq[y * bw + x] = lf_out[x * 8 + y];
        else if( y < 4 ) dq[y * bw + x] = lf_out[32 + ( x - 4 ) * 4 + y];
      }
      else
      {
        if( y < 4 ) dq[y * bw + x] = lf_out[y * 8 + x];
        else if( x < 4 ) dq[y * bw + x] = lf_out[32 + ( y - 4 ) * 4 + x];
      }
    }
    __syncthreads();
  }
  // ---- inverse transform; every thread produces the final residual of its samples and emits it right away
  int dcVal = 0;
  if( dcOnly )
  {
    if( oneD ) dcVal = (int16_t) ( ( dq[0] * 64 + ( 1 << shift2 ) ) >> ( shift2 + 1 ) );
    else
    {
      dcVal = ( dq[0] * 64 + ( 1 << ( shift1 - 1 ) ) ) >> shift1;
      dcVal = (int16_t) ( ( dcVal * 64 + ( 1 << ( shift2 - 1 ) ) ) >> shift2 );
    }
  }
  else if( twoD )
  {
    // pass 1 (vertical): tmp[x*bh + y] = clip16( ( sum_k dq[k*bw + x] * Mv[k*bh + y] + 64 ) >> 7 ), x < redW
    // Four neighbouring columns per work item: one 16-byte read of dq and one basis value per step instead of a read of each per multiply-add
    // (the kernel lives on the LDS pipe: 512 + 1024 scalar reads per thread for a 64x64 block with a 32x32 corner before this)
    if( bw >= 4 )
    {
      const int grpX = ( redW + 3 ) >> 2;
      for( int i = tid; i < grpX * bh; i += NT )
      {
        const int xg = i >> lh, y = i & ( bh - 1 ), x0 = xg << 2;
        int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for( int k = 0; k < cutH; k++ )
        {
          const int m = mvS[k * bh + y];
          const int4 d = *reinterpret_cast<const int4*>( &dq[k * bw + x0] );
          s0 += d.x * m; s1 += d.y * m; s2 += d.z * m; s3 += d.w * m;
        }
        const int sv[4] = { s0, s1, s2, s3 };
#pragma unroll
        for( int r = 0; r < 4; r++ ) if( x0 + r < redW ) tmp[( x0 + r ) * bh + y] = clip3( -32768, 32767, ( sv[r] + ( 1 << ( shift1 - 1 ) ) ) >> shift1 );
      }
    }
    else
    for( int i = tid; i < redW * bh; i += NT )
    {
      const int x = i >> lh, y = i & ( bh - 1 );
      int sum = 0;
      for( int k = 0; k < cutH; k++ ) sum += dq[k * bw + x] * mvS[k * bh + y];
      tmp[x * bh + y] = clip3( -32768, 32767, ( sum + ( 1 << ( shift1 - 1 ) ) ) >> shift1 );
    }
    __syncthreads();
  }
  // ---- pass 2 (horizontal) + output
  // the residual r of sample (x, y) goes where it belongs: onto the prediction (inter blocks) or into the residual plane; joint Cb-Cr derives the second one
  auto emit = [&]( int x, int y, int r )
  {
    int rOther = 0, cOther = 0;
    const int cSelf = comp;
    if( ict )
    {
      // invTransformCbCr<mode> (TrQuant.cpp:108): derive the second chroma residual
      if(      ict ==  1 ) { rOther =  r >> 1; cOther = 2; }
      else if( ict == -1 ) { rOther = -r >> 1; cOther = 2; }
      else if( ict ==  2 ) { rOther =  r;      cOther = 2; }
      else if( ict == -2 ) { rOther = -r;      cOther = 2; }
      else if( ict ==  3 ) { rOther =  r >> 1; cOther = 1; }
      else                 { rOther = -r >> 1; cOther = 1; }
      rOther = (int16_t) rOther;
    }
    if( it.mode == TB_ADD )
    {
      pel_t* d = &reco.p[cSelf][(size_t) ( by + y ) * reco.stride[cSelf] + bx + x];
      *d = (pel_t) clip_pel( *d + r, bd );
      if( ict ) { pel_t* e = &reco.p[cOther][(size_t) ( by + y ) * reco.stride[cOther] + bx + x]; *e = (pel_t) clip_pel( *e + rOther, bd ); }
    }
    else
    {
      resi.p[cSelf][(size_t) ( by + y ) * resi.stride[cSelf] + bx + x] = (pel_t) r;
      if( ict ) resi.p[cOther][(size_t) ( by + y ) * resi.stride[cOther] + bx + x] = (pel_t) rOther;
    }
  };
  if( fourRows )
  {
    // out[y*bw + x] = clip16( ( sum_{k<redW} tmp[k*bh + y] * Mh[k*bw + x] + rnd ) >> shift2 ): four neighbouring rows per work item (one 16-byte read of
    // tmp and one basis value per step); neighbouring lanes hold neighbouring columns, so the stores of a row stay contiguous.  Transform-skipped and
    // DC-only blocks take the same form (their residual is there already).
    const int nItems = ( bh >> 2 ) * bw;
#pragma unroll
    for( int e = 0; e < ITER_E; e++ )
    {
      const int i = tid + e * NT;
      if( i < nItems )
      {
        const int yg = i >> lw, x = i & ( bw - 1 ), y0 = yg << 2;
        int rv[4];
        if( twoD )
        {
          int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
          for( int k = 0; k < redW; k++ )
          {
            const int m = mhS[k * bw + x];
            const int4 t = *reinterpret_cast<const int4*>( &tmp[k * bh + y0] );
            s0 += t.x * m; s1 += t.y * m; s2 += t.z * m; s3 += t.w * m;
          }
          const int sv[4] = { s0, s1, s2, s3 };
#pragma unroll
          for( int r = 0; r < 4; r++ ) rv[r] = clip3( -32768, 32767, ( sv[r] + ( 1 << ( shift2 - 1 ) ) ) >> shift2 );
        }
        else
        {
#pragma unroll
          for( int r = 0; r < 4; r++ ) rv[r] = isTS ? (int) (int16_t) dq[( y0 + r ) * bw + x] : dcVal;
        }
        if( havePred )
        {
          pel_t* d = &reco.p[comp][(size_t) ( by + y0 ) * reco.stride[comp] + bx + x];
#pragma unroll
          for( int r = 0; r < 4; r++ ) d[(size_t) r * reco.stride[comp]] = (pel_t) clip_pel( predv[e][r] + rv[r], bd );
        }
        else
        {
#pragma unroll
          for( int r = 0; r < 4; r++ ) emit( x, y0 + r, rv[r] );
        }
      }
    }
    return;
  }
  for( int i = tid; i < n; i += NT )
  {
    const int y = i >> lw, x = i & ( bw - 1 );
    int r;
    if( isTS ) r = (int16_t) dq[i];
    else if( dcOnly ) r = dcVal;
    else if( oneD )
    {
      const int n1 = bw == 1 ? bh : bw;
      int sum = 0;
      for( int k = 0; k < redW; k++ ) sum += dq[k] * mhS[k * n1 + i];        // (i runs along the only dimension)
      r = clip3( -32768, 32767, ( sum + ( 1 << shift2 ) ) >> ( shift2 + 1 ) );
    }
    else
    {
      // out[y*bw + x] = clip16( ( sum_{k<redW} tmp[k*bh + y] * Mh[k*bw + x] + rnd ) >> shift2 )
      int sum = 0;
      for( int k = 0; k < redW; k++ ) sum += tmp[k * bh + y] * mhS[k * bw + x];
      r = clip3( -32768, 32767, ( sum + ( 1 << ( shift2 - 1 ) ) ) >> shift2 );
    }
    emit( x, y, r );
  }
}

void launch_itrans( hipStream_t s, const PicDev& pic, DevPlanes reco, DevPlanes resi, const TbItem* items, int numItems, int sizeClass )
{
  if( !numItems ) return;
  if( sizeClass <= 16 )      hipLaunchKernelGGL( ( k_itrans<16, 64> ),  dim3( numItems ), dim3( 64 ),  0, s, pic, reco, resi, items, numItems );
  else if( sizeClass <= 32 ) hipLaunchKernelGGL( ( k_itrans<32, 128> ), dim3( numItems ), dim3( 128 ), 0, s, pic, reco, resi, items, numItems );
  else                       hipLaunchKernelGGL( ( k_itrans<64, 256> ), dim3( numItems ), dim3( 256 ), 0, s, pic, reco, resi, items, numItems );
}

// =====================================================================================================================
// deblocking: the edge decisions and filters shared by k_deblock_tile (a tile per workgroup, out of place) and k_deblock4 (in place).
//   LoopFilter::xDeblockCtuArea (LoopFilter.cpp:419), xEdgeFilterLuma (:1464), xEdgeFilterChroma (:1620) and the
//   filters :106-335.  All edges of one direction are independent for a valid edge-parameter table (maximum filter
//   lengths never overlap, :910-922), so the whole picture is one launch per direction.
// =====================================================================================================================
#define BS_GET( v, c ) ( ( ( v ) >> ( ( c ) << 1 ) ) & 3 )

__device__ __forceinline__ int calc_dp( const pel_t* s, int o ) { return iabs( s[-o * 3] - 2 * s[-o * 2] + s[-o] ); }
__device__ __forceinline__ int calc_dp_ctb( const pel_t* s, int o ) { return iabs( s[-o * 2] - 2 * s[-o * 2] + s[-o] ); }
__device__ __forceinline__ int calc_dq( const pel_t* s, int o ) { return iabs( s[0] - 2 * s[o] + s[o * 2] ); }

__device__ bool use_strong( const pel_t* s, int o, int d, int beta, int tc, bool pLarge, bool qLarge, int lenP, int lenQ, bool chromaCtb )
{
  const int m3 = s[-o], m4 = s[0];
  if( !( d < ( beta >> 2 ) && iabs( m3 - m4 ) < ( ( tc * 5 + 1 ) >> 1 ) ) ) return false;
  const int m0 = s[-4 * o], m7 = s[3 * o], m2 = s[-2 * o];
  int sp3 = iabs( m0 - m3 );
  if( chromaCtb ) sp3 = iabs( m2 - m3 );
  int sq3 = iabs( m7 - m4 );
  const int d_strong = sp3 + sq3;
  if( pLarge || qLarge )
  {
    if( pLarge )
    {
      const int mP4 = s[-o * lenP - o];
      if( lenP == 7 ) sp3 = sp3 + iabs( s[-o * 5] - s[-o * 6] - s[-o * 7] + mP4 );
      sp3 = ( sp3 + iabs( m0 - mP4 ) + 1 ) >> 1;
    }
    if( qLarge )
    {
      const int m11 = s[o * lenQ];
      if( lenQ == 7 ) sq3 = sq3 + iabs( s[o * 4] - s[o * 5] - s[o * 6] + m11 );
      sq3 = ( sq3 + iabs( m11 - m7 ) + 1 ) >> 1;
    }
    return ( ( sp3 + sq3 ) < ( beta * 3 >> 5 ) ) && ( d < ( beta >> 4 ) ) && ( iabs( m3 - m4 ) < ( ( tc * 5 + 1 ) >> 1 ) );
  }
  return d_strong < ( beta >> 3 );
}

// Long luma filter (VVC 8.8.3.6.7, the reference's xFilteringPandQCore, LoopFilter.cpp:129-196), table-driven: the filtered samples are a linear
// interpolation between a "middle" value at the edge and a reference value at the far end of each side, limited to +-( tc * tcFactor ) / 2.
//   middle  = weighted mean (weights sum to 16) of the samples next to the edge; the weights of a side depend on its own length and the other's
//   coefficient of sample k of a side of n samples = round-down( ( 64 * ( 2 * ( n - k ) - 1 ) + n ) / ( 2 * n ) ): 59 50 41 32 23 14 5 / 58 45 32 19 6 / 53 32 11
__constant__ uint8_t c_dbLongMidW[3][3][7] = {       // [own length 3 / 5 / 7][other side's length][sample]
  { { 0, 0, 0, 0, 0, 0, 0 }, { 2, 2, 2, 2, 0, 0, 0 }, { 3, 3, 2, 0, 0, 0, 0 } },
  { { 2, 2, 2, 2, 0, 0, 0 }, { 2, 2, 2, 1, 1, 0, 0 }, { 2, 2, 1, 1, 1, 1, 0 } },
  { { 2, 1, 1, 1, 1, 1, 1 }, { 2, 2, 1, 1, 1, 1, 0 }, { 2, 1, 1, 1, 1, 1, 1 } } };
__constant__ uint8_t c_dbLongTc[2][7] = { { 6, 4, 2, 0, 0, 0, 0 }, { 6, 5, 4, 3, 2, 1, 1 } };      // tc factor per sample: sides of 3, sides of 5 or 7
// interpolation weights of the long filters, ( 64 * ( 2 * ( n - k ) - 1 ) + n ) / ( 2 * n ) for n = 3, 5, 7 (dbCoeffs3 / 5 / 7 of the standard): a table, not a
// division per sample
__constant__ uint8_t c_dbLongCf[3][8] = { { 53, 32, 11, 0, 0, 0, 0, 0 }, { 58, 45, 32, 19, 6, 0, 0, 0 }, { 59, 50, 41, 32, 23, 14, 5, 0 } };
__device__ void filter_long( pel_t* src, int step, int o, int nP, int nQ, int tc )
{
  const int iP = ( nP - 3 ) >> 1, iQ = ( nQ - 3 ) >> 1;
  for( int line = 0; line < 4; line++ )
  {
    pel_t* q0 = src + step * line; pel_t* p0 = q0 - o;       // sample k of the P side: p0[-k * o], of the Q side: q0[k * o]
    int mid = 8;
    for( int k = 0; k < 7; k++ ) mid += c_dbLongMidW[iP][iQ][k] * p0[-k * o] + c_dbLongMidW[iQ][iP][k] * q0[k * o];
    mid >>= 4;
    const int farP = ( p0[-( nP - 1 ) * o] + p0[-nP * o] + 1 ) >> 1, farQ = ( q0[( nQ - 1 ) * o] + q0[nQ * o] + 1 ) >> 1;
    for( int side = 0; side < 2; side++ )
    {
      pel_t* s = side ? q0 : p0; const int d = side ? o : -o, n = side ? nQ : nP, far = side ? farQ : farP;
      for( int k = 0; k < n; k++ )
      {
        const int cf = c_dbLongCf[( n - 3 ) >> 1][k], lim = ( tc * c_dbLongTc[n > 3][k] ) >> 1, v = s[d * k];
        s[d * k] = (pel_t) clip3( v - lim, v + lim, ( mid * cf + far * ( 64 - cf ) + 32 ) >> 6 );
      }
    }
  }
}

__device__ __forceinline__ void filter_luma_pel( pel_t* s, int o, int tc, bool sw, int thrCut, bool fP, bool fQ, int bd )
{
  const int m1 = s[-3 * o], m2 = s[-2 * o], m3 = s[-o], m4 = s[0], m5 = s[o], m6 = s[2 * o];
  if( sw )
  {
    const int m0 = s[-4 * o], m7 = s[3 * o];
    s[-3 * o] = (pel_t) clip3( m1 - 1 * tc, m1 + 1 * tc, ( 2 * m0 + 3 * m1 + m2 + m3 + m4 + 4 ) >> 3 );
    s[-2 * o] = (pel_t) clip3( m2 - 2 * tc, m2 + 2 * tc, ( m1 + m2 + m3 + m4 + 2 ) >> 2 );
    s[-1 * o] = (pel_t) clip3( m3 - 3 * tc, m3 + 3 * tc, ( m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4 ) >> 3 );
    s[0]      = (pel_t) clip3( m4 - 3 * tc, m4 + 3 * tc, ( m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4 ) >> 3 );
    s[o]      = (pel_t) clip3( m5 - 2 * tc, m5 + 2 * tc, ( m3 + m4 + m5 + m6 + 2 ) >> 2 );
    s[2 * o]  = (pel_t) clip3( m6 - 1 * tc, m6 + 1 * tc, ( m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4 ) >> 3 );
  }
  else
  {
    int delta = ( 9 * ( m4 - m3 ) - 3 * ( m5 - m2 ) + 8 ) >> 4;
    if( iabs( delta ) < thrCut )
    {
      delta = clip3( -tc, tc, delta );
      const int tc2 = tc >> 1;
      s[-o] = (pel_t) clip_pel( m3 + delta, bd );
      if( fP ) s[-2 * o] = (pel_t) clip_pel( m2 + clip3( -tc2, tc2, ( ( ( m1 + m3 + 1 ) >> 1 ) - m2 + delta ) >> 1 ), bd );
      s[0] = (pel_t) clip_pel( m4 - delta, bd );
      if( fQ ) s[o] = (pel_t) clip_pel( m5 + clip3( -tc2, tc2, ( ( ( m6 + m4 + 1 ) >> 1 ) - m5 - delta ) >> 1 ), bd );
    }
  }
}

__device__ __forceinline__ void filter_chroma_pel( pel_t* s, int o, int tc, bool sw, int bd, bool ctb )
{
  const int m2 = s[-2 * o], m3 = s[-o], m4 = s[0], m5 = s[o];
  if( sw )
  {
    const int m6 = s[2 * o], m7 = s[3 * o];
    if( ctb )
    {
      s[-o]    = (pel_t) clip3( m3 - tc, m3 + tc, ( 3 * m2 + 2 * m3 + m4 + m5 + m6 + 4 ) >> 3 );
      s[0]     = (pel_t) clip3( m4 - tc, m4 + tc, ( 2 * m2 + m3 + 2 * m4 + m5 + m6 + m7 + 4 ) >> 3 );
      s[o]     = (pel_t) clip3( m5 - tc, m5 + tc, ( m2 + m3 + m4 + 2 * m5 + m6 + 2 * m7 + 4 ) >> 3 );
      s[2 * o] = (pel_t) clip3( m6 - tc, m6 + tc, ( m3 + m4 + m5 + 2 * m6 + 3 * m7 + 4 ) >> 3 );
    }
    else
    {
      const int m0 = s[-4 * o], m1 = s[-3 * o];
      s[-3 * o] = (pel_t) clip3( m1 - tc, m1 + tc, ( 3 * m0 + 2 * m1 + m2 + m3 + m4 + 4 ) >> 3 );
      s[-2 * o] = (pel_t) clip3( m2 - tc, m2 + tc, ( 2 * m0 + m1 + 2 * m2 + m3 + m4 + m5 + 4 ) >> 3 );
      s[-o]     = (pel_t) clip3( m3 - tc, m3 + tc, ( m0 + m1 + m2 + 2 * m3 + m4 + m5 + m6 + 4 ) >> 3 );
      s[0]      = (pel_t) clip3( m4 - tc, m4 + tc, ( m1 + m2 + m3 + 2 * m4 + m5 + m6 + m7 + 4 ) >> 3 );
      s[o]      = (pel_t) clip3( m5 - tc, m5 + tc, ( m2 + m3 + m4 + 2 * m5 + m6 + 2 * m7 + 4 ) >> 3 );
      s[2 * o]  = (pel_t) clip3( m6 - tc, m6 + tc, ( m3 + m4 + m5 + 2 * m6 + 3 * m7 + 4 ) >> 3 );
    }
  }
  else
  {
    const int delta = clip3( -tc, tc, ( ( ( m4 - m3 ) * 4 ) + m2 - m5 + 4 ) >> 3 );
    s[-o] = (pel_t) clip_pel( m3 + delta, bd );
    s[0]  = (pel_t) clip_pel( m4 - delta, bd );
  }
}

__device__ __forceinline__ int tc_value( int idx, int bd ) { const int t = d_db_tc_table[idx]; return bd < 10 ? ( t + ( 1 << ( 9 - bd ) ) ) >> ( 10 - bd ) : t << ( bd - 10 ); }

// An edge whose P side may be filtered over 7 samples reaches the samples of a coding-sub-block edge 8 samples before it
// (SbTMVP CU on the P side: the reference keeps 7 there, LoopFilter.cpp:920, and filters the edges of a CTU in raster
// order, :447-462, so the sub-block edge comes first).  Such a pair is handled by ONE thread, in that order; every other
// pair of edges of one direction touches disjoint samples.
__device__ __forceinline__ bool db_luma_p7( const vvr_lfp& l ) { return BS_GET( l.bs, 0 ) && ( ( l.side_max_filt_length >> 4 ) & 7 ) == 7; }

// ---------------------------------------------------------------------------------------------------------------------
// k_deblock4 - the edge filters with FOUR LANES PER EDGE SEGMENT (round 4).  k_deblock (rounds 1 - 3, removed in round 6) gave a thread a whole 4-sample segment: ~70 dependent
// memory operations per wavefront on 2-byte accesses, one round of wavefronts on the device - latency, not bandwidth, not arithmetic (DESIGN.md
// section 6, profiles/round3_deblock_counters.json).  Here a lane owns ONE LINE of the segment: it fetches the line's 16 samples across the edge at
// once (four 8-byte loads along a row for vertical edges; 16 two-byte loads down a column for horizontal edges, 128 bytes per row over the
// wavefront), parks them in LDS next to the other three lines of its quad, takes the segment's decisions from lines 0 and 3 there (every lane for
// itself: the arithmetic is cheap, nothing is exchanged), filters its own line in LDS and writes back exactly the samples the chosen filter touched.
// Memory round trips per lane: the edge parameters, the line, the stores.  Chroma: the two lines of a unit's Cb and Cr edge are the four lanes.
// ---------------------------------------------------------------------------------------------------------------------
#define DB_LS 20          // samples per staged line in LDS (16 used: p7 .. p0 q0 .. q7; 40 bytes keep 8-byte alignment)

// The DECISIONS of a luma edge segment, taken from its lines 0 and 3 (VVC 8.8.3.6.2, xEdgeFilterLuma LoopFilter.cpp:1389-1570): `src` = q0 of line 0, `o` = distance
// of neighbouring samples across the edge, `step` = distance of the segment's lines.  -> code: bits 0-1 the filter (0 none, 1 long, 2 strong, 3 weak), bits 2-4
// the P-side length of the long filter (weak: bit 2 = the second P sample too), bits 5-7 the same for Q, bits 8.. tc
__device__ __forceinline__ uint32_t deblock_luma_decide( const PicDev& pic, const pel_t* src, int o, int step, int x, int y, int dir, const vvr_lfp& l )
{
  const vvr_pic_header& H = pic.hdr;
  const int bd = H.bit_depth;
  const int bsY = BS_GET( l.bs, 0 );
  int qp = l.qp[0];
  if( H.ladf_num_intervals )
  {
    // luma-adaptive deblocking: QP offset chosen by the mean of four samples at the corners of the segment (deriveLADFShift, LoopFilter.cpp:1363-1386)
    const int level = ( src[0] + src[3 * step] + src[-o] + src[3 * step - o] ) >> 2;
    int shift = H.ladf_qp_offset[0];
    for( int k = 1; k < H.ladf_num_intervals; k++ ) { if( level > H.ladf_lower_bound[k] ) shift = H.ladf_qp_offset[k]; else break; }
    qp += shift;
  }
  const int lenP = ( l.side_max_filt_length >> 4 ) & 7, lenQ = l.side_max_filt_length & 7;
  bool pLarge = lenP > 3, qLarge = lenQ > 3;
  if( dir == 1 && ( y & ( ( 1 << H.log2_ctu ) - 1 ) ) == 0 ) pLarge = false;
  // the offsets of the slice the deblocked CTU belongs to - the CTU that holds the segment, its Q side (LoopFilter.cpp:421,1473)
  const int offs = deblock_offsets_at( pic, x, y, 0 );
  const int idxTC = clip3( 0, 65, qp + 2 * ( bsY - 1 ) + 2 * ( ( offs >> 8 ) - 64 ) );
  const int idxB  = clip3( 0, 63, qp + 2 * ( ( offs & 255 ) - 64 ) );
  const int tc = tc_value( idxTC, bd ), beta = d_db_beta_table[idxB] << ( bd - 8 );
  const int sideThr = ( beta + ( beta >> 1 ) ) >> 3;
  const pel_t* s0 = src; const pel_t* s3 = src + 3 * step;
  const int dp0 = calc_dp( s0, o ), dq0 = calc_dq( s0, o ), dp3 = calc_dp( s3, o ), dq3 = calc_dq( s3, o );
  const int d0 = dp0 + dq0, d3 = dp3 + dq3;
  if( pLarge || qLarge )
  {
    const int o3 = 3 * o;
    const int dp0L = pLarge ? ( dp0 + calc_dp( s0 - o3, o ) + 1 ) >> 1 : dp0;
    const int dq0L = qLarge ? ( dq0 + calc_dq( s0 + o3, o ) + 1 ) >> 1 : dq0;
    const int dp3L = pLarge ? ( dp3 + calc_dp( s3 - o3, o ) + 1 ) >> 1 : dp3;
    const int dq3L = qLarge ? ( dq3 + calc_dq( s3 + o3, o ) + 1 ) >> 1 : dq3;
    const int d0L = dp0L + dq0L, d3L = dp3L + dq3L, dL = d0L + d3L;
    if( dL < beta )
    {
      const bool swL = use_strong( s0, o, 2 * d0L, beta, tc, pLarge, qLarge, lenP, lenQ, false ) && use_strong( s3, o, 2 * d3L, beta, tc, pLarge, qLarge, lenP, lenQ, false );
      if( swL ) return 1u | ( ( pLarge ? lenP : 3 ) << 2 ) | ( ( qLarge ? lenQ : 3 ) << 5 ) | ( (uint32_t) tc << 8 );
    }
  }
  const int dp = dp0 + dp3, dq = dq0 + dq3, d = d0 + d3;
  if( d < beta )
  {
    bool fP = false, fQ = false, sw = false;
    if( lenP > 1 && lenQ > 1 ) { fP = dp < sideThr; fQ = dq < sideThr; }
    if( lenP > 2 && lenQ > 2 ) sw = use_strong( s0, o, 2 * d0, beta, tc, false, false, 7, 7, false ) && use_strong( s3, o, 2 * d3, beta, tc, false, false, 7, 7, false );
    if( sw ) return 2u | ( (uint32_t) tc << 8 );
    return 3u | ( fP ? 4u : 0u ) | ( fQ ? 32u : 0u ) | ( (uint32_t) tc << 8 );
  }
  return 0;
}

// long luma filter of one line (filter_long, one of its four lines)
__device__ __forceinline__ void filter_long_line( pel_t* q0, int o, int nP, int nQ, int tc )
{
  const int iP = ( nP - 3 ) >> 1, iQ = ( nQ - 3 ) >> 1;
  pel_t* p0 = q0 - o;
  int mid = 8;
  for( int k = 0; k < 7; k++ ) mid += c_dbLongMidW[iP][iQ][k] * p0[-k * o] + c_dbLongMidW[iQ][iP][k] * q0[k * o];
  mid >>= 4;
  const int farP = ( p0[-( nP - 1 ) * o] + p0[-nP * o] + 1 ) >> 1, farQ = ( q0[( nQ - 1 ) * o] + q0[nQ * o] + 1 ) >> 1;
  for( int side = 0; side < 2; side++ )
  {
    pel_t* s = side ? q0 : p0; const int d = side ? o : -o, n = side ? nQ : nP, far = side ? farQ : farP;
    for( int k = 0; k < n; k++ )
    {
      const int cf = c_dbLongCf[( n - 3 ) >> 1][k], lim = ( tc * c_dbLongTc[n > 3][k] ) >> 1, v = s[d * k];
      s[d * k] = (pel_t) clip3( v - lim, v + lim, ( mid * cf + far * ( 64 - cf ) + 32 ) >> 6 );
    }
  }
}

// the filter a segment's decisions chose, on ONE of its lines (`q0` = the line's first sample behind the edge)
__device__ __forceinline__ void deblock_luma_apply( pel_t* q0, int o, uint32_t code, int bd )
{
  const int kind = code & 3, tc = code >> 8;
  if( kind == 1 ) filter_long_line( q0, o, ( code >> 2 ) & 7, ( code >> 5 ) & 7, tc );
  else if( kind == 2 ) filter_luma_pel( q0, o, tc, true, 0, false, false, bd );
  else if( kind == 3 ) filter_luma_pel( q0, o, tc, false, tc * 10, ( code & 4 ) != 0, ( code & 32 ) != 0, bd );
}

// One line (`li`) of the luma edge segment at sample position (x, y) whose four lines are staged at `seg` (line stride `step`, q0 of a line at index 8): decisions
// by every lane for itself, then the lane's own line.  Returns the samples this line's filter modified: P side | Q side << 4.
__device__ __forceinline__ int deblock_luma_line( const PicDev& pic, pel_t* seg, int li, int x, int y, int dir, const vvr_lfp& l, const int step = DB_LS )
{
  const uint32_t code = deblock_luma_decide( pic, seg + 8, 1, step, x, y, dir, l );
  // (the other lanes take their decisions from lines 0 and 3 too - which their own lanes change: every lane has read them before any lane writes)
  asm volatile( "s_waitcnt lgkmcnt(0)" ::: "memory" );
  pel_t* s = seg + 8 + step * li;
  const int kind = code & 3;
  if( !kind ) return 0;
  if( kind == 1 ) { deblock_luma_apply( s, 1, code, pic.hdr.bit_depth ); return ( ( code >> 2 ) & 7 ) | ( ( ( code >> 5 ) & 7 ) << 4 ); }
  if( kind == 2 ) { deblock_luma_apply( s, 1, code, pic.hdr.bit_depth ); return 3 | ( 3 << 4 ); }
  const int tc = code >> 8, m2 = s[-2], m3 = s[-1], m4 = s[0], m5 = s[1];
  const int delta = ( 9 * ( m4 - m3 ) - 3 * ( m5 - m2 ) + 8 ) >> 4;
  if( iabs( delta ) >= tc * 10 ) return 0;
  deblock_luma_apply( s, 1, code, pic.hdr.bit_depth );
  return ( ( code & 4 ) ? 2 : 1 ) | ( ( ( code & 32 ) ? 2 : 1 ) << 4 );
}

__global__ __launch_bounds__( 256 ) void k_deblock4( PicDev pic, DevPlanes r, int dir )
{
  __shared__ pel_t lines[256 * DB_LS];
  const int tid = threadIdx.x, quad = tid >> 2, li = tid & 3;
  // a wavefront = 16 neighbouring units of a unit row: rows of 128 bytes over the wavefront for horizontal edges, neighbouring 32-byte windows of four
  // rows for vertical edges; a workgroup = 16 x 4 units
  const int x4 = blockIdx.x * 16 + ( quad & 15 ), y4 = blockIdx.y * 4 + ( quad >> 4 );
  if( x4 >= pic.w4 || y4 >= pic.h4 ) return;
  const vvr_lfp* lp = pic.lfp[dir] + (size_t) y4 * pic.w4 + x4;
  const vvr_lfp l = *lp;
  const vvr_pic_header& H = pic.hdr;
  const int bd = H.bit_depth;
  pel_t* const seg = &lines[( tid & ~3 ) * DB_LS];           // the quad's four lines
  pel_t* const mine = seg + li * DB_LS;
  const int nStep = dir == 0 ? 2 : 2 * pic.w4;                      // table distance of the unit 8 samples across the edge
  const bool hasNext = dir == 0 ? x4 + 2 < pic.w4 : y4 + 2 < pic.h4, hasPrev = dir == 0 ? x4 >= 2 : y4 >= 2;
  // (the LDS accesses of a wavefront execute in order; the quad's lanes sit in one wavefront: what the compiler must not do is move them across)
#define DB_SYNC() asm volatile( "s_waitcnt lgkmcnt(0)" ::: "memory" )
  if( BS_GET( l.bs, 0 ) && !( hasNext && db_luma_p7( lp[nStep] ) ) )
  {
    pel_t* __restrict__ P = r.p[0];
    const int stride = r.stride[0], W = r.w[0], Hh = r.h[0];
    // sample k of this lane's line (k = 0: 8 samples before the edge): where it lies in the plane
    const int x = x4 * 4, y = y4 * 4;
    auto loadLine = [&]( int ex, int ey, int k0, int k1 )            // samples k0 .. k1 - 1 of the line of the segment at (ex, ey) into `mine`
    {
      if( dir == 0 )
      {
        const pel_t* row = P + (size_t) ( ey + li ) * stride;
#pragma unroll
        for( int c = 0; c < 4; c++ )
        {
          if( 4 * c < k0 || 4 * c >= k1 ) continue;
          const int cx = ex - 8 + 4 * c;
          uint2 v = make_uint2( 0, 0 );
          if( cx >= 0 && cx < W ) v = *reinterpret_cast<const uint2*>( row + cx );
          *reinterpret_cast<uint2*>( mine + 4 * c ) = v;
        }
      }
      else
      {
        const pel_t* col = P + ex + li;
        int v[16];
#pragma unroll
        for( int k = 0; k < 16; k++ ) { const int ry = ey - 8 + k; v[k] = ( k >= k0 && k < k1 && ry >= 0 && ry < Hh ) ? (int) col[(size_t) ry * stride] : 0; }
#pragma unroll
        for( int k = 0; k < 16; k++ ) if( k >= k0 && k < k1 ) mine[k] = (pel_t) v[k];
      }
    };
    auto storeLine = [&]( int ex, int ey, int mod )                   // the samples the filter modified: mod & 15 before the edge, mod >> 4 behind it
    {
      const int nP = mod & 15, nQ = mod >> 4;
      if( dir == 0 )
      {
        pel_t* row = P + (size_t) ( ey + li ) * stride + ex;
        for( int k = 1; k <= nP; k++ ) row[-k] = mine[8 - k];
        for( int k = 0; k < nQ; k++ ) row[k] = mine[8 + k];
      }
      else
      {
        pel_t* col = P + (size_t) ey * stride + ex + li;
        for( int k = 1; k <= nP; k++ ) col[-(ptrdiff_t) k * stride] = mine[8 - k];
        for( int k = 0; k < nQ; k++ ) col[(size_t) k * stride] = mine[8 + k];
      }
    };
    bool havePside = false;
    if( db_luma_p7( l ) && hasPrev )
    {
      // the one overlapping pair: a coding-sub-block edge 8 samples before an edge whose P side is filtered over 7 samples - that edge first, by the
      // same four lanes (the reference's raster order), and its Q side is what this edge finds on its P side
      const vvr_lfp lPrev = lp[-nStep];
      if( BS_GET( lPrev.bs, 0 ) )
      {
        const int px = dir == 0 ? x - 8 : x, py = dir == 0 ? y : y - 8;
        loadLine( px, py, 0, 16 );
        DB_SYNC();
        const int mod = deblock_luma_line( pic, seg, li, px, py, dir, lPrev );
        DB_SYNC();
        storeLine( px, py, mod );
        // its Q side (samples 8 .. 15 of the staged line) becomes this edge's P side (samples 0 .. 7)
        uint2 a = *reinterpret_cast<const uint2*>( mine + 8 ), b = *reinterpret_cast<const uint2*>( mine + 12 );
        DB_SYNC();
        *reinterpret_cast<uint2*>( mine ) = a; *reinterpret_cast<uint2*>( mine + 4 ) = b;
        havePside = true;
      }
    }
    loadLine( x, y, havePside ? 8 : 0, 16 );
    DB_SYNC();
    const int mod = deblock_luma_line( pic, seg, li, x, y, dir, l );
    DB_SYNC();
    storeLine( x, y, mod );
  }
  if( !l.bs ) return;
  // ---- chroma (4:2:0): edges on the 8-chroma-sample grid, two chroma lines per 4x4 luma unit: lanes 0, 1 = the lines of Cb, lanes 2, 3 = the lines of Cr
  if( !H.chroma_format ) return;
  if( dir == 0 ? ( x4 & 3 ) : ( y4 & 3 ) ) return;
  const int bS[2] = { BS_GET( l.bs, 1 ), BS_GET( l.bs, 2 ) };
  if( !bS[0] && !bS[1] ) return;
  const int c = li >> 1, cl = li & 1;
  const bool large = ( l.flags >> 5 ) & 1;
  const int bSc = c ? bS[1] : bS[0];
  const bool active = bSc == 2 || ( large && bSc == 1 );
  const int stride = r.stride[1], cx = x4 * 2, cy = y4 * 2, CW = r.w[1], CH = r.h[1];
  pel_t* __restrict__ Pc = c ? r.p[2] : r.p[1];
  const bool ctb = dir == 1 && ( cy & ( ( ( 1 << H.log2_ctu ) - 1 ) >> 1 ) ) == 0;
  // this lane's line: samples 4 .. 11 of the staged line = 4 before and 4 behind the edge (q0 at index 8, like luma); both lines of a component are needed
  // for the decision, so both of its lanes stage theirs first
  DB_SYNC();
  if( active )
  {
    if( dir == 0 )
    {
      const pel_t* row = Pc + (size_t) ( cy + cl ) * stride;
#pragma unroll
      for( int k = 0; k < 2; k++ )
      {
        const int sx = cx - 4 + 4 * k;
        uint2 v = make_uint2( 0, 0 );
        if( sx >= 0 && sx < CW ) v = *reinterpret_cast<const uint2*>( row + sx );
        *reinterpret_cast<uint2*>( mine + 4 + 4 * k ) = v;
      }
    }
    else
    {
      const pel_t* col = Pc + cx + cl;
      int v[8];
#pragma unroll
      for( int k = 0; k < 8; k++ ) { const int ry = cy - 4 + k; v[k] = ( ry >= 0 && ry < CH ) ? (int) col[(size_t) ry * stride] : 0; }
#pragma unroll
      for( int k = 0; k < 8; k++ ) mine[4 + k] = (pel_t) v[k];
    }
  }
  DB_SYNC();
  if( !active ) return;
  {
    pel_t* src = seg + ( c * 2 ) * DB_LS + 8;                     // q0 of the component's first line; its second line is DB_LS further
    const int o = 1, step = DB_LS;
    const int qp = c ? l.qp[2] : l.qp[1];
    const int offs = deblock_offsets_at( pic, x4 * 4, y4 * 4, 1 + c );      // offsets of the deblocked CTU's slice (LoopFilter.cpp:1637-1638)
    const int idxTC = clip3( 0, 65, qp + 2 * ( bSc - 1 ) + 2 * ( ( offs >> 8 ) - 64 ) );
    const int tc = tc_value( idxTC, bd );
    bool sw = false;
    if( large )
    {
      const int idxB = clip3( 0, 63, qp + 2 * ( ( offs & 255 ) - 64 ) );
      const int beta = d_db_beta_table[idxB] * ( 1 << ( bd - 8 ) );
      const int dp0 = ctb ? calc_dp_ctb( src, o ) : calc_dp( src, o ), dq0 = calc_dq( src, o );
      const int dp3 = ctb ? calc_dp_ctb( src + step, o ) : calc_dp( src + step, o ), dq3 = calc_dq( src + step, o );
      const int d0 = dp0 + dq0, d3 = dp3 + dq3, d = d0 + d3;
      if( d < beta ) sw = use_strong( src, o, 2 * d0, beta, tc, false, false, 7, 7, ctb ) && use_strong( src + step, o, 2 * d3, beta, tc, false, false, 7, 7, ctb );
    }
    DB_SYNC();                                                      // (both lanes of the component have read both lines)
    pel_t* s = src + step * cl;
    filter_chroma_pel( s, o, tc, sw, bd, ctb );
    DB_SYNC();
    const int nP = sw ? ( ctb ? 1 : 3 ) : 1, nQ = sw ? 3 : 1;
    if( dir == 0 )
    {
      pel_t* row = Pc + (size_t) ( cy + cl ) * stride + cx;
      for( int k = 1; k <= nP; k++ ) row[-k] = s[-k];
      for( int k = 0; k < nQ; k++ ) row[k] = s[k];
    }
    else
    {
      pel_t* col = Pc + (size_t) cy * stride + cx + cl;
      for( int k = 1; k <= nP; k++ ) col[-(ptrdiff_t) k * stride] = s[-k];
      for( int k = 0; k < nQ; k++ ) col[(size_t) k * stride] = s[k];
    }
  }
#undef DB_SYNC
}


// One line (`cl` = 0 / 1) of the chroma edge segment of a 4x4 luma unit, the two lines staged at `src` (q0 of the first line; line stride `step`): decision from both
// lines (`o` = distance of neighbouring samples across the edge), filter of the own line.  Returns the samples modified: P side | Q side << 4.  The caller keeps the pair's reads in front of its writes (DB_SYNC).
__device__ __forceinline__ int deblock_chroma_line( const PicDev& pic, pel_t* src, int o, int step, int cl, int c, int x4, int y4, int dir, const vvr_lfp& l, int bSc, bool large )
{
  const vvr_pic_header& H = pic.hdr;
  const int bd = H.bit_depth, cy = y4 * 2;
  const bool ctb = dir == 1 && ( cy & ( ( ( 1 << H.log2_ctu ) - 1 ) >> 1 ) ) == 0;
  const int qp = c ? l.qp[2] : l.qp[1];
  const int offs = deblock_offsets_at( pic, x4 * 4, y4 * 4, 1 + c );      // offsets of the deblocked CTU's slice (LoopFilter.cpp:1637-1638)
  const int idxTC = clip3( 0, 65, qp + 2 * ( bSc - 1 ) + 2 * ( ( offs >> 8 ) - 64 ) );
  const int tc = tc_value( idxTC, bd );
  bool sw = false;
  if( large )
  {
    const int idxB = clip3( 0, 63, qp + 2 * ( ( offs & 255 ) - 64 ) );
    const int beta = d_db_beta_table[idxB] * ( 1 << ( bd - 8 ) );
    const int dp0 = ctb ? calc_dp_ctb( src, o ) : calc_dp( src, o ), dq0 = calc_dq( src, o );
    const int dp3 = ctb ? calc_dp_ctb( src + step, o ) : calc_dp( src + step, o ), dq3 = calc_dq( src + step, o );
    const int d0 = dp0 + dq0, d3 = dp3 + dq3, d = d0 + d3;
    if( d < beta ) sw = use_strong( src, o, 2 * d0, beta, tc, false, false, 7, 7, ctb ) && use_strong( src + step, o, 2 * d3, beta, tc, false, false, 7, 7, ctb );
  }
  asm volatile( "s_waitcnt lgkmcnt(0)" ::: "memory" );                // (both lanes of the component have read both lines)
  filter_chroma_pel( src + step * cl, o, tc, sw, bd, ctb );
  return ( sw ? ( ctb ? 1 : 3 ) : 1 ) | ( ( sw ? 3 : 1 ) << 4 );
}

// ---------------------------------------------------------------------------------------------------------------------
// k_deblock_tile<LMCS, DIR> - the edges of one direction, a TILE per workgroup, OUT OF PLACE (round 4); vertical edges: 128 x 16 luma samples (and the 64 x 8 of
// Cb and Cr) with the inverse luma mapping in the load (k_lmcs is not launched), horizontal edges: 64 x 64.
// With a lane per line (k_deblock4) the vertical pass still took 46 us: its lines run along rows, so every store instruction of a wavefront put 2 bytes into each
// of 64 different cache lines, 14 such instructions per edge; and in both passes the four lanes of a segment each took the segment's decisions - three quarters
// of the instructions the device issued.  Here the workgroup OWNS a tile.  It loads it with the halo its edges read - 16 samples before the tile: the P side of
// the first edge and the sub-block edge 8 samples before that, the one overlapping pair; 8 behind it: the Q side of the edge on the tile's far border, whose P
// side changes the tile's last samples - with wide loads, all issued before the first barrier; lists the segments that have an edge at all (an eighth at a mean
// block size of 32); takes the decisions ONE LANE PER SEGMENT; applies the filters four lanes per segment, a line each; and writes the whole tile with 16-byte
// stores into ANOTHER picture buffer (a neighbour's halo must see unfiltered samples).  The edge on the far border is filtered by two workgroups, each keeping
// its side.  Chroma: two lines per segment and component, a lane per line.
// ---------------------------------------------------------------------------------------------------------------------
#define DT_HL 16                                    // halo before the tile (across the edges)
#define DT_HR 8                                     // halo behind it
template<int DIR> struct DbTile;
template<> struct DbTile<0> { static constexpr int N = 128, T = 16; };      // N: owned samples across the edges, T: along them
template<> struct DbTile<1> { static constexpr int N = 64, T = 64; };
template<bool LMCS, int DIR>
__global__ __launch_bounds__( 256 ) __attribute__( ( amdgpu_waves_per_eu( 8, 8 ) ) ) void k_deblock_tile( PicDev pic, DevPlanes s, DevPlanes d, int dbg )
{
  constexpr int NT = 256;
  constexpr int N = DbTile<DIR>::N, T = DbTile<DIR>::T, NN = DT_HL + N + DT_HR, NNC = 4 + N / 2 + 4;
  // LDS row strides (rows stay 8-byte aligned).  Vertical edges: lines run along rows, the lanes of a segment are a row stride apart - 4 samples of padding spread them
  // over the banks; horizontal edges: the lanes of a segment are neighbours in a row whatever the stride, and without the padding the tile's 20.2 KB let EIGHT
  // workgroups share a compute unit - a 4K picture's 2040 tiles are resident at once (with 21.5 KB: seven, and a second round for 248 of them)
  constexpr int TS = DIR == 0 ? NN + 4 : T, CS = DIR == 0 ? NNC + 4 : T / 2;
  constexpr int YO = DIR == 0 ? 1 : TS, YSTEP = DIR == 0 ? TS : 1;                            // luma: across the edge / from line to line
  constexpr int CO = DIR == 0 ? 1 : CS, CSTEP = DIR == 0 ? CS : 1;
  constexpr int EC = N / 4 + 1, EA = T / 4, NE = EC * EA;                                      // edge positions across (the far border's included) x units along
  __shared__ pel_t ty[( DIR == 0 ? T : NN ) * TS];
  __shared__ pel_t tc[2][( DIR == 0 ? T / 2 : NNC ) * CS];
  __shared__ int16_t lut[LMCS ? 1024 : 2];
  __shared__ vvr_lfp edge[NE];
  __shared__ uint16_t list[NE];
  __shared__ uint32_t codes[NE];
  __shared__ int cnt;
  const int tid = threadIdx.x;
  const int X0 = blockIdx.x * ( DIR == 0 ? N : T ), Y0 = blockIdx.y * ( DIR == 0 ? T : N ), X4 = X0 >> 2, Y4 = Y0 >> 2;
  const vvr_pic_header& H = pic.hdr;
  const int W = s.w[0], Hh = s.h[0], bd = H.bit_depth;
  const bool chroma = H.chroma_format != 0;
  // ---- every global load of the workgroup first, into registers: the table, the edge parameters, the samples (one memory latency, not three)
  constexpr int NLF = ( NE + NT - 1 ) / NT;
  constexpr int NLU = 512 / NT;
  if( tid == 0 ) cnt = 0;
  uint32_t lutv[NLU];
  if( LMCS )
  {
    const uint32_t* lp32 = reinterpret_cast<const uint32_t*>( pic.lmcs->inv_lut ); const int n2 = ( 1 << bd ) >> 1;
#pragma unroll
    for( int k = 0; k < NLU; k++ ) { lutv[k] = 0; if( tid + NT * k < n2 ) lutv[k] = lp32[tid + NT * k]; }
  }
  const int lfAcross = DIR == 0 ? 2 : 2 * pic.w4;                    // table distance of the unit 8 samples across the edge
  vvr_lfp lf[NLF]; uint32_t lfMine = 0;
#pragma unroll
  for( int k = 0; k < NLF; k++ )
  {
    vvr_lfp& l = lf[k];
    l.qp[0] = l.qp[1] = l.qp[2] = 0; l.bs = 0; l.side_max_filt_length = 0; l.flags = 0; l.pad[0] = l.pad[1] = 0;
    const int t = tid + NT * k, e = t % EC, a = t / EC, x4 = X4 + ( DIR == 0 ? e : a ), y4 = Y4 + ( DIR == 0 ? a : e );
    if( t < NE && x4 < pic.w4 && y4 < pic.h4 )
    {
      const vvr_lfp* lp = pic.lfp[DIR] + (size_t) y4 * pic.w4 + x4;
      l = *lp;
      bool nextP7 = false;
      if( DIR == 0 ? x4 + 2 < pic.w4 : y4 + 2 < pic.h4 ) nextP7 = db_luma_p7( lp[lfAcross] );
      // a luma edge that is this segment's to filter (not the first of an overlapping pair: its partner takes both)
      if( BS_GET( l.bs, 0 ) && !nextP7 ) lfMine |= 1u << k;
    }
  }
  // luma chunks: vertical edges 4 samples (8 bytes) of a row, horizontal edges 8 samples (16 bytes) of a row; chroma the same
  constexpr int LCPR = DIR == 0 ? NN / 4 : T / 8, LROWS = DIR == 0 ? T : NN, NLC = ( LROWS * LCPR + NT - 1 ) / NT;
  constexpr int CCPR = DIR == 0 ? NNC / 4 : T / 16, CROWS = DIR == 0 ? T / 2 : NNC, NCC = ( 2 * CROWS * CCPR + NT - 1 ) / NT;
  uint4 yv[NLC], cv[NCC];
  uint32_t ymap = 0;
  {
    const pel_t* __restrict__ P = s.p[0];
    const int stride = s.stride[0];
#pragma unroll
    for( int k = 0; k < NLC; k++ )
    {
      const int i = tid + NT * k, r = i / LCPR, c = i - r * LCPR;
      const int x = DIR == 0 ? X0 - DT_HL + 4 * c : X0 + 8 * c, y = DIR == 0 ? Y0 + r : Y0 - DT_HL + r;
      yv[k] = make_uint4( 0, 0, 0, 0 );
      if( i < LROWS * LCPR && y >= 0 && y < Hh && x >= 0 && x < W && !( dbg & 2 ) )
      {
        if( DIR == 0 ) { const uint2 v = *reinterpret_cast<const uint2*>( P + (size_t) y * stride + x ); yv[k].x = v.x; yv[k].y = v.y; }
        else yv[k] = *reinterpret_cast<const uint4*>( P + (size_t) y * stride + x );
        if( LMCS && ( !pic.slices || ( flags_at( pic, x, y ) & VVR_TOOL_LMCS ) ) ) ymap |= 1u << k;        // the CTU's slice uses LMCS (Reshape.cpp:385)
      }
    }
    if( chroma )
    {
      const int CW = s.w[1], CH = s.h[1], cstride = s.stride[1];
#pragma unroll
      for( int k = 0; k < NCC; k++ )
      {
        const int i = tid + NT * k, pl = i / ( CROWS * CCPR ), j = i - pl * CROWS * CCPR, r = j / CCPR, c = j - r * CCPR;
        const int x = DIR == 0 ? X0 / 2 - 4 + 4 * c : X0 / 2 + 8 * c, y = DIR == 0 ? Y0 / 2 + r : Y0 / 2 - 4 + r;
        cv[k] = make_uint4( 0, 0, 0, 0 );
        if( i < 2 * CROWS * CCPR && y >= 0 && y < CH && x >= 0 && x < CW )
        {
          const pel_t* p = s.p[1 + pl] + (size_t) y * cstride + x;
          if( DIR == 0 ) { const uint2 v = *reinterpret_cast<const uint2*>( p ); cv[k].x = v.x; cv[k].y = v.y; }
          else cv[k] = *reinterpret_cast<const uint4*>( p );
        }
      }
    }
  }
  if( LMCS )
  {
    uint32_t* l32 = reinterpret_cast<uint32_t*>( lut ); const int n2 = ( 1 << bd ) >> 1;
#pragma unroll
    for( int k = 0; k < NLU; k++ ) if( tid + NT * k < n2 ) l32[tid + NT * k] = lutv[k];
  }
  __syncthreads();
  // ---- the tile's edge parameters and the list of the segments to filter
#pragma unroll
  for( int k = 0; k < NLF; k++ )
  {
    const int t = tid + NT * k;
    if( t < NE ) { edge[t] = lf[k]; if( ( lfMine >> k ) & 1 ) list[atomicAdd( &cnt, 1 )] = (uint16_t) t; }
  }
  // ---- the samples into the tile, luma through the inverse table of the slice
  {
    const int n1 = ( 1 << bd ) - 1;
#pragma unroll
    for( int k = 0; k < NLC; k++ )
    {
      const int i = tid + NT * k, r = i / LCPR, c = i - r * LCPR;
      uint4 v = yv[k];
      if( LMCS && ( ( ymap >> k ) & 1 ) )
      {
        v.x = (uint32_t) (uint16_t) lut[v.x & 0xffff & n1] | ( (uint32_t) (uint16_t) lut[( v.x >> 16 ) & n1] << 16 );
        v.y = (uint32_t) (uint16_t) lut[v.y & 0xffff & n1] | ( (uint32_t) (uint16_t) lut[( v.y >> 16 ) & n1] << 16 );
        if( DIR == 1 )
        {
          v.z = (uint32_t) (uint16_t) lut[v.z & 0xffff & n1] | ( (uint32_t) (uint16_t) lut[( v.z >> 16 ) & n1] << 16 );
          v.w = (uint32_t) (uint16_t) lut[v.w & 0xffff & n1] | ( (uint32_t) (uint16_t) lut[( v.w >> 16 ) & n1] << 16 );
        }
      }
      if( i < LROWS * LCPR )
      {
        pel_t* q = &ty[r * TS + ( DIR == 0 ? 4 : 8 ) * c];
        *reinterpret_cast<uint2*>( q ) = make_uint2( v.x, v.y );
        if( DIR == 1 ) *reinterpret_cast<uint2*>( q + 4 ) = make_uint2( v.z, v.w );
      }
    }
    if( chroma )
    {
#pragma unroll
      for( int k = 0; k < NCC; k++ )
      {
        const int i = tid + NT * k, pl = i / ( CROWS * CCPR ), j = i - pl * CROWS * CCPR, r = j / CCPR, c = j - r * CCPR;
        if( i < 2 * CROWS * CCPR )
        {
          pel_t* q = &tc[pl][r * CS + ( DIR == 0 ? 4 : 8 ) * c];
          *reinterpret_cast<uint2*>( q ) = make_uint2( cv[k].x, cv[k].y );
          if( DIR == 1 ) *reinterpret_cast<uint2*>( q + 4 ) = make_uint2( cv[k].z, cv[k].w );
        }
      }
    }
  }
  __syncthreads();
  // ---- luma decisions: a lane per listed segment.  q0 of line 0 of segment (e, a): DT_HL + 4 e across, 4 a along
  // (a tile has some 25 of them: one wavefront's work, and a different wavefront - SIMD - of the workgroup from tile to tile, or the first SIMD of every compute
  // unit would take the decisions of all its workgroups)
  const int nEdges = ( dbg & 1 ) ? 0 : cnt;
  const int rot = ( dbg & 16 ) ? 0 : ( ( blockIdx.x + blockIdx.y ) & 3 ) * 64;
  for( int i = ( tid + rot ) & ( NT - 1 ); i < nEdges; i += NT )
  {
    const int t = list[i], e = t % EC, a = t / EC, x4 = X4 + ( DIR == 0 ? e : a ), y4 = Y4 + ( DIR == 0 ? a : e ), x = x4 * 4, y = y4 * 4;
    const vvr_lfp le = edge[t];
    pel_t* q0 = &ty[( DT_HL + 4 * e ) * YO + ( 4 * a ) * YSTEP];
    if( db_luma_p7( le ) && ( DIR == 0 ? x4 >= 2 : y4 >= 2 ) )
    {
      // the one overlapping pair: a coding-sub-block edge 8 samples before an edge whose P side is filtered over 7 samples - that edge first, decisions and all four
      // lines by this lane (the reference's raster order); its Q side is what this edge finds on its P side
      const vvr_lfp lPrev = e >= 2 ? edge[t - 2] : pic.lfp[DIR][(size_t) y4 * pic.w4 + x4 - lfAcross];
      if( BS_GET( lPrev.bs, 0 ) )
      {
        const uint32_t cp = deblock_luma_decide( pic, q0 - 8 * YO, YO, YSTEP, DIR == 0 ? x - 8 : x, DIR == 0 ? y : y - 8, DIR, lPrev );
        for( int li = 0; li < 4; li++ ) deblock_luma_apply( q0 - 8 * YO + li * YSTEP, YO, cp, bd );
      }
    }
    codes[i] = ( dbg & 64 ) ? ( 2u | ( 5u << 8 ) ) : deblock_luma_decide( pic, q0, YO, YSTEP, x, y, DIR, le );
  }
  __syncthreads();
  // ---- luma filters: four lanes per listed segment, a line each
  for( int i = ( tid + rot + 64 ) & ( NT - 1 ); i < 4 * nEdges; i += NT )
  {
    const uint32_t code = codes[i >> 2];
    if( !( code & 3 ) || ( dbg & 32 ) ) continue;
    const int t = list[i >> 2], e = t % EC, a = t / EC;
    deblock_luma_apply( &ty[( DT_HL + 4 * e ) * YO + ( 4 * a + ( i & 3 ) ) * YSTEP], YO, code, bd );
  }
  // ---- chroma: edges on the 8-chroma-sample grid (every 4th edge position), two lines per unit and component, a lane per line
  if( chroma && !( dbg & 8 ) )
  {
    constexpr int CE = N / 16 + 1, PER = CE * ( T / 2 );             // lines of a component
    for( int t = tid; t < 2 * PER; t += NT )
    {
      const int c = t / PER, rem = t - c * PER, ce = rem / ( T / 2 ), line = rem % ( T / 2 ), a = line >> 1, cl = line & 1;
      const vvr_lfp lc = edge[a * EC + 4 * ce];
      const int bSc = BS_GET( lc.bs, 1 + c );
      const bool large = ( lc.flags >> 5 ) & 1;
      if( bSc == 2 || ( large && bSc == 1 ) )
        deblock_chroma_line( pic, &tc[c][( 4 + 8 * ce ) * CO + ( 2 * a ) * CSTEP], CO, CSTEP, cl, c, X4 + ( DIR == 0 ? 4 * ce : a ), Y4 + ( DIR == 0 ? a : 4 * ce ), DIR, lc, bSc, large );
    }
  }
  __syncthreads();
  // ---- the tile, 16 bytes per lane and store
  {
    constexpr int OW = DIR == 0 ? N : T, OH = DIR == 0 ? T : N, OX = DIR == 0 ? DT_HL : 0, OY = DIR == 0 ? 0 : DT_HL;
    for( int i = tid; i < OH * ( OW / 8 ); i += NT )
    {
      const int r = i / ( OW / 8 ), c = i % ( OW / 8 ), x = X0 + 8 * c, y = Y0 + r;
      if( y < Hh && x < W && !( dbg & 4 ) )
      {
        const pel_t* q = &ty[( OY + r ) * TS + OX + 8 * c];
        const uint2 a = *reinterpret_cast<const uint2*>( q ), b = *reinterpret_cast<const uint2*>( q + 4 );
        *reinterpret_cast<uint4*>( d.p[0] + (size_t) y * d.stride[0] + x ) = make_uint4( a.x, a.y, b.x, b.y );
      }
    }
    if( chroma )
    {
      constexpr int CWo = OW / 2, CHo = OH / 2, CX = DIR == 0 ? 4 : 0, CY = DIR == 0 ? 0 : 4;
      for( int i = tid; i < 2 * CHo * ( CWo / 8 ); i += NT )
      {
        const int pl = i / ( CHo * ( CWo / 8 ) ), j = i % ( CHo * ( CWo / 8 ) ), r = j / ( CWo / 8 ), c = j % ( CWo / 8 ), x = X0 / 2 + 8 * c, y = Y0 / 2 + r;
        if( y < s.h[1] && x < s.w[1] )
        {
          const pel_t* q = &tc[pl][( CY + r ) * CS + CX + 8 * c];
          const uint2 a = *reinterpret_cast<const uint2*>( q ), b = *reinterpret_cast<const uint2*>( q + 4 );
          *reinterpret_cast<uint4*>( d.p[1 + pl] + (size_t) y * d.stride[1] + x ) = make_uint4( a.x, a.y, b.x, b.y );
        }
      }
    }
  }
}

// one pass out of place (src -> dst); vertical edges: the inverse luma mapping folded in when `lmcs`
void launch_deblock_tile( hipStream_t st, const PicDev& pic, DevPlanes src, DevPlanes dst, int dir, bool lmcs )
{
  int dbg = 0;
#ifdef VVR_DEV_ENV
  static const int dbgEnv = getenv( "VVR_DBV_DBG" ) ? atoi( getenv( "VVR_DBV_DBG" ) ) : 0;      // developer build: 1 no edges, 2 no luma loads, 4 no luma stores, 8 no chroma edges (timing only)
  dbg = dbgEnv;
#endif
  if( dir == 0 )
  {
    const dim3 grid( ( src.w[0] + DbTile<0>::N - 1 ) / DbTile<0>::N, ( src.h[0] + DbTile<0>::T - 1 ) / DbTile<0>::T );
    if( lmcs ) hipLaunchKernelGGL( ( k_deblock_tile<true, 0> ), grid, dim3( 256 ), 0, st, pic, src, dst, dbg );
    else       hipLaunchKernelGGL( ( k_deblock_tile<false, 0> ), grid, dim3( 256 ), 0, st, pic, src, dst, dbg );
  }
  else
  {
    const dim3 grid( ( src.w[0] + DbTile<1>::T - 1 ) / DbTile<1>::T, ( src.h[0] + DbTile<1>::N - 1 ) / DbTile<1>::N );
    hipLaunchKernelGGL( ( k_deblock_tile<false, 1> ), grid, dim3( 256 ), 0, st, pic, src, dst, dbg );
  }
}

// =====================================================================================================================
// k_lf_* — the reference's LF_INIT task (LoopFilter::calcFilterStrengthsCTU, LoopFilter.cpp:495-1360; DecLibRecon.cpp:807-829) for pictures that leave the
// deblocking edge parameters to the back-end (VVR_TOOL_LFP_ON_DEVICE).  The derivation itself is vvr_lf_init.h (one function per 4x4 cell and direction, also
// compiled for the host by the tests); here: the per-cell records of both trees and the motion field as the filter sees it (one thread per transform unit; one per
// cell of the host's list for CUs whose motion varies inside the CU) in one launch, the two tables (one thread per cell) in another.  HBM-bound and small: about
// 27 MB written and 10 MB read back per 4K B picture against 8.3 MB of tables that no longer cross PCIe.
// =====================================================================================================================
// eight lanes per transform unit: its record into the cells it covers, the motion of those cells where the CU's record holds it (three stores per cell: 16-byte
// record, 16 bytes of vectors, the reference indices).  A transform unit of up to eight cells is written by its own lanes, a cell each; the larger ones of a
// wavefront - eight at most - are written by all 64 lanes together, one after the other (a 64x64 unit has 256 cells), from values broadcast out of the owner's
// registers: nothing is loaded in those loops.  (One lane per unit left 64 such rounds per wavefront on a fifth of the chip's SIMDs: 34 us instead of 9.)
// Behind the transform units: one thread per cell of the host's list of sub-block motion.
__device__ __forceinline__ void prep_lf_maps( int bid, const PicDev& pic, int numCu, int numTu, LfCell* __restrict__ cell, LfCell* __restrict__ cellC, LfMv* __restrict__ mvs, uint32_t* __restrict__ refs,
                                              const LfSbCell* __restrict__ sb, int numSb, int dbg )
{
  const int gt = bid * 256 + threadIdx.x, lane = threadIdx.x & 63, t = gt >> 3, sub = gt & 7;
  const int w4 = pic.w4, h4 = pic.h4;
  if( t >= numTu )
  {
    const int i = gt - 8 * numTu;
    if( i < numSb ) { const LfSbCell e = sb[i]; if( e.cell < (uint32_t) ( w4 * h4 ) ) { const LfMv v = lfi_pack_mv( e.m ); *reinterpret_cast<uint4*>( &mvs[e.cell] ) = make_uint4( v.v[0][0], v.v[0][1], v.v[1][0], v.v[1][1] ); refs[e.cell] = lfi_pack_refs( e.m ); } }
  }
  LfCell rec; rec.a = rec.b = rec.c = rec.d = 0;
  int x0 = 0, y0 = 0, nx = 0, ny = 0, cuIdx = 0, cuX4 = 0, cuY4 = 0;
  int kind = 0;               // bit 0..1: motion of the cells (0 none, 1 the same for all of them: mo, 2 from the host's list, 3 spanned per cell from the control points); bit 2: chroma tree
  uint32_t mo[5] = { 0, 0, 0, 0, 0 };
  if( t < numTu )
  {
    const vvr_tu T = pic.tu[t];
    cuIdx = lfi_idx( (int) T.cu, numCu );
    const vvr_cu& C = pic.cu[cuIdx];
    if( lfi_tu_owns_cells( T ) && !( C.tree == VVR_TREE_CHROMA && !cellC ) )
    {
      rec = lfi_pack_cell( T, t, C, pic.tu[lfi_idx( (int) ( C.first_tu + C.num_tu ) - 1, numTu )] );
      x0 = T.x >> 2; y0 = T.y >> 2; nx = min( ( T.x + T.w + 3 ) >> 2, w4 ) - x0; ny = min( ( T.y + T.h + 3 ) >> 2, h4 ) - y0;
      if( nx <= 0 || ny <= 0 ) nx = ny = 0;
      cuX4 = C.x >> 2; cuY4 = C.y >> 2;
      vvr_motion m;
      kind = lfi_cell_motion( pic.hdr, C, x0, y0, m );
      if( kind == 1 && C.pred_mode == VVR_PRED_INTER && ( C.flags & VVR_CU_AFFINE ) ) kind = 3;
      if( kind == 1 ) { mo[0] = (uint32_t) m.mv[0][0]; mo[1] = (uint32_t) m.mv[0][1]; mo[2] = (uint32_t) m.mv[1][0]; mo[3] = (uint32_t) m.mv[1][1]; mo[4] = lfi_pack_refs( m ); }
      if( C.tree == VVR_TREE_CHROMA ) kind |= 4;
      if( ( dbg & 1 ) && ( kind & 3 ) == 3 ) kind = ( kind & 4 ) | 1;      // (timing: affine cells like plain ones)
      if( dbg & 2 ) kind &= 4;                                            // (timing: no motion stores)
    }
  }
  auto put = [&]( const LfCell& r, const vvr_cu& cuRec, int cx4, int cy4, int k, const uint32_t* mv, int x, int y )
  {
    const LfCell q = lfi_cell_at( r, cx4, cy4, x, y );
    const size_t at = (size_t) y * w4 + x;
    *reinterpret_cast<uint4*>( &( ( k & 4 ) ? cellC : cell )[at] ) = make_uint4( q.a, q.b, q.c, q.d );
    if( ( k & 3 ) == 1 ) { *reinterpret_cast<uint4*>( &mvs[at] ) = make_uint4( mv[0], mv[1], mv[2], mv[3] ); refs[at] = mv[4]; }
    else if( ( k & 3 ) == 3 )
    {   // (affine CU, VVR_TOOL_AFFINE_MV_ON_DEVICE: per cell from the control points)
      vvr_motion m; lfi_cell_motion( pic.hdr, cuRec, x, y, m );
      *reinterpret_cast<uint4*>( &mvs[at] ) = make_uint4( (uint32_t) m.mv[0][0], (uint32_t) m.mv[0][1], (uint32_t) m.mv[1][0], (uint32_t) m.mv[1][1] ); refs[at] = lfi_pack_refs( m );
    }
  };
  const int n = ( dbg & 4 ) ? 0 : nx * ny;                                // (timing: no cell stores at all: the loads and the packing)
  if( sub < n && n <= 8 ) put( rec, pic.cu[cuIdx], cuX4, cuY4, kind, mo, x0 + sub % nx, y0 + sub / nx );
  unsigned long long big = __ballot( n > 8 && sub == 0 );
  while( big )
  {
    const int L = __builtin_amdgcn_readfirstlane( __ffsll( (long long) big ) - 1 ); big &= big - 1;
#define BC( V ) ( (int) __builtin_amdgcn_readlane( (int) ( V ), L ) )
    LfCell r; r.a = (uint32_t) BC( rec.a ); r.b = (uint32_t) BC( rec.b ); r.c = (uint32_t) BC( rec.c ); r.d = (uint32_t) BC( rec.d );
    const int bx0 = BC( x0 ), by0 = BC( y0 ), bnx = BC( nx ), bn = BC( n ), bcu = BC( cuIdx ), bcx = BC( cuX4 ), bcy = BC( cuY4 ), bk = BC( kind );
    uint32_t bm[5]; for( int e = 0; e < 5; e++ ) bm[e] = (uint32_t) BC( mo[e] );
#undef BC
    // (an affine CU: its record into registers ONCE - read per cell through the pointer it came back after every store, a chain of loads per round of 64 cells:
    // 8 such units in a wavefront were the launch's 30 us)
    vvr_cu bC; if( ( bk & 3 ) == 3 ) bC = pic.cu[bcu];
    for( int i = lane; i < bn; i += 64 ) put( r, bC, bcx, bcy, bk, bm, bx0 + i % bnx, by0 + i / bnx );
  }
}
// one thread per cell: both directions (the cell's record is read once, as one 16-byte load)
__global__ __launch_bounds__( 256 ) void k_lf_init( PicDev pic, const LfCell* __restrict__ cell, const LfCell* __restrict__ cellC, const LfMv* __restrict__ mvs, const uint32_t* __restrict__ refs,
                                                   vvr_lfp* __restrict__ out0, vvr_lfp* __restrict__ out1 )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= pic.w4 * pic.h4 ) return;
  const int y4 = i / pic.w4, x4 = i - y4 * pic.w4;
  LfInitView V; V.hdr = &pic.hdr; V.cell = cell; V.cellC = cellC; V.mv = mvs; V.ref = refs; V.ctuSlice = pic.ctuSlice; V.ctuTile = pic.ctuTile;
  V.ctuSubpic = pic.ctuSubpic; V.subpics = pic.subpics; V.slices = pic.slices; V.w4 = pic.w4; V.h4 = pic.h4; V.ctusX = pic.ctus_x;
  static_assert( sizeof( vvr_lfp ) == 8 && sizeof( LfCell ) == 16, "one 8-byte store per table entry, one 16-byte load per cell record" );
  // the cell and the cells before it in both directions: three loads in flight together (nine cells in ten need nothing else)
  // (as one asm statement: written as three loads the compiler moved the third behind the first direction's decisions - where its value is first used)
  typedef uint32_t u32x4 __attribute__( ( ext_vector_type( 4 ) ) );
  u32x4 qv, pv0, pv1;
  {
    const LfCell* aq = &cell[i]; const LfCell* a0 = &cell[x4 > 0 ? i - 1 : i]; const LfCell* a1 = &cell[y4 > 0 ? i - pic.w4 : i];
    asm volatile( "global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %4, off\n\tglobal_load_dwordx4 %2, %5, off\n\ts_waitcnt vmcnt(0)"
                  : "=&v"( qv ), "=&v"( pv0 ), "=&v"( pv1 ) : "v"( aq ), "v"( a0 ), "v"( a1 ) : "memory" );
  }
  LfCell Q, P0, P1; Q.a = qv.x; Q.b = qv.y; Q.c = qv.z; Q.d = qv.w; P0.a = pv0.x; P0.b = pv0.y; P0.c = pv0.z; P0.d = pv0.w; P1.a = pv1.x; P1.b = pv1.y; P1.c = pv1.z; P1.d = pv1.w;
  const vvr_lfp a = lf_init_cell( V, 0, x4, y4, Q, P0 ), b = lf_init_cell( V, 1, x4, y4, Q, P1 );
  *reinterpret_cast<uint2*>( &out0[i] ) = *reinterpret_cast<const uint2*>( &a );
  *reinterpret_cast<uint2*>( &out1[i] ) = *reinterpret_cast<const uint2*>( &b );
}
void launch_lf_init( hipStream_t s, const PicDev& pic, uint32_t numCu, uint32_t numTu, LfCell* cell, LfCell* cellC, LfMv* mv, uint32_t* ref, const LfSbCell* sbCells, int numSbCells, vvr_lfp* out0, vvr_lfp* out1 )
{
  // (the cell maps were written by launch_prep at the head of the picture)
  if( !numTu || !numCu ) return;
  const int cells = pic.w4 * pic.h4;
  hipLaunchKernelGGL( k_lf_init, dim3( ( cells + 255 ) / 256 ), dim3( 256 ), 0, s, pic, (const LfCell*) cell, (const LfCell*) cellC, (const LfMv*) mv, (const uint32_t*) ref, out0, out1 );
}

void launch_deblock( hipStream_t s, const PicDev& pic, DevPlanes reco, int dir )
{
  if( pic.hdr.tool_flags & VVR_TOOL_DEBLOCK_OFF ) return;
  hipLaunchKernelGGL( k_deblock4, dim3( ( pic.w4 + 15 ) / 16, ( pic.h4 + 3 ) / 4 ), dim3( 256 ), 0, s, pic, reco, dir );
}

// =====================================================================================================================
// slice / tile boundaries in the in-loop filters.  SAO and ALF of a CTU may not look at samples of a neighbouring CTU in another slice
// (tile) when pps_loop_filter_across_slices (tiles)_enabled_flag is off: SAO leaves the samples whose neighbour would lie there alone
// (deriveLoopFilterBoundaryAvailibility, SampleAdaptiveOffset.cpp:741-805), ALF reads a replicated border instead
// (isClipOrCrossedByVirtualBoundaries + the padded copy of filterCTU, AdaptiveLoopFilter.cpp:118-291,764-840).
// =====================================================================================================================
__device__ __forceinline__ bool lf_restricted( const PicDev& pic )
{
  return ( ( pic.hdr.tool_flags & VVR_TOOL_NO_LF_ACROSS_SLICES ) && pic.ctuSlice ) || ( ( pic.hdr.tool_flags & VVR_TOOL_NO_LF_ACROSS_TILES ) && pic.ctuTile ) || pic.ctuSubpic;
}
// may a filter working on CTU a read samples of CTU b?
__device__ __forceinline__ bool lf_may_cross( const PicDev& pic, int a, int b )
{
  if( a == b ) return true;
  if( ( pic.hdr.tool_flags & VVR_TOOL_NO_LF_ACROSS_SLICES ) && pic.ctuSlice && pic.ctuSlice[a] != pic.ctuSlice[b] ) return false;
  if( ( pic.hdr.tool_flags & VVR_TOOL_NO_LF_ACROSS_TILES ) && pic.ctuTile && pic.ctuTile[a] != pic.ctuTile[b] ) return false;
  // sub-pictures: the flag of the sub-picture the filtered CTU lies in decides (SampleAdaptiveOffset.cpp:806-818, AdaptiveLoopFilter.cpp:183-186)
  if( pic.ctuSubpic && pic.ctuSubpic[a] != pic.ctuSubpic[b] && !pic.subpics[pic.ctuSubpic[a]].lf_across ) return false;
  return true;
}
// ALF: the part of the plane a CTU may read, as a clamp of the coordinates.  Bits of f: 1 left, 2 right, 4 top, 8 bottom edge of the CTU clipped;
// 16 / 32: the CTU diagonally above-left / below-right lies in another slice while the CTUs above and left (below and right) do not (raster-
// scan slices): the corner beyond is read from the CTU's first (last) column of the same row (AreaBuf::padBorderPel, Buffer.h:608).
struct AlfClip { int x0, y0, x1, y1; uint32_t f; };
__device__ __forceinline__ AlfClip alf_clip_of_ctu( const PicDev& pic, int ctuX, int ctuY, int cs )
{
  AlfClip k; k.f = 0;
  const int S = ( 1 << pic.hdr.log2_ctu ) >> cs;
  k.x0 = ctuX * S; k.y0 = ctuY * S; k.x1 = k.x0 + S - 1; k.y1 = k.y0 + S - 1;
  if( !lf_restricted( pic ) ) return k;
  const int a = ctuY * pic.ctus_x + ctuX;
  const bool hasL = ctuX > 0, hasR = ctuX + 1 < pic.ctus_x, hasT = ctuY > 0, hasB = ctuY + 1 < pic.ctus_y;
  if( hasL && !lf_may_cross( pic, a, a - 1 ) ) k.f |= 1;
  if( hasR && !lf_may_cross( pic, a, a + 1 ) ) k.f |= 2;
  if( hasT && !lf_may_cross( pic, a, a - pic.ctus_x ) ) k.f |= 4;
  if( hasB && !lf_may_cross( pic, a, a + pic.ctus_x ) ) k.f |= 8;
  if( ( pic.hdr.tool_flags & VVR_TOOL_NO_LF_ACROSS_SLICES ) && pic.ctuSlice )
  {
    if( !( k.f & 5 ) && hasL && hasT && pic.ctuSlice[a - pic.ctus_x - 1] != pic.ctuSlice[a] ) k.f |= 16;
    if( !( k.f & 10 ) && hasR && hasB && pic.ctuSlice[a + pic.ctus_x + 1] != pic.ctuSlice[a] ) k.f |= 32;
  }
  return k;
}
__device__ __forceinline__ void alf_clip_coord( const AlfClip& k, int& x, int& y )
{
  if( !k.f ) return;
  if( ( k.f & 1 ) && x < k.x0 ) x = k.x0;
  if( ( k.f & 2 ) && x > k.x1 ) x = k.x1;
  if( ( k.f & 4 ) && y < k.y0 ) y = k.y0;
  if( ( k.f & 8 ) && y > k.y1 ) y = k.y1;
  if( ( k.f & 16 ) && x < k.x0 && y < k.y0 ) x = k.x0;
  if( ( k.f & 32 ) && x > k.x1 && y > k.y1 ) x = k.x1;
}
// Virtual boundaries of the picture header.  ALF: the boundaries that touch or cross the CTU cut it into parts, each filtered with a replicated
// border of its own (filterCTU, AdaptiveLoopFilter.cpp:764-850; a boundary on the CTU's edge is a clipped edge, :146-172).  For the part that holds
// the luma position (lx, ly): the CTU's clip `k` (component cs) narrowed to that part.  The corner padding of raster-scan slices belongs to the
// part at the CTU's origin / end only, which is what remains of it when the clipped sides are applied first (alf_clip_coord).
__device__ __forceinline__ bool vb_present( const PicDev& pic ) { return ( pic.hdr.num_ver_vb | pic.hdr.num_hor_vb ) != 0; }
// (the boundary positions are read with constant indices: a run-time index into the by-value kernel argument would make the compiler keep an addressable
// copy of it per thread, in LDS)
#define VB_EACH_X( BODY ) { if( pic.hdr.num_ver_vb > 0 ) { const int v = pic.hdr.vb_pos_x[0]; BODY } if( pic.hdr.num_ver_vb > 1 ) { const int v = pic.hdr.vb_pos_x[1]; BODY } if( pic.hdr.num_ver_vb > 2 ) { const int v = pic.hdr.vb_pos_x[2]; BODY } }
#define VB_EACH_Y( BODY ) { if( pic.hdr.num_hor_vb > 0 ) { const int v = pic.hdr.vb_pos_y[0]; BODY } if( pic.hdr.num_hor_vb > 1 ) { const int v = pic.hdr.vb_pos_y[1]; BODY } if( pic.hdr.num_hor_vb > 2 ) { const int v = pic.hdr.vb_pos_y[2]; BODY } }
__device__ __forceinline__ AlfClip alf_clip_vb( const PicDev& pic, AlfClip k, int lx, int ly, int cs )
{
  const int S = 1 << pic.hdr.log2_ctu, cx0 = lx & ~( S - 1 ), cy0 = ly & ~( S - 1 );
  VB_EACH_X( if( v >= cx0 && v <= cx0 + S ) { if( v <= lx ) { k.f = ( k.f | 1 ) & ~16u; k.x0 = max( k.x0, v >> cs ); } else { k.f = ( k.f | 2 ) & ~32u; k.x1 = min( k.x1, ( v >> cs ) - 1 ); } } )
  VB_EACH_Y( if( v >= cy0 && v <= cy0 + S ) { if( v <= ly ) { k.f = ( k.f | 4 ) & ~16u; k.y0 = max( k.y0, v >> cs ); } else { k.f = ( k.f | 8 ) & ~32u; k.y1 = min( k.y1, ( v >> cs ) - 1 ); } } )
  return k;
}
// SAO: a sample in the column (row) on either side of a virtual boundary is left alone by the edge classes that look across it
// (SampleAdaptiveOffset::isProcessDisabled, SampleAdaptiveOffset.cpp:823; the horizontal class only knows vertical boundaries, the vertical class only
// horizontal ones, :112,156).  x, y in component samples.
__device__ __forceinline__ bool sao_at_vb( const PicDev& pic, int x, int y, int cs, bool ver, bool hor )
{
  bool at = false;
  if( ver ) VB_EACH_X( { const int p = v >> cs; at = at || x == p || x == p - 1; } )
  if( hor ) VB_EACH_Y( { const int p = v >> cs; at = at || y == p || y == p - 1; } )
  return at;
}

// =====================================================================================================================
// k_sao — SampleAdaptiveOffset::offsetBlock_core (SampleAdaptiveOffset.cpp:64) per sample; reads the deblocked picture,
// writes a second picture, so neighbour reads always see pre-SAO samples (the reference needs a line copy for that, :400).
// =====================================================================================================================
__global__ __launch_bounds__( 256 ) void k_sao( PicDev pic, DevPlanes src, DevPlanes dst )
{
  // eight consecutive samples of a row per thread (one 16-byte load / store; they share a CTU, hence the SAO parameters); a wavefront
  // covers 512 samples of a row, a workgroup four rows.  Everything a thread may need is loaded before anything is looked at - its own
  // eight samples, the eight above and below, the CTU's parameters - so that the kernel pays ONE memory round trip (it used to pay three:
  // parameters, then the rows the class asks for, then their outer samples); the samples left and right of the three runs come from the
  // neighbouring lanes, only the first and the last lane of a wavefront load them.
  const int c = blockIdx.z;
  const int cs = c ? 1 : 0;
  // (selects instead of src.p[c]: a run-time index into the by-value argument makes the compiler keep a per-thread copy of the array in LDS)
  const int cw = c ? src.w[1] : src.w[0], chh = c ? src.h[1] : src.h[0];
  const int nbx = ( cw + 511 ) >> 9, nby = ( chh + 3 ) >> 2;                 // workgroups this plane needs (the grid is sized for luma)
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  if( blk >= nbx * nby ) return;
  const int blkLin = xcd_contiguous( blk, nbx * nby );
  const int lane = threadIdx.x & 63;
  const int x0 = ( ( blkLin % nbx ) * 64 + lane ) * 8;
  const int y = ( blkLin / nbx ) * 4 + ( threadIdx.x >> 6 );
  if( y >= chh ) return;                                                    // (whole wavefronts leave: a wavefront is one row)
  const bool inside = x0 < cw;
  const int xl = inside ? x0 : 0;                                           // lanes right of the plane take part in the shuffles with harmless data
  const int bd = pic.hdr.bit_depth, ctuC = ( 1 << pic.hdr.log2_ctu ) >> cs;
  const pel_t* __restrict__ S = c == 0 ? src.p[0] : c == 1 ? src.p[1] : src.p[2];
  const int st = c ? src.stride[1] : src.stride[0];
  const int ya = max( y - 1, 0 ), yb = min( y + 1, chh - 1 );
  const uint4 cv = *reinterpret_cast<const uint4*>( &S[(size_t) y * st + xl] );
  const uint4 av = *reinterpret_cast<const uint4*>( &S[(size_t) ya * st + xl] );
  const uint4 bv = *reinterpret_cast<const uint4*>( &S[(size_t) yb * st + xl] );
  const bool enabled = pic.sao && ( pic.hdr.tool_flags & ( c ? VVR_TOOL_SAO_CHROMA : VVR_TOOL_SAO_LUMA ) );
  const int curCtu = ( y / ctuC ) * pic.ctus_x + ( xl / ctuC );
  int mode = 0, type = 0, bandPos = 0; int off[4] = { 0, 0, 0, 0 };
  if( enabled )
  {
    const vvr_sao_ctu s = pic.sao[curCtu];
    mode = c == 0 ? s.mode[0] : c == 1 ? s.mode[1] : s.mode[2]; type = c == 0 ? s.type[0] : c == 1 ? s.type[1] : s.type[2]; bandPos = c == 0 ? s.band_pos[0] : c == 1 ? s.band_pos[1] : s.band_pos[2];
#pragma unroll
    for( int k = 0; k < 4; k++ ) off[k] = c == 0 ? s.offset[0][k] : c == 1 ? s.offset[1][k] : s.offset[2][k];
  }
  // outer samples of the three runs: lane - 1's last and lane + 1's first sample; the wavefront's first / last lane load them
  int cl = __shfl_up( (int) ( cv.w >> 16 ), 1 ), al = __shfl_up( (int) ( av.w >> 16 ), 1 ), bl = __shfl_up( (int) ( bv.w >> 16 ), 1 );
  int cr = __shfl_down( (int) ( cv.x & 0xffff ), 1 ), ar = __shfl_down( (int) ( av.x & 0xffff ), 1 ), br = __shfl_down( (int) ( bv.x & 0xffff ), 1 );
  if( lane == 0 && x0 > 0 ) { cl = S[(size_t) y * st + x0 - 1]; al = S[(size_t) ya * st + x0 - 1]; bl = S[(size_t) yb * st + x0 - 1]; }
  if( lane == 63 && x0 + 8 < cw ) { cr = S[(size_t) y * st + x0 + 8]; ar = S[(size_t) ya * st + x0 + 8]; br = S[(size_t) yb * st + x0 + 8]; }
  if( !inside ) return;
  int v[8] = { (int) ( cv.x & 0xffff ), (int) ( cv.x >> 16 ), (int) ( cv.y & 0xffff ), (int) ( cv.y >> 16 ), (int) ( cv.z & 0xffff ), (int) ( cv.z >> 16 ), (int) ( cv.w & 0xffff ), (int) ( cv.w >> 16 ) };
  int out[8];
#pragma unroll
  for( int i = 0; i < 8; i++ ) out[i] = v[i];
  if( mode )
  {
    if( type == 4 )
    {
      // band offset (SampleAdaptiveOffset.cpp offsetBlock_core, SAO_TYPE_BO)
#pragma unroll
      for( int i = 0; i < 8; i++ ) { const int k = ( ( v[i] >> ( bd - 5 ) ) - bandPos ) & 31; if( k < 4 ) out[i] = clip_pel( v[i] + ( k == 0 ? off[0] : k == 1 ? off[1] : k == 2 ? off[2] : off[3] ), bd ); }
    }
    else
    {
      // edge offset: neighbours a = ( x - dx, y - dy ), b = ( x + dx, y + dy ); nothing across the picture boundary
      const int dx = type == 1 ? 0 : ( type == 3 ? -1 : 1 ), dy = type == 0 ? 0 : 1;
      if( y - dy >= 0 && y + dy < chh )
      {
        // windows [x0 - 1, x0 + 8] of the two neighbour rows (the row itself for the horizontal class)
        int wa[10], wb[10];
        {
          const uint4 ua = dy ? av : cv, ub = dy ? bv : cv;
          wa[1] = ua.x & 0xffff; wa[2] = ua.x >> 16; wa[3] = ua.y & 0xffff; wa[4] = ua.y >> 16; wa[5] = ua.z & 0xffff; wa[6] = ua.z >> 16; wa[7] = ua.w & 0xffff; wa[8] = ua.w >> 16;
          wb[1] = ub.x & 0xffff; wb[2] = ub.x >> 16; wb[3] = ub.y & 0xffff; wb[4] = ub.y >> 16; wb[5] = ub.z & 0xffff; wb[6] = ub.z >> 16; wb[7] = ub.w & 0xffff; wb[8] = ub.w >> 16;
          wa[0] = dy ? al : cl; wb[0] = dy ? bl : cl; wa[9] = dy ? ar : cr; wb[9] = dy ? br : cr;
        }
        const bool restricted = lf_restricted( pic ), vbOn = vb_present( pic );
        const int yA = y - dy, yB = y + dy;
#pragma unroll
        for( int i = 0; i < 8; i++ )
        {
          const int x = x0 + i;
          if( x - dx < 0 || x - dx >= cw || x + dx < 0 || x + dx >= cw ) continue;
          // nothing across a slice / tile boundary the loop filters must not cross
          if( restricted && ( !lf_may_cross( pic, curCtu, ( yA / ctuC ) * pic.ctus_x + ( x - dx ) / ctuC ) || !lf_may_cross( pic, curCtu, ( yB / ctuC ) * pic.ctus_x + ( x + dx ) / ctuC ) ) ) continue;
          if( vbOn && sao_at_vb( pic, x, y, cs, dx != 0, dy != 0 ) ) continue;
          // (static indices + selects: run-time indices would put the windows into LDS)
          const int na = dx == 0 ? wa[1 + i] : dx == 1 ? wa[i] : wa[2 + i], nb = dx == 0 ? wb[1 + i] : dx == 1 ? wb[2 + i] : wb[i];
          const int e = sgn( v[i] - na ) + sgn( v[i] - nb );
          if( e ) out[i] = clip_pel( v[i] + ( e == -2 ? off[0] : e == -1 ? off[1] : e == 1 ? off[2] : off[3] ), bd );      // (selects: an indexed private array would be put into LDS)
        }
      }
    }
  }
  pel_t* __restrict__ D = c == 0 ? dst.p[0] : c == 1 ? dst.p[1] : dst.p[2];
  *reinterpret_cast<uint4*>( &D[(size_t) y * ( c ? dst.stride[1] : dst.stride[0] ) + x0] ) =
      make_uint4( (uint32_t) out[0] | ( (uint32_t) out[1] << 16 ), (uint32_t) out[2] | ( (uint32_t) out[3] << 16 ), (uint32_t) out[4] | ( (uint32_t) out[5] << 16 ), (uint32_t) out[6] | ( (uint32_t) out[7] << 16 ) );
}
void launch_sao( hipStream_t s, const PicDev& pic, DevPlanes src, DevPlanes dst )
{
  const int ncomp = pic.hdr.chroma_format ? 3 : 1;
  hipLaunchKernelGGL( k_sao, dim3( ( src.w[0] + 511 ) / 512, ( src.h[0] + 3 ) / 4, ncomp ), dim3( 256 ), 0, s, pic, src, dst );
}

// =====================================================================================================================
// k_alf — AdaptiveLoopFilter::filterCTU (AdaptiveLoopFilter.cpp:664) for the no-virtual-boundary / single-slice case:
//   deriveClassificationBlk (:969), filterBlk<ALF_FILTER_7/5> (:1176), filterBlkCcAlf (:1348), prepareCTU (:453, border
//   replication = clamped reads).  One workgroup filters a 32x32 luma tile (or 32x32 chroma tile): the tile plus a
//   3-sample halo is staged in LDS; each thread first classifies one 4x4 block (luma), results are exchanged through LDS,
//   then every thread filters 4 samples.
// =====================================================================================================================
#define ALF_T   32
#define ALF_HALO 3
#define ALF_LW  ( ALF_T + 2 * ALF_HALO + 2 )     // LDS row stride (40)

__device__ __forceinline__ int clip_alf( int clip, int ref, int v0, int v1 ) { return clip3( -clip, clip, v0 - ref ) + clip3( -clip, clip, v1 - ref ); }

__constant__ int8_t c_alf_perm[4][12] = {
  { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 },
  { 9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6 },
  { 0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11 },
  { 9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6 } };

template<bool VB /* virtual boundaries of the picture header present */>
__global__ __launch_bounds__( 256 ) void k_alf_luma( PicDev pic, DevPlanes src, DevPlanes dst )
{
  __shared__ pel_t tile[( ALF_T + 2 * ALF_HALO ) * ALF_LW];
  __shared__ int16_t fCoef[25 * 12], fClip[25 * 12];      // the CTU's filter set: per class, un-transposed
  __shared__ uint8_t cls[64], trp[64];
  const int tileLin = xcd_contiguous( blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y );
  const int tx0 = ( tileLin % gridDim.x ) * ALF_T, ty0 = ( tileLin / gridDim.x ) * ALF_T;
  const int W = src.w[0], H = src.h[0], st = src.stride[0];
  const int tid = threadIdx.x, bd = pic.hdr.bit_depth, ctu = 1 << pic.hdr.log2_ctu;
  const pel_t* __restrict__ S = src.p[0];
  // the whole tile lies in one CTU (ALF_T divides the CTU size)
  const vvr_alf_ctu f = pic.alf[( ty0 >> pic.hdr.log2_ctu ) * pic.ctus_x + ( tx0 >> pic.hdr.log2_ctu )];
  if( !f.enable[0] )
  {
    for( int i = tid; i < ALF_T * ALF_T; i += 256 )
    {
      const int y = ty0 + i / ALF_T, x = tx0 + i % ALF_T;
      if( x < W && y < H ) dst.p[0][(size_t) y * dst.stride[0] + x] = S[(size_t) y * st + x];
    }
    return;
  }
  const int TW = ALF_T + 2 * ALF_HALO;
  const AlfClip kctu = alf_clip_of_ctu( pic, tx0 >> pic.hdr.log2_ctu, ty0 >> pic.hdr.log2_ctu, 0 );
  // virtual boundaries (picture header) that run through the tile cut it into parts with a border of their own: one pass per part (usually: one).
  // The parts are walked by their start: the next boundary inside the tile ends a part (no local arrays: they would live in scratch memory).
  // (two instantiations: without virtual boundaries the loops below disappear and the kernel is the single-pass one, 61 VGPRs instead of 102)
  constexpr bool vbOn = VB;
  bool firstPart = true;
  for( int ay0 = ty0; ay0 < ty0 + ALF_T; )
  {
  int ay1 = ty0 + ALF_T;
  if constexpr( VB ) VB_EACH_Y( if( v > ay0 && v < ay1 ) ay1 = v; )
  for( int ax0 = tx0; ax0 < tx0 + ALF_T; )
  {
  int ax1 = tx0 + ALF_T;
  if constexpr( VB ) VB_EACH_X( if( v > ax0 && v < ax1 ) ax1 = v; )
  AlfClip kclip = kctu;
  if constexpr( VB ) kclip = alf_clip_vb( pic, kctu, ax0, ay0, 0 );
  const int part = firstPart ? 0 : 1;
  if( part ) __syncthreads();                                     // the previous part is done with the tile and the classes
  firstPart = false;
  for( int i = tid; i < TW * TW; i += 256 )
  {
    const int yy = i / TW, xx = i - yy * TW;
    int sx = tx0 - ALF_HALO + xx, sy = ty0 - ALF_HALO + yy;
    alf_clip_coord( kclip, sx, sy );                              // slice / tile boundaries the filter must not cross
    sx = clip3( 0, W - 1, sx ); sy = clip3( 0, H - 1, sy );
    tile[yy * ALF_LW + xx] = S[(size_t) sy * st + sx];
  }
  if( part == 0 )
  {
    const vvr_alf_params* __restrict__ A = alf_set_at( pic, tx0, ty0 );      // the filters of the APSs the CTU's slice refers to (AdaptiveLoopFilter.cpp:515)
    const int clipDef = 1 << bd;      // m_alfClippVls[bd-8][0] = 256 << (bd - 8)
    for( int i = tid; i < 25 * 12; i += 256 )
    {
      const int cl = i / 12, k = i - cl * 12;
      if( f.luma_filter_idx < 16 ) { fCoef[i] = d_alf_fixed_coeff[d_alf_class_to_filter[f.luma_filter_idx][cl]][k]; fClip[i] = (int16_t) clipDef; }
      else { fCoef[i] = A->luma_coeff[f.luma_filter_idx - 16][cl][k]; fClip[i] = A->luma_clip[f.luma_filter_idx - 16][cl][k]; }
    }
  }
  __syncthreads();
#define T( x, y ) tile[( ( y ) + ALF_HALO ) * ALF_LW + ( x ) + ALF_HALO]      // tile-relative sample
  const int vbPos = ctu - 4;
  {
    // ---- classification: 4 lanes per 4x4 block, one Laplacian cell-row each, reduced with lane shuffles
    const int blk = tid >> 2, i = ( tid & 3 ) * 2;
    const int bx = ( blk & 7 ) * 4, by = ( blk >> 3 ) * 4;
    const int yInCtu = ( ty0 + by ) & ( ctu - 1 );
    int sumV = 0, sumH = 0, sumD0 = 0, sumD1 = 0;
    const bool inPart = tx0 + bx >= ax0 && tx0 + bx < ax1 && ty0 + by >= ay0 && ty0 + by < ay1;      // (4x4 blocks never straddle a boundary: multiples of 8)
    if( inPart && !( ( yInCtu == vbPos - 4 && i == 6 ) || ( yInCtu == vbPos && i == 0 ) ) )
    {
      const int r = by - 2 + i, rel = yInCtu - 2 + i;
      int rm1 = r - 1, rp2 = r + 2;
      if( rel > 0 && ( rel % ctu ) == vbPos - 2 ) rp2 = r + 1;
      else if( rel > 0 && ( rel % ctu ) == vbPos ) rm1 = r;
      for( int j = 0; j < 8; j += 2 )
      {
        const int cX = bx - 2 + j;
        const int y0 = T( cX, r ) << 1, yup1 = T( cX + 1, r + 1 ) << 1;
        sumV  += iabs( y0 - T( cX, rm1 ) - T( cX, r + 1 ) )         + iabs( yup1 - T( cX + 1, r ) - T( cX + 1, rp2 ) );
        sumH  += iabs( y0 - T( cX + 1, r ) - T( cX - 1, r ) )       + iabs( yup1 - T( cX + 2, r + 1 ) - T( cX, r + 1 ) );
        sumD0 += iabs( y0 - T( cX - 1, rm1 ) - T( cX + 1, r + 1 ) ) + iabs( yup1 - T( cX, r ) - T( cX + 2, rp2 ) );
        sumD1 += iabs( y0 - T( cX - 1, r + 1 ) - T( cX + 1, rm1 ) ) + iabs( yup1 - T( cX, rp2 ) - T( cX + 2, r ) );
      }
    }
    sumV  += __shfl_xor( sumV, 1 );  sumV  += __shfl_xor( sumV, 2 );
    sumH  += __shfl_xor( sumH, 1 );  sumH  += __shfl_xor( sumH, 2 );
    sumD0 += __shfl_xor( sumD0, 1 ); sumD0 += __shfl_xor( sumD0, 2 );
    sumD1 += __shfl_xor( sumD1, 1 ); sumD1 += __shfl_xor( sumD1, 2 );
    if( ( tid & 3 ) == 0 && inPart )
    {
      const int th[16] = { 0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4 };
      const int act = clip3( 0, 15, ( ( sumV + sumH ) * ( ( yInCtu == vbPos - 4 || yInCtu == vbPos ) ? 96 : 64 ) ) >> ( bd + 4 ) );
      int cl = th[act];
      int hv1, hv0, d1, d0, dirHV, dirD, hvd1, hvd0, mainDir, secDir;
      if( sumV > sumH ) { hv1 = sumV; hv0 = sumH; dirHV = 1; } else { hv1 = sumH; hv0 = sumV; dirHV = 3; }
      if( sumD0 > sumD1 ) { d1 = sumD0; d0 = sumD1; dirD = 0; } else { d1 = sumD1; d0 = sumD0; dirD = 2; }
      if( (uint32_t) d1 * (uint32_t) hv0 > (uint32_t) hv1 * (uint32_t) d0 ) { hvd1 = d1; hvd0 = d0; mainDir = dirD; secDir = dirHV; }
      else { hvd1 = hv1; hvd0 = hv0; mainDir = dirHV; secDir = dirD; }
      int strength = 0;
      if( hvd1 > 2 * hvd0 ) strength = 1;
      if( hvd1 * 2 > 9 * hvd0 ) strength = 2;
      if( strength ) cl += ( ( ( mainDir & 1 ) << 1 ) + strength ) * 5;
      const int tt[8] = { 0, 1, 0, 2, 2, 3, 1, 3 };
      cls[blk] = (uint8_t) cl; trp[blk] = (uint8_t) tt[mainDir * 2 + ( secDir >> 1 )];
    }
  }
  __syncthreads();
  // ---- filtering: thread -> (row, 4 consecutive columns) = one row of one 4x4 block
  {
    const int y = tid >> 3, x4 = ( tid & 7 ) * 4;
    const int gy = ty0 + y;
    if( gy < H && tx0 + x4 < W && tx0 + x4 >= ax0 && tx0 + x4 < ax1 && gy >= ay0 && gy < ay1 )
    {
      const int b = ( y >> 2 ) * 8 + ( x4 >> 2 );
      const int cl = cls[b], tr = trp[b];
      int cf[12], cp[12];
#pragma unroll
      for( int k = 0; k < 12; k++ ) { const int sk = c_alf_perm[tr][k]; cf[k] = fCoef[cl * 12 + sk]; cp[k] = fClip[cl * 12 + sk]; }
      const int yVb = gy & ( ctu - 1 );
      int r1 = y + 1, r2 = y - 1, r3 = y + 2, r4 = y - 2, r5 = y + 3, r6 = y - 3;
      if( yVb < vbPos && yVb >= vbPos - 4 )
      {
        r1 = ( yVb == vbPos - 1 ) ? y : r1;  r3 = ( yVb >= vbPos - 2 ) ? r1 : r3;  r5 = ( yVb >= vbPos - 3 ) ? r3 : r5;
        r2 = ( yVb == vbPos - 1 ) ? y : r2;  r4 = ( yVb >= vbPos - 2 ) ? r2 : r4;  r6 = ( yVb >= vbPos - 3 ) ? r4 : r6;
      }
      else if( yVb >= vbPos && yVb <= vbPos + 3 )
      {
        r2 = ( yVb == vbPos ) ? y : r2;  r4 = ( yVb <= vbPos + 1 ) ? r2 : r4;  r6 = ( yVb <= vbPos + 2 ) ? r4 : r6;
        r1 = ( yVb == vbPos ) ? y : r1;  r3 = ( yVb <= vbPos + 1 ) ? r1 : r3;  r5 = ( yVb <= vbPos + 2 ) ? r3 : r5;
      }
      const bool nearVb = ( yVb == vbPos - 1 ) || ( yVb == vbPos );
#pragma unroll
      for( int xx = x4; xx < x4 + 4; xx++ )
      {
        if( tx0 + xx >= W ) break;
        const int cur = T( xx, y );
        int sum = 0;
        sum += cf[0]  * clip_alf( cp[0],  cur, T( xx, r5 ),     T( xx, r6 ) );
        sum += cf[1]  * clip_alf( cp[1],  cur, T( xx + 1, r3 ), T( xx - 1, r4 ) );
        sum += cf[2]  * clip_alf( cp[2],  cur, T( xx, r3 ),     T( xx, r4 ) );
        sum += cf[3]  * clip_alf( cp[3],  cur, T( xx - 1, r3 ), T( xx + 1, r4 ) );
        sum += cf[4]  * clip_alf( cp[4],  cur, T( xx + 2, r1 ), T( xx - 2, r2 ) );
        sum += cf[5]  * clip_alf( cp[5],  cur, T( xx + 1, r1 ), T( xx - 1, r2 ) );
        sum += cf[6]  * clip_alf( cp[6],  cur, T( xx, r1 ),     T( xx, r2 ) );
        sum += cf[7]  * clip_alf( cp[7],  cur, T( xx - 1, r1 ), T( xx + 1, r2 ) );
        sum += cf[8]  * clip_alf( cp[8],  cur, T( xx - 2, r1 ), T( xx + 2, r2 ) );
        sum += cf[9]  * clip_alf( cp[9],  cur, T( xx + 3, y ),  T( xx - 3, y ) );
        sum += cf[10] * clip_alf( cp[10], cur, T( xx + 2, y ),  T( xx - 2, y ) );
        sum += cf[11] * clip_alf( cp[11], cur, T( xx + 1, y ),  T( xx - 1, y ) );
        sum = nearVb ? ( sum + 512 ) >> 10 : ( sum + 64 ) >> 7;
        dst.p[0][(size_t) gy * dst.stride[0] + tx0 + xx] = (pel_t) clip_pel( sum + cur, bd );
      }
    }
  }
  ax0 = ax1;
  }     // parts of a row of parts
  ay0 = ay1;
  }     // rows of parts
#undef T
}

// chroma 5x5 diamond + CC-ALF: one thread per chroma sample, straight from global memory (L1/L2 serve the 13 + 8 taps)
__global__ __launch_bounds__( 256 ) void k_alf_chroma( PicDev pic, DevPlanes src, DevPlanes dst )
{
  const int c = 1 + blockIdx.z;
  const int blkLin = xcd_contiguous( blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y );
  const int x = ( blkLin % gridDim.x ) * 64 + ( threadIdx.x & 63 ), y = ( blkLin / gridDim.x ) * 4 + ( threadIdx.x >> 6 );
  const int W = src.w[c], H = src.h[c];
  if( x >= W || y >= H ) return;
  const int bd = pic.hdr.bit_depth, ctu = 1 << pic.hdr.log2_ctu, ctuC = ctu >> 1;
  const vvr_alf_ctu& f = pic.alf[( y / ctuC ) * pic.ctus_x + ( x / ctuC )];
  const pel_t* __restrict__ S = src.p[c];
  const int st = src.stride[c];
  AlfClip kc = alf_clip_of_ctu( pic, x / ctuC, y / ctuC, 1 ), kl = alf_clip_of_ctu( pic, x / ctuC, y / ctuC, 0 );
  if( vb_present( pic ) ) { kc = alf_clip_vb( pic, kc, x << 1, y << 1, 1 ); kl = alf_clip_vb( pic, kl, x << 1, y << 1, 0 ); }
  auto fetch = [&]( const pel_t* __restrict__ P, int pst, int PW, int PH, const AlfClip& k, int xx, int yy ) -> int
  {
    alf_clip_coord( k, xx, yy );
    return P[(size_t) clip3( 0, PH - 1, yy ) * pst + clip3( 0, PW - 1, xx )];
  };
#define C( xx, yy ) fetch( S, st, W, H, kc, ( xx ), ( yy ) )
  const int cur = S[(size_t) y * st + x];
  int v = cur;
  const vvr_alf_params* __restrict__ A = alf_set_at( pic, x << 1, y << 1 );
  if( f.enable[c] )
  {
    const int16_t* cf = A->chroma_coeff[f.alt[c - 1]]; const int16_t* cp = A->chroma_clip[f.alt[c - 1]];
    const int vbPos = ctuC - 2, yVb = y & ( ctuC - 1 );
    int r1 = y + 1, r2 = y - 1, r3 = y + 2, r4 = y - 2;
    if( yVb < vbPos && yVb >= vbPos - 2 )
    {
      r1 = ( yVb == vbPos - 1 ) ? y : r1;  r3 = ( yVb >= vbPos - 2 ) ? r1 : r3;
      r2 = ( yVb == vbPos - 1 ) ? y : r2;  r4 = ( yVb >= vbPos - 2 ) ? r2 : r4;
    }
    else if( yVb >= vbPos && yVb <= vbPos + 1 )
    {
      r2 = ( yVb == vbPos ) ? y : r2;  r4 = ( yVb <= vbPos + 1 ) ? r2 : r4;
      r1 = ( yVb == vbPos ) ? y : r1;  r3 = ( yVb <= vbPos + 1 ) ? r1 : r3;
    }
    const bool nearVb = ( yVb == vbPos - 1 ) || ( yVb == vbPos );
    int sum = 0;
    sum += cf[0] * clip_alf( cp[0], cur, C( x, r3 ),     C( x, r4 ) );
    sum += cf[1] * clip_alf( cp[1], cur, C( x + 1, r1 ), C( x - 1, r2 ) );
    sum += cf[2] * clip_alf( cp[2], cur, C( x, r1 ),     C( x, r2 ) );
    sum += cf[3] * clip_alf( cp[3], cur, C( x - 1, r1 ), C( x + 1, r2 ) );
    sum += cf[4] * clip_alf( cp[4], cur, C( x + 2, y ),  C( x - 2, y ) );
    sum += cf[5] * clip_alf( cp[5], cur, C( x + 1, y ),  C( x - 1, y ) );
    sum = nearVb ? ( sum + 512 ) >> 10 : ( sum + 64 ) >> 7;
    v = clip_pel( sum + cur, bd );
  }
#undef C
  if( ( pic.hdr.tool_flags & VVR_TOOL_CCALF ) && f.cc_idc[c - 1] )
  {
    const int16_t* cf = A->ccalf_coeff[c - 1][f.cc_idc[c - 1] - 1];
    const pel_t* __restrict__ L = src.p[0];
    const int ls = src.stride[0], LW = src.w[0], LH = src.h[0];
#define Y( xx, yy ) fetch( L, ls, LW, LH, kl, ( xx ), ( yy ) )
    const int vbPos = ctu - 4, lx = x << 1, ly = y << 1, pos = ly & ( ctu - 1 );
    int o1 = 1, o2 = -1, o3 = 2;
    if( pos == vbPos - 2 || pos == vbPos + 1 ) o3 = o1;
    else if( pos == vbPos - 1 || pos == vbPos ) { o1 = 0; o2 = 0; o3 = 0; }
    const int cc = Y( lx, ly );
    int sum = 0;
    sum += cf[0] * ( Y( lx,     ly + o2 ) - cc );
    sum += cf[1] * ( Y( lx - 1, ly      ) - cc );
    sum += cf[2] * ( Y( lx + 1, ly      ) - cc );
    sum += cf[3] * ( Y( lx - 1, ly + o1 ) - cc );
    sum += cf[4] * ( Y( lx,     ly + o1 ) - cc );
    sum += cf[5] * ( Y( lx + 1, ly + o1 ) - cc );
    sum += cf[6] * ( Y( lx,     ly + o3 ) - cc );
#undef Y
    sum = ( sum + 64 ) >> 7;
    const int off = 1 << bd >> 1;
    sum = clip_pel( sum + off, bd ) - off;
    v = clip_pel( v + sum, bd );
  }
  dst.p[c][(size_t) y * dst.stride[c] + x] = (pel_t) v;
}

// The same for the common case (no picture-header virtual boundaries, CTU of at least 64): one workgroup filters a 32x32 tile of BOTH chroma planes -
// the tile lies in one chroma CTU, so one clip rectangle serves it.  Tile + 2-sample halo of each plane and, when the CTU uses CC-ALF, the co-located
// 66x66 luma window sit in LDS (16-byte / coalesced loads instead of 21 two-byte loads per sample: the per-sample kernel is bound by its load
// instructions).  Arithmetic, row re-mapping at the CTU-row boundary and clipping are those of k_alf_chroma.
#define ALFC_T   32
#define ALFC_LW  40                    // LDS row stride of a chroma tile (36 used)
#define ALFC_LL  72                    // LDS row stride of the luma window (66 used)
__global__ __launch_bounds__( 256 ) void k_alf_chroma_tile( PicDev pic, DevPlanes src, DevPlanes dst )
{
  __shared__ pel_t tc[2][( ALFC_T + 4 ) * ALFC_LW];
  __shared__ pel_t tl[( 2 * ALFC_T + 2 ) * ALFC_LL];
  const int tileLin = xcd_contiguous( blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y );
  const int tx0 = ( tileLin % gridDim.x ) * ALFC_T, ty0 = ( tileLin / gridDim.x ) * ALFC_T;
  const int W = src.w[1], H = src.h[1], st = src.stride[1];
  const int tid = threadIdx.x, bd = pic.hdr.bit_depth, ctu = 1 << pic.hdr.log2_ctu, ctuC = ctu >> 1;
  const vvr_alf_ctu f = pic.alf[( ty0 / ctuC ) * pic.ctus_x + ( tx0 / ctuC )];
  const bool ccOn = ( pic.hdr.tool_flags & VVR_TOOL_CCALF ) != 0;
  const bool cc[2] = { ccOn && f.cc_idc[0], ccOn && f.cc_idc[1] };
  const AlfClip kc = alf_clip_of_ctu( pic, tx0 / ctuC, ty0 / ctuC, 1 ), kl = alf_clip_of_ctu( pic, tx0 / ctuC, ty0 / ctuC, 0 );
  // ---- stage the tiles (clipped coordinates = what the filter of this CTU may read)
  for( int i = tid; i < 2 * ( ALFC_T + 4 ) * ( ALFC_T + 4 ); i += 256 )
  {
    const int k = i / ( ( ALFC_T + 4 ) * ( ALFC_T + 4 ) ), r = i - k * ( ALFC_T + 4 ) * ( ALFC_T + 4 ), yy = r / ( ALFC_T + 4 ), xx = r - yy * ( ALFC_T + 4 );
    if( !f.enable[1 + k] && !cc[k] ) continue;
    int sx = tx0 - 2 + xx, sy = ty0 - 2 + yy;
    alf_clip_coord( kc, sx, sy );
    tc[k][yy * ALFC_LW + xx] = ( k ? src.p[2] : src.p[1] )[(size_t) clip3( 0, H - 1, sy ) * st + clip3( 0, W - 1, sx )];
  }
  if( cc[0] || cc[1] )
  {
    const pel_t* __restrict__ L = src.p[0];
    const int ls = src.stride[0], LW = src.w[0], LH = src.h[0];
    for( int i = tid; i < ( 2 * ALFC_T + 2 ) * ( 2 * ALFC_T + 2 ); i += 256 )
    {
      const int yy = i / ( 2 * ALFC_T + 2 ), xx = i - yy * ( 2 * ALFC_T + 2 );
      int sx = 2 * tx0 - 1 + xx, sy = 2 * ty0 - 1 + yy;
      alf_clip_coord( kl, sx, sy );
      tl[yy * ALFC_LL + xx] = L[(size_t) clip3( 0, LH - 1, sy ) * ls + clip3( 0, LW - 1, sx )];
    }
  }
  __syncthreads();
  // ---- thread -> (row, 4 consecutive columns) of the tile, both planes
  const int ly = tid >> 3, lx4 = ( tid & 7 ) * 4;
  const int y = ty0 + ly;
  if( y >= H || tx0 + lx4 >= W ) return;
  const vvr_alf_params* __restrict__ A = alf_set_at( pic, tx0 << 1, ty0 << 1 );      // (a tile lies inside one CTU)
  // rows of the 5x5 diamond at the ALF line-buffer boundary of the CTU row (chroma: 2 rows above the CTU's last 2)
  const int vbPos = ctuC - 2, yVb = y & ( ctuC - 1 );
  int r1 = ly + 1, r2 = ly - 1, r3 = ly + 2, r4 = ly - 2;
  if( yVb < vbPos && yVb >= vbPos - 2 )
  {
    r1 = ( yVb == vbPos - 1 ) ? ly : r1;  r3 = ( yVb >= vbPos - 2 ) ? r1 : r3;
    r2 = ( yVb == vbPos - 1 ) ? ly : r2;  r4 = ( yVb >= vbPos - 2 ) ? r2 : r4;
  }
  else if( yVb >= vbPos && yVb <= vbPos + 1 )
  {
    r2 = ( yVb == vbPos ) ? ly : r2;  r4 = ( yVb <= vbPos + 1 ) ? r2 : r4;
    r1 = ( yVb == vbPos ) ? ly : r1;  r3 = ( yVb <= vbPos + 1 ) ? r1 : r3;
  }
  const bool nearVb = ( yVb == vbPos - 1 ) || ( yVb == vbPos );
  // luma rows of the CC-ALF cross (filterBlkCcAlf :1348)
  const int vbL = ctu - 4, posL = ( y << 1 ) & ( ctu - 1 );
  int o1 = 1, o2 = -1, o3 = 2;
  if( posL == vbL - 2 || posL == vbL + 1 ) o3 = o1;
  else if( posL == vbL - 1 || posL == vbL ) { o1 = 0; o2 = 0; o3 = 0; }
#pragma unroll
  for( int k = 0; k < 2; k++ )
  {
    pel_t* __restrict__ D = k ? dst.p[2] : dst.p[1];
    const pel_t* __restrict__ S = k ? src.p[2] : src.p[1];
    const bool en = f.enable[1 + k] != 0;
    if( !en && !cc[k] )
    {
      for( int xx = lx4; xx < lx4 + 4 && tx0 + xx < W; xx++ ) D[(size_t) y * dst.stride[1] + tx0 + xx] = S[(size_t) y * st + tx0 + xx];
      continue;
    }
#define C( xx, rr ) ( (int) tc[k][( ( rr ) + 2 ) * ALFC_LW + ( xx ) + 2] )
#define Y( xx, yy ) ( (int) tl[( ( yy ) + 1 ) * ALFC_LL + ( xx ) + 1] )
    const int16_t* cf = A->chroma_coeff[k ? f.alt[1] : f.alt[0]]; const int16_t* cp = A->chroma_clip[k ? f.alt[1] : f.alt[0]];
    const int16_t* ccf = cc[k] ? A->ccalf_coeff[k][( k ? f.cc_idc[1] : f.cc_idc[0] ) - 1] : nullptr;
    for( int xx = lx4; xx < lx4 + 4; xx++ )
    {
      if( tx0 + xx >= W ) break;
      const int cur = C( xx, ly );
      int v = cur;
      if( en )
      {
        int sum = 0;
        sum += cf[0] * clip_alf( cp[0], cur, C( xx, r3 ),     C( xx, r4 ) );
        sum += cf[1] * clip_alf( cp[1], cur, C( xx + 1, r1 ), C( xx - 1, r2 ) );
        sum += cf[2] * clip_alf( cp[2], cur, C( xx, r1 ),     C( xx, r2 ) );
        sum += cf[3] * clip_alf( cp[3], cur, C( xx - 1, r1 ), C( xx + 1, r2 ) );
        sum += cf[4] * clip_alf( cp[4], cur, C( xx + 2, ly ), C( xx - 2, ly ) );
        sum += cf[5] * clip_alf( cp[5], cur, C( xx + 1, ly ), C( xx - 1, ly ) );
        sum = nearVb ? ( sum + 512 ) >> 10 : ( sum + 64 ) >> 7;
        v = clip_pel( sum + cur, bd );
      }
      if( ccf )
      {
        const int qx = 2 * xx, qy = 2 * ly;
        const int cl = Y( qx, qy );
        int sum = 0;
        sum += ccf[0] * ( Y( qx,     qy + o2 ) - cl );
        sum += ccf[1] * ( Y( qx - 1, qy      ) - cl );
        sum += ccf[2] * ( Y( qx + 1, qy      ) - cl );
        sum += ccf[3] * ( Y( qx - 1, qy + o1 ) - cl );
        sum += ccf[4] * ( Y( qx,     qy + o1 ) - cl );
        sum += ccf[5] * ( Y( qx + 1, qy + o1 ) - cl );
        sum += ccf[6] * ( Y( qx,     qy + o3 ) - cl );
        sum = ( sum + 64 ) >> 7;
        const int off = 1 << bd >> 1;
        sum = clip_pel( sum + off, bd ) - off;
        v = clip_pel( v + sum, bd );
      }
      D[(size_t) y * dst.stride[1] + tx0 + xx] = (pel_t) v;
    }
#undef C
#undef Y
  }
}

void launch_alf( hipStream_t s, const PicDev& pic, DevPlanes src, DevPlanes dst )
{
  const dim3 grid( ( src.w[0] + ALF_T - 1 ) / ALF_T, ( src.h[0] + ALF_T - 1 ) / ALF_T );
  if( pic.hdr.num_ver_vb | pic.hdr.num_hor_vb ) hipLaunchKernelGGL( k_alf_luma<true>, grid, dim3( 256 ), 0, s, pic, src, dst );
  else hipLaunchKernelGGL( k_alf_luma<false>, grid, dim3( 256 ), 0, s, pic, src, dst );
  if( pic.hdr.chroma_format )
  {
    if( !( pic.hdr.num_ver_vb | pic.hdr.num_hor_vb ) && pic.hdr.log2_ctu >= 6 )
      hipLaunchKernelGGL( k_alf_chroma_tile, dim3( ( src.w[1] + ALFC_T - 1 ) / ALFC_T, ( src.h[1] + ALFC_T - 1 ) / ALFC_T ), dim3( 256 ), 0, s, pic, src, dst );
    else
      hipLaunchKernelGGL( k_alf_chroma, dim3( ( src.w[1] + 63 ) / 64, ( src.h[1] + 3 ) / 4, 2 ), dim3( 256 ), 0, s, pic, src, dst );
  }
}

// =====================================================================================================================
// k_sao_alf — SAO, ALF and CC-ALF of one 64x64 luma region (and its 32x32 chroma regions) in ONE pass over the deblocked picture:
//   SampleAdaptiveOffset::offsetBlock_core (SampleAdaptiveOffset.cpp:64) + deriveLoopFilterBoundaryAvailibility (:741), AdaptiveLoopFilter::filterCTU
//   (AdaptiveLoopFilter.cpp:664): deriveClassificationBlk (:969), filterBlk<ALF_FILTER_7 / 5> (:1176), filterBlkCcAlf (:1348).
// The reference runs SAO over the picture into a second picture and ALF back (m_fltBuf); k_sao / k_alf_* did the same: 4 bytes per sample more
// than needed.  Here the deblocked window of the region (+ 4 samples around it: 3 for the 7x7 diamond, 1 for the SAO edge classes) is staged in LDS
// with 16-byte loads, SAO is applied where the window is copied into the ALF tile - with the coordinate clamp of the CTU's ALF (slice / tile /
// picture boundaries: the ALF of a CTU reads replicated samples, each of them a SAO OUTPUT of the sample it replicates, which SAO computes from
// that sample's own neighbours and its own CTU's parameters) -, classification, luma filter, chroma filters and the CC-ALF cross (from the same
// SAO-filtered luma tile) follow from LDS.  The picture is reconstructed into the lane's scratch picture and this kernel writes the DPB slot.
// For CTUs of at least 64 samples and pictures without picture-header virtual boundaries (else: k_sao + k_alf_*).
// =====================================================================================================================
#define SA_T    64
#define SA_DLW  80      // luma window of deblocked samples: columns tx0 - 8 .. tx0 + 71 (16-byte chunks), rows ty0 - 4 .. ty0 + 67
#define SA_DLH  72
#define SA_ALW  72      // SAO-filtered luma tile: rows ty0 - 3 .. ty0 + 66, column index = x - ( tx0 - 4 )
#define SA_ALH  70
#define SA_DCW  40      // chroma windows: columns cx0 - 4 .. cx0 + 35 (8-byte chunks), rows cy0 - 3 .. cy0 + 34
#define SA_DCH  38
#define SA_ACW  40      // SAO-filtered chroma tiles: rows cy0 - 2 .. cy0 + 33, column index = x - ( cx0 - 4 )
#define SA_ACH  36
struct SaoAlfTables {
  uint32_t fPack[4 * 25 * 12];                 // the CTU's luma filter set, per transpose and class the 12 taps in the order the filter reads them: coefficient | clip value << 16
  uint8_t cls[256];                            // per 4x4 block: transpose * 25 + class (row of fPack)
};
// 27.7 KB (round 6; 39 KB before): five workgroups per compute unit instead of four.  The luma window `dl` is dead once SAO has copied it into `al` - the filter
// tables live in its place from then on; Cb and Cr go through ONE pair of chroma buffers, one after the other.
struct SaoAlfShared {
  union { __attribute__( ( aligned( 16 ) ) ) pel_t dl[SA_DLH * SA_DLW]; SaoAlfTables t; };
  __attribute__( ( aligned( 16 ) ) ) pel_t al[SA_ALH * SA_ALW];
  __attribute__( ( aligned( 16 ) ) ) pel_t dc[SA_DCH * SA_DCW];
  __attribute__( ( aligned( 16 ) ) ) pel_t ac[SA_ACH * SA_ACW];
  uint32_t sao[9][3][2];                       // SAO parameters of the 3 x 3 CTUs around the region's, per component: mode | type << 8 | band << 16; the four offsets
};

// SAO of sample (sx, sy) of component C (cs = 1 for chroma): win = the window of deblocked samples (row stride ws) whose entry (0, 0) is sample (wx0, wy0)
template<bool ON>
__device__ __forceinline__ int sao_at( const PicDev& pic, const SaoAlfShared& sh, const pel_t* win, int ws, int wx0, int wy0, int c, int cs, int sx, int sy, int PW, int PH,
                                       int ctuX, int ctuY, bool restricted )
{
  const pel_t* p = win + ( sy - wy0 ) * ws + ( sx - wx0 );
  const int v = p[0];
  if( !ON ) return v;
  const int l2 = pic.hdr.log2_ctu - cs, bd = pic.hdr.bit_depth;
  const int qx = sx >> l2, qy = sy >> l2;
  const uint32_t* P = sh.sao[( qy - ctuY + 1 ) * 3 + ( qx - ctuX + 1 )][c];
  const uint32_t m = P[0], ov = P[1];
  if( !( m & 0xff ) ) return v;
  const int type = ( m >> 8 ) & 0xff;
  int k;
  if( type == 4 )
  {
    k = ( ( v >> ( bd - 5 ) ) - (int) ( m >> 16 ) ) & 31;                                      // band offset
    if( k >= 4 ) return v;
  }
  else
  {
    // edge offset: neighbours a = ( x - dx, y - dy ), b = ( x + dx, y + dy ); nothing across the picture boundary, nor across a slice / tile boundary the filters must not cross
    const int dx = type == 1 ? 0 : ( type == 3 ? -1 : 1 ), dy = type == 0 ? 0 : 1;
    if( sy - dy < 0 || sy + dy >= PH || sx - dx < 0 || sx - dx >= PW || sx + dx < 0 || sx + dx >= PW ) return v;
    if( restricted )
    {
      const int cur = qy * pic.ctus_x + qx;
      if( !lf_may_cross( pic, cur, ( ( sy - dy ) >> l2 ) * pic.ctus_x + ( ( sx - dx ) >> l2 ) ) || !lf_may_cross( pic, cur, ( ( sy + dy ) >> l2 ) * pic.ctus_x + ( ( sx + dx ) >> l2 ) ) ) return v;
    }
    const int o = dy * ws + dx;
    const int e = sgn( v - (int) p[-o] ) + sgn( v - (int) p[o] );
    if( !e ) return v;
    k = e == -2 ? 0 : e == -1 ? 1 : e == 1 ? 2 : 3;
  }
  const int off = (int8_t) ( ov >> ( 8 * k ) );
  return clip_pel( v + off, bd );
}

typedef short alf_s2 __attribute__( ( ext_vector_type( 2 ) ) );      // two neighbouring samples (or their differences) in one register

// SAO of the two samples (x, y), (x + 1, y) (x even) of component c in the register *w of a window of deblocked samples (ws registers per row) that lies inside
// the picture, for a CTU whose filters may read everything around it: no boundary case applies.  Edge classes with packed 16-bit arithmetic.
__device__ __forceinline__ uint32_t sao_pair( const SaoAlfShared& sh, const uint32_t* w, int ws, int c, int l2, int bd, int x, int y, int ctuX, int ctuY )
{
  const uint32_t v = w[0];
  const uint32_t* P = sh.sao[( ( y >> l2 ) - ctuY + 1 ) * 3 + ( ( x >> l2 ) - ctuX + 1 )][c];
  const uint32_t m = P[0];
  if( !( m & 0xff ) ) return v;
  const uint32_t ov = P[1];
  const int type = ( m >> 8 ) & 0xff;
  const int v0 = v & 0xffff, v1 = v >> 16;
  int k0, k1;                                            // class of the two samples; >= 4: none
  if( type == 4 )
  {
    const int bp = (int) ( m >> 16 );
    k0 = ( ( v0 >> ( bd - 5 ) ) - bp ) & 31; k1 = ( ( v1 >> ( bd - 5 ) ) - bp ) & 31;
  }
  else
  {
    uint32_t a, b;
    if( type == 0 )      { a = __builtin_amdgcn_alignbit( v, w[-1], 16 );           b = __builtin_amdgcn_alignbit( w[1], v, 16 ); }
    else if( type == 1 ) { a = w[-ws];                                              b = w[ws]; }
    else if( type == 2 ) { a = __builtin_amdgcn_alignbit( w[-ws], w[-ws - 1], 16 ); b = __builtin_amdgcn_alignbit( w[ws + 1], w[ws], 16 ); }
    else                 { a = __builtin_amdgcn_alignbit( w[-ws + 1], w[-ws], 16 ); b = __builtin_amdgcn_alignbit( w[ws], w[ws - 1], 16 ); }
    const alf_s2 vs = __builtin_bit_cast( alf_s2, v ), one = alf_s2{ 1, 1 }, mone = alf_s2{ -1, -1 };
    const alf_s2 e = __builtin_elementwise_min( __builtin_elementwise_max( vs - __builtin_bit_cast( alf_s2, a ), mone ), one )
                   + __builtin_elementwise_min( __builtin_elementwise_max( vs - __builtin_bit_cast( alf_s2, b ), mone ), one );
    const int e0 = e.x, e1 = e.y;
    k0 = e0 == 0 ? 4 : e0 < 0 ? e0 + 2 : e0 + 1; k1 = e1 == 0 ? 4 : e1 < 0 ? e1 + 2 : e1 + 1;
  }
  const int o0 = k0 < 4 ? (int) (int8_t) ( ov >> ( 8 * k0 ) ) : 0, o1 = k1 < 4 ? (int) (int8_t) ( ov >> ( 8 * k1 ) ) : 0;
  return (uint32_t) clip_pel( v0 + o0, bd ) | ( (uint32_t) clip_pel( v1 + o1, bd ) << 16 );
}

// acc + ( low / high half of d ) * ( low half of pk ): v_mad_i32_i16 with its operand-select bits
__device__ __forceinline__ int alf_mad_lo( alf_s2 d, uint32_t pk, int acc ) { int r; asm( "v_mad_i32_i16 %0, %1, %2, %3" : "=v"( r ) : "v"( __builtin_bit_cast( uint32_t, d ) ), "v"( pk ), "v"( acc ) ); return r; }
__device__ __forceinline__ int alf_mad_hi( alf_s2 d, uint32_t pk, int acc ) { int r; asm( "v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"( r ) : "v"( __builtin_bit_cast( uint32_t, d ) ), "v"( pk ), "v"( acc ) ); return r; }

template<bool SAO, bool ALF>
__global__ __launch_bounds__( 256 ) void k_sao_alf( PicDev pic, DevPlanes src, DevPlanes dst )
{
  __shared__ SaoAlfShared sh;
  const int tilesX = gridDim.x;
  const int tileLin = xcd_contiguous( blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y );
  const int tx0 = ( tileLin % tilesX ) * SA_T, ty0 = ( tileLin / tilesX ) * SA_T;
  const int tid = threadIdx.x, bd = pic.hdr.bit_depth, l2 = pic.hdr.log2_ctu, ctu = 1 << l2, ctuC = ctu >> 1;
  const int W = src.w[0], H = src.h[0], st = src.stride[0];
  const bool chroma = pic.hdr.chroma_format != 0;
  const int CW = src.w[1], CH = src.h[1], cst = src.stride[1];
  const int cx0 = tx0 >> 1, cy0 = ty0 >> 1;
  const int ctuX = tx0 >> l2, ctuY = ty0 >> l2;                // (the region lies in one CTU)
#ifndef SA_SKIP
#define SA_SKIP 0          /* developer builds: phases left out for timing (1 SAO, 2 classification, 4 luma filter, 8 chroma filters, 16 CC-ALF); results are wrong */
#endif
  const bool saoL = !( SA_SKIP & 1 ) && SAO && pic.sao && ( pic.hdr.tool_flags & VVR_TOOL_SAO_LUMA ), saoC = !( SA_SKIP & 1 ) && SAO && pic.sao && ( pic.hdr.tool_flags & VVR_TOOL_SAO_CHROMA );
  // the window of deblocked samples of one chroma component
  auto loadChroma = [&]( int k )
  {
    const pel_t* __restrict__ C = k ? src.p[2] : src.p[1];
    for( int j = tid; j < SA_DCH * ( SA_DCW / 4 ); j += 256 )
    {
      const int r = j / ( SA_DCW / 4 ), cch = j - r * ( SA_DCW / 4 );
      const int y = clip3( 0, CH - 1, cy0 - 3 + r ), x = cx0 - 4 + 4 * cch;
      uint2 v;
      if( x >= 0 && x < CW ) v = *reinterpret_cast<const uint2*>( &C[(size_t) y * cst + x] );   // (the chroma width is a multiple of 4)
      else { const uint32_t e = (uint16_t) C[(size_t) y * cst + ( x < 0 ? 0 : CW - 1 )]; v.x = v.y = e | ( e << 16 ); }
      *reinterpret_cast<uint2*>( &sh.dc[r * SA_DCW + 4 * cch] ) = v;
    }
  };
  // ---- the windows of deblocked samples; coordinates outside the picture repeat its border (the ALF's clamp; SAO never uses such a neighbour)
  {
    const pel_t* __restrict__ S = src.p[0];
    for( int i = tid; i < SA_DLH * ( SA_DLW / 8 ); i += 256 )
    {
      const int r = i / ( SA_DLW / 8 ), cch = i - r * ( SA_DLW / 8 );
      const int y = clip3( 0, H - 1, ty0 - 4 + r ), x = tx0 - 8 + 8 * cch;
      uint4 v;
      if( x >= 0 && x < W ) v = *reinterpret_cast<const uint4*>( &S[(size_t) y * st + x] );      // (the picture width is a multiple of 8: a chunk lies inside or outside)
      else { const uint32_t e = (uint16_t) S[(size_t) y * st + ( x < 0 ? 0 : W - 1 )]; v.x = v.y = v.z = v.w = e | ( e << 16 ); }
      *reinterpret_cast<uint4*>( &sh.dl[r * SA_DLW + 8 * cch] ) = v;
    }
    if( chroma ) loadChroma( 0 );
    if( SAO && pic.sao && tid < 27 )
    {
      const int q = tid / 3, c = tid - q * 3;
      const int nx = clip3( 0, pic.ctus_x - 1, ctuX + q % 3 - 1 ), ny = clip3( 0, pic.ctus_y - 1, ctuY + q / 3 - 1 );
      const vvr_sao_ctu* __restrict__ sp = &pic.sao[ny * pic.ctus_x + nx];
      const uint8_t* ob = reinterpret_cast<const uint8_t*>( sp->offset[c] );
      sh.sao[q][c][0] = (uint32_t) sp->mode[c] | ( (uint32_t) sp->type[c] << 8 ) | ( (uint32_t) sp->band_pos[c] << 16 );
      sh.sao[q][c][1] = (uint32_t) ob[0] | ( (uint32_t) ob[1] << 8 ) | ( (uint32_t) ob[2] << 16 ) | ( (uint32_t) ob[3] << 24 );
    }
  }
  vvr_alf_ctu f; f.enable[0] = f.enable[1] = f.enable[2] = 0; f.cc_idc[0] = f.cc_idc[1] = 0; f.alt[0] = f.alt[1] = 0; f.luma_filter_idx = 0;
  const vvr_alf_params* __restrict__ A = nullptr;
  if( ALF )
  {
    f = pic.alf[ctuY * pic.ctus_x + ctuX];
    A = alf_set_at( pic, tx0, ty0 );                     // the filters of the APSs the CTU's slice refers to (AdaptiveLoopFilter.cpp:515)
  }
  const bool ccOn = ALF && ( pic.hdr.tool_flags & VVR_TOOL_CCALF ) != 0;
  const bool cc[2] = { ccOn && f.cc_idc[0], ccOn && f.cc_idc[1] };
  const bool restricted = lf_restricted( pic );
  const AlfClip kl = alf_clip_of_ctu( pic, ctuX, ctuY, 0 ), kc = alf_clip_of_ctu( pic, ctuX, ctuY, 1 );
  __syncthreads();
  // ---- SAO where the windows are copied into the tiles the ALF reads: entry (x, y) of a tile is the SAO output of the sample the CTU's ALF reads there
  // A region whose windows lie inside the picture, in a CTU whose filters may read everything around it (nearly all): no clamp, no boundary case -
  // two neighbouring samples per step; else sample by sample with every rule
  const bool plain = !kl.f && !kc.f && !restricted && tx0 >= 8 && ty0 >= 8 && tx0 + SA_T + 4 <= W && ty0 + SA_T + 4 <= H;
  // SAO of chroma component k: its window `dc` into the tile `ac`
  auto saoChroma = [&]( int k )
  {
    if( plain )
    {
      for( int r = tid; r < SA_ACH * 18; r += 256 )
      {
        const int ay = r / 18, j = 1 + ( r - ay * 18 );
        const uint32_t* wp = reinterpret_cast<const uint32_t*>( sh.dc ) + ( ay + 1 ) * ( SA_DCW / 2 ) + j;
        reinterpret_cast<uint32_t*>( sh.ac )[ay * ( SA_ACW / 2 ) + j] = saoC ? sao_pair( sh, wp, SA_DCW / 2, 1 + k, l2 - 1, bd, cx0 - 4 + 2 * j, cy0 - 2 + ay, ctuX, ctuY ) : wp[0];
      }
    }
    else
    {
      for( int j = tid; j < SA_ACH * 36; j += 256 )
      {
        const int ay = j / 36, ax = j - ay * 36;
        int sx = cx0 - 2 + ax, sy = cy0 - 2 + ay;
        alf_clip_coord( kc, sx, sy );
        sx = clip3( 0, CW - 1, sx ); sy = clip3( 0, CH - 1, sy );
        sh.ac[ay * SA_ACW + ax + 2] = (pel_t) ( saoC ? sao_at<true>( pic, sh, sh.dc, SA_DCW, cx0 - 4, cy0 - 3, 1 + k, 1, sx, sy, CW, CH, ctuX, ctuY, restricted )
                                                     : sao_at<false>( pic, sh, sh.dc, SA_DCW, cx0 - 4, cy0 - 3, 1 + k, 1, sx, sy, CW, CH, ctuX, ctuY, restricted ) );
      }
    }
  };
  if( plain )
  {
    const uint32_t* dlw = reinterpret_cast<const uint32_t*>( sh.dl );
    uint32_t* alw = reinterpret_cast<uint32_t*>( sh.al );
    for( int i = tid; i < SA_ALH * ( SA_ALW / 2 ); i += 256 )
    {
      const int ay = i / ( SA_ALW / 2 ), j = i - ay * ( SA_ALW / 2 );
      const uint32_t* wp = dlw + ( ay + 1 ) * ( SA_DLW / 2 ) + j + 2;
      alw[ay * ( SA_ALW / 2 ) + j] = saoL ? sao_pair( sh, wp, SA_DLW / 2, 0, l2, bd, tx0 - 4 + 2 * j, ty0 - 3 + ay, ctuX, ctuY ) : wp[0];
    }
    if( chroma ) saoChroma( 0 );
  }
  else
  {
    for( int i = tid; i < SA_ALH * 70; i += 256 )
    {
      const int ay = i / 70, ax = i - ay * 70;
      int sx = tx0 - 3 + ax, sy = ty0 - 3 + ay;
      alf_clip_coord( kl, sx, sy );
      sx = clip3( 0, W - 1, sx ); sy = clip3( 0, H - 1, sy );
      sh.al[ay * SA_ALW + ax + 1] = (pel_t) ( saoL ? sao_at<true>( pic, sh, sh.dl, SA_DLW, tx0 - 8, ty0 - 4, 0, 0, sx, sy, W, H, ctuX, ctuY, restricted )
                                                   : sao_at<false>( pic, sh, sh.dl, SA_DLW, tx0 - 8, ty0 - 4, 0, 0, sx, sy, W, H, ctuX, ctuY, restricted ) );
    }
    if( chroma ) saoChroma( 0 );
  }
  __syncthreads();
  // ---- the CTU's luma filter set (where the luma window was: SAO is through with it)
  int lumaClips = 0;                                     // some tap of the CTU's luma filter set clips its differences (else the clipping is left out: a fixed set, or an APS whose clip indices are 0)
  if( ALF && f.enable[0] )
  {
    const int clipDef = 1 << bd;                       // m_alfClippVls[bd-8][0] = 256 << (bd - 8): such a tap is never clipped
    for( int i = tid; i < 4 * 25 * 12; i += 256 )
    {
      const int tr = i / 300, r = i - tr * 300, cl = r / 12, k = c_alf_perm[tr][r - cl * 12];
      int cf, cp;
      if( f.luma_filter_idx < 16 ) { cf = d_alf_fixed_coeff[d_alf_class_to_filter[f.luma_filter_idx][cl]][k]; cp = clipDef; }
      else { cf = A->luma_coeff[f.luma_filter_idx - 16][cl][k]; cp = A->luma_clip[f.luma_filter_idx - 16][cl][k]; }
      sh.t.fPack[i] = (uint32_t) (uint16_t) cf | ( (uint32_t) (uint16_t) cp << 16 );
      lumaClips |= cp < clipDef;
    }
  }
  const bool lumaClip = __syncthreads_or( lumaClips ) != 0;
#define T( x, y ) sh.al[( ( y ) + 3 ) * SA_ALW + ( x ) + 4]      // region-relative luma sample
  const int vbPos = ctu - 4;
  if( !( SA_SKIP & 2 ) && ALF && f.enable[0] )
  {
    // ---- classification: ONE LANE per 4x4 block (round 6; before: four lanes per block, a cell-row pair each, lane shuffles, the decision by all four - 256
    // blocks in four rounds of the workgroup): the four cell-row pairs one after the other, Laplacians of two cells per packed operation, sums in the two
    // halves of a register (16 cells of at most 2 * 1023 per half), no cross-lane traffic
    {
      const int blk = tid;
      const int bx = ( blk & 15 ) * 4, by = ( blk >> 4 ) * 4;
      const int yInCtu = ( ty0 + by ) & ( ctu - 1 );
      alf_s2 aV = alf_s2{ 0, 0 }, aH = aV, aD0 = aV, aD1 = aV;
      const alf_s2 zero = alf_s2{ 0, 0 };
#define LOHI( L, Hh ) __builtin_bit_cast( alf_s2, __builtin_amdgcn_perm( Hh, L, 0x07060100u ) )
#define ALN( Hh, L )  __builtin_bit_cast( alf_s2, __builtin_amdgcn_alignbit( Hh, L, 16 ) )
#define LAPL( ACC, B, C ) { const alf_s2 t_ = a2 - ( B ) - ( C ); ACC += __builtin_elementwise_max( t_, zero - t_ ); }
#pragma unroll
      for( int i = 0; i < 8; i += 2 )
      {
        if( ( yInCtu == vbPos - 4 && i == 6 ) || ( yInCtu == vbPos && i == 0 ) ) continue;
        const int r = by - 2 + i, rel = yInCtu - 2 + i;
        int rm1 = r - 1, rp2 = r + 2;
        if( rel > 0 && ( rel % ctu ) == vbPos - 2 ) rp2 = r + 1;
        else if( rel > 0 && ( rel % ctu ) == vbPos ) rm1 = r;
        // rows rm1, r, r + 1, rp2, columns bx - 4 .. bx + 7 (three 8-byte reads each).  The Laplacians of the two cells of a step - (cX, r) and
        // (cX + 1, r + 1) - are computed together in the two halves of a register: | 2 A - B - C | with A = the cells, B / C = their two neighbours
        uint32_t wr[4][6];
        {
          const int rr[4] = { rm1, r, r + 1, rp2 };
#pragma unroll
          for( int k = 0; k < 4; k++ )
          {
            const uint2* rp = reinterpret_cast<const uint2*>( &sh.al[( rr[k] + 3 ) * SA_ALW + bx] );
            const uint2 u0 = rp[0], u1 = rp[1], u2 = rp[2];
            wr[k][0] = u0.x; wr[k][1] = u0.y; wr[k][2] = u1.x; wr[k][3] = u1.y; wr[k][4] = u2.x; wr[k][5] = u2.y;
          }
        }
#pragma unroll
        for( int d = 1; d <= 4; d++ )          // cX = bx - 2 + 2 ( d - 1 ): register d of a row holds columns ( cX, cX + 1 )
        {
          const alf_s2 a1 = LOHI( wr[1][d], wr[2][d] ), a2 = a1 + a1;
          LAPL( aV,  LOHI( wr[0][d], wr[1][d] ),    LOHI( wr[2][d], wr[3][d] ) )
          LAPL( aH,  ALN( wr[2][d + 1], wr[1][d] ), ALN( wr[2][d], wr[1][d - 1] ) )
          LAPL( aD0, ALN( wr[1][d], wr[0][d - 1] ), ALN( wr[3][d + 1], wr[2][d] ) )
          LAPL( aD1, ALN( wr[3][d], wr[2][d - 1] ), ALN( wr[1][d + 1], wr[0][d] ) )
        }
      }
#undef LAPL
#undef ALN
#undef LOHI
#define HSUM( A ) ( (int) ( __builtin_bit_cast( uint32_t, A ) & 0xffffu ) + (int) ( __builtin_bit_cast( uint32_t, A ) >> 16 ) )
      const int sumV = HSUM( aV ), sumH = HSUM( aH ), sumD0 = HSUM( aD0 ), sumD1 = HSUM( aD1 );
#undef HSUM
      {
        const int act = clip3( 0, 15, ( ( sumV + sumH ) * ( ( yInCtu == vbPos - 4 || yInCtu == vbPos ) ? 96 : 64 ) ) >> ( bd + 4 ) );
        int cl = (int) ( ( 0x4333333332222210ull >> ( 4 * act ) ) & 15 );          // { 0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4 }
        int hv1, hv0, d1, d0, dirHV, dirD, hvd1, hvd0, mainDir, secDir;
        if( sumV > sumH ) { hv1 = sumV; hv0 = sumH; dirHV = 1; } else { hv1 = sumH; hv0 = sumV; dirHV = 3; }
        if( sumD0 > sumD1 ) { d1 = sumD0; d0 = sumD1; dirD = 0; } else { d1 = sumD1; d0 = sumD0; dirD = 2; }
        if( (uint32_t) d1 * (uint32_t) hv0 > (uint32_t) hv1 * (uint32_t) d0 ) { hvd1 = d1; hvd0 = d0; mainDir = dirD; secDir = dirHV; }
        else { hvd1 = hv1; hvd0 = hv0; mainDir = dirHV; secDir = dirD; }
        int strength = 0;
        if( hvd1 > 2 * hvd0 ) strength = 1;
        if( hvd1 * 2 > 9 * hvd0 ) strength = 2;
        if( strength ) cl += ( ( ( mainDir & 1 ) << 1 ) + strength ) * 5;
        const int tr = (int) ( ( 0x31322010u >> ( 4 * ( mainDir * 2 + ( secDir >> 1 ) ) ) ) & 15 );      // { 0, 1, 0, 2, 2, 3, 1, 3 }
        sh.t.cls[blk] = (uint8_t) ( tr * 25 + cl );
      }
    }
    __syncthreads();
  }
  // ---- luma: thread -> (row, 4 consecutive columns) = one row of one 4x4 block, four rounds
  {
    pel_t* __restrict__ D = dst.p[0];
    const int dstride = dst.stride[0];
#pragma unroll 1
    for( int q = 0; q < 4; q++ )
    {
      const int task = q * 256 + tid, y = task >> 4, x4 = ( task & 15 ) * 4;
      const int gy = ty0 + y;
      if( gy >= H || tx0 + x4 >= W ) continue;
      int o[4];
      if( !( SA_SKIP & 4 ) && ALF && f.enable[0] )
      {
        const int b = ( y >> 2 ) * 16 + ( x4 >> 2 );
        // the 12 taps of the block's class, in the order of its transpose: coefficient | clip value << 16 (three 16-byte LDS reads)
        uint32_t pk[12];
        {
          const uint4* tp = reinterpret_cast<const uint4*>( &sh.t.fPack[(int) sh.t.cls[b] * 12] );
          const uint4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
          pk[0] = t0.x; pk[1] = t0.y; pk[2] = t0.z; pk[3] = t0.w; pk[4] = t1.x; pk[5] = t1.y; pk[6] = t1.z; pk[7] = t1.w; pk[8] = t2.x; pk[9] = t2.y; pk[10] = t2.z; pk[11] = t2.w;
        }
        const int yVb = gy & ( ctu - 1 );
        int r1 = y + 1, r2 = y - 1, r3 = y + 2, r4 = y - 2, r5 = y + 3, r6 = y - 3;
        if( yVb < vbPos && yVb >= vbPos - 4 )
        {
          r1 = ( yVb == vbPos - 1 ) ? y : r1;  r3 = ( yVb >= vbPos - 2 ) ? r1 : r3;  r5 = ( yVb >= vbPos - 3 ) ? r3 : r5;
          r2 = ( yVb == vbPos - 1 ) ? y : r2;  r4 = ( yVb >= vbPos - 2 ) ? r2 : r4;  r6 = ( yVb >= vbPos - 3 ) ? r4 : r6;
        }
        else if( yVb >= vbPos && yVb <= vbPos + 3 )
        {
          r2 = ( yVb == vbPos ) ? y : r2;  r4 = ( yVb <= vbPos + 1 ) ? r2 : r4;  r6 = ( yVb <= vbPos + 2 ) ? r4 : r6;
          r1 = ( yVb == vbPos ) ? y : r1;  r3 = ( yVb <= vbPos + 1 ) ? r1 : r3;  r5 = ( yVb <= vbPos + 2 ) ? r3 : r5;
        }
        const bool nearVb = ( yVb == vbPos - 1 ) || ( yVb == vbPos );
        // the rows of the diamond, columns x4 - 4 .. x4 + 7 of each (three 8-byte LDS reads): everything below works on PAIRS of neighbouring samples
        // in the two halves of a register (v_pk_sub / max / min / add_i16, v_dot2_i32_i16): a pair of outputs per pass, 12 taps of ( 2 clipped
        // differences, their sum, two multiply-adds ) each
        uint32_t wv[7][6];
        {
          const int rr[7] = { r6, r4, r2, y, r1, r3, r5 };
#pragma unroll
          for( int k = 0; k < 7; k++ )
          {
            const uint2* rp = reinterpret_cast<const uint2*>( &sh.al[( rr[k] + 3 ) * SA_ALW + x4] );
            const uint2 u0 = rp[0], u1 = rp[1], u2 = rp[2];
            wv[k][0] = u0.x; wv[k][1] = u0.y; wv[k][2] = u1.x; wv[k][3] = u1.y; wv[k][4] = u2.x; wv[k][5] = u2.y;
          }
        }
        // pair of samples at columns ( x4 - 4 + OFF, + 1 ) of row K
#define AP( K, OFF ) ( ( ( OFF ) & 1 ) ? __builtin_amdgcn_alignbit( wv[K][( ( OFF ) + 1 ) >> 1], wv[K][( ( OFF ) - 1 ) >> 1], 16 ) : wv[K][( OFF ) >> 1] )
        // s0 / s1: the sums of the two samples of a pair; a tap adds coefficient * ( two differences to the centre, clipped to the tap's clip value where the CTU's
        // filter set clips at all ): v_mad_i32_i16 takes the coefficient out of the low half of the packed tap and the difference out of either half
#define ALF_MADS( D, PK ) { s0 = alf_mad_lo( D, PK, s0 ); s1 = alf_mad_hi( D, PK, s1 ); }
        // rows: 0 = r6, 1 = r4, 2 = r2, 3 = y, 4 = r1, 5 = r3, 6 = r5; OFF = 2 p + dx + 4
#define ALF_TAPS \
          TAP( 0, 6, 0, 0, 0 ) TAP( 1, 5, 1, 1, -1 ) TAP( 2, 5, 0, 1, 0 ) TAP( 3, 5, -1, 1, 1 ) TAP( 4, 4, 2, 2, -2 ) TAP( 5, 4, 1, 2, -1 ) \
          TAP( 6, 4, 0, 2, 0 ) TAP( 7, 4, -1, 2, 1 ) TAP( 8, 4, -2, 2, 2 ) TAP( 9, 3, 3, 3, -3 ) TAP( 10, 3, 2, 3, -2 ) TAP( 11, 3, 1, 3, -1 )
        if( lumaClip )
        {
          alf_s2 cpP[12], cpN[12];
#pragma unroll
          for( int k = 0; k < 12; k++ ) { cpP[k] = __builtin_bit_cast( alf_s2, __builtin_amdgcn_perm( pk[k], pk[k], 0x07060706u ) ); cpN[k] = alf_s2{ 0, 0 } - cpP[k]; }
#pragma unroll
          for( int p = 0; p < 2; p++ )
          {
            const alf_s2 cur = __builtin_bit_cast( alf_s2, wv[3][2 + p] );
            int s0 = 0, s1 = 0;
#define TAP( K, KA, DXA, KB, DXB ) { const alf_s2 a = __builtin_bit_cast( alf_s2, AP( KA, 2 * p + ( DXA ) + 4 ) ), b = __builtin_bit_cast( alf_s2, AP( KB, 2 * p + ( DXB ) + 4 ) ); \
              const alf_s2 d = __builtin_elementwise_min( __builtin_elementwise_max( a - cur, cpN[K] ), cpP[K] ) + __builtin_elementwise_min( __builtin_elementwise_max( b - cur, cpN[K] ), cpP[K] ); \
              ALF_MADS( d, pk[K] ) }
            ALF_TAPS
#undef TAP
            s0 = nearVb ? ( s0 + 512 ) >> 10 : ( s0 + 64 ) >> 7;
            s1 = nearVb ? ( s1 + 512 ) >> 10 : ( s1 + 64 ) >> 7;
            o[2 * p] = clip_pel( s0 + (int) cur.x, bd ); o[2 * p + 1] = clip_pel( s1 + (int) cur.y, bd );
          }
        }
        else
        {
#pragma unroll
          for( int p = 0; p < 2; p++ )
          {
            const alf_s2 cur = __builtin_bit_cast( alf_s2, wv[3][2 + p] ), cur2 = cur + cur;
            int s0 = 0, s1 = 0;
#define TAP( K, KA, DXA, KB, DXB ) { const alf_s2 d = __builtin_bit_cast( alf_s2, AP( KA, 2 * p + ( DXA ) + 4 ) ) + __builtin_bit_cast( alf_s2, AP( KB, 2 * p + ( DXB ) + 4 ) ) - cur2; ALF_MADS( d, pk[K] ) }
            ALF_TAPS
#undef TAP
            s0 = nearVb ? ( s0 + 512 ) >> 10 : ( s0 + 64 ) >> 7;
            s1 = nearVb ? ( s1 + 512 ) >> 10 : ( s1 + 64 ) >> 7;
            o[2 * p] = clip_pel( s0 + (int) cur.x, bd ); o[2 * p + 1] = clip_pel( s1 + (int) cur.y, bd );
          }
        }
#undef ALF_TAPS
#undef ALF_MADS
#undef AP
      }
      else
      {
#pragma unroll
        for( int e = 0; e < 4; e++ ) o[e] = (uint16_t) T( x4 + e, y );
      }
      *reinterpret_cast<uint2*>( &D[(size_t) gy * dstride + tx0 + x4] ) = make_uint2( (uint32_t) o[0] | ( (uint32_t) o[1] << 16 ), (uint32_t) o[2] | ( (uint32_t) o[3] << 16 ) );
    }
  }
  if( !chroma ) return;
  // ---- chroma: thread -> (row, 4 consecutive columns) of the 32x32 region, Cb then Cr through the one pair of buffers: 5x5 diamond + the CC-ALF cross over the luma tile
  {
    const int ly = tid >> 3, lx4 = ( tid & 7 ) * 4;
    const int y = cy0 + ly;
    const bool inPic = y < CH && cx0 + lx4 < CW;
    // rows of the 5x5 diamond at the ALF line-buffer boundary of the CTU row (chroma: 2 rows above the CTU's last 2)
    const int vbC = ctuC - 2, yVb = y & ( ctuC - 1 );
    int r1 = ly + 1, r2 = ly - 1, r3 = ly + 2, r4 = ly - 2;
    if( yVb < vbC && yVb >= vbC - 2 )
    {
      r1 = ( yVb == vbC - 1 ) ? ly : r1;  r3 = ( yVb >= vbC - 2 ) ? r1 : r3;
      r2 = ( yVb == vbC - 1 ) ? ly : r2;  r4 = ( yVb >= vbC - 2 ) ? r2 : r4;
    }
    else if( yVb >= vbC && yVb <= vbC + 1 )
    {
      r2 = ( yVb == vbC ) ? ly : r2;  r4 = ( yVb <= vbC + 1 ) ? r2 : r4;
      r1 = ( yVb == vbC ) ? ly : r1;  r3 = ( yVb <= vbC + 1 ) ? r1 : r3;
    }
    const bool nearVb = ( yVb == vbC - 1 ) || ( yVb == vbC );
    // luma rows of the CC-ALF cross (filterBlkCcAlf :1348)
    const int posL = ( y << 1 ) & ( ctu - 1 );
    int o1 = 1, o2 = -1, o3 = 2;
    if( posL == vbPos - 2 || posL == vbPos + 1 ) o3 = o1;
    else if( posL == vbPos - 1 || posL == vbPos ) { o1 = 0; o2 = 0; o3 = 0; }
#pragma unroll
    for( int k = 0; k < 2; k++ )
    {
      if( k )
      {
        __syncthreads();          // Cb is through with the buffers
        loadChroma( 1 );
        __syncthreads();
        saoChroma( 1 );
        __syncthreads();
      }
      if( !inPic ) continue;
      pel_t* __restrict__ D = k ? dst.p[2] : dst.p[1];
      const bool en = !( SA_SKIP & 8 ) && ALF && f.enable[1 + k] != 0;
      const int16_t* cf = nullptr; const int16_t* cp = nullptr; const int16_t* ccf = nullptr;
      if( en ) { cf = A->chroma_coeff[k ? f.alt[1] : f.alt[0]]; cp = A->chroma_clip[k ? f.alt[1] : f.alt[0]]; }
      if( !( SA_SKIP & 16 ) && cc[k] ) ccf = A->ccalf_coeff[k][( k ? f.cc_idc[1] : f.cc_idc[0] ) - 1];
      int o[4];
      // the rows of the 5x5 diamond, columns lx4 - 4 .. lx4 + 7 (pairs of neighbouring samples per register, as for luma)
      uint32_t wv[5][6];
      {
        const int rr[5] = { r4, r2, ly, r1, r3 };
#pragma unroll
        for( int j = 0; j < 5; j++ )
        {
          const uint2* rp = reinterpret_cast<const uint2*>( &sh.ac[( rr[j] + 2 ) * SA_ACW + lx4] );
          const uint2 u0 = rp[0], u1 = rp[1], u2 = rp[2];
          wv[j][0] = u0.x; wv[j][1] = u0.y; wv[j][2] = u1.x; wv[j][3] = u1.y; wv[j][4] = u2.x; wv[j][5] = u2.y;
        }
      }
#pragma unroll
      for( int e = 0; e < 4; e++ ) o[e] = (int) ( ( wv[2][2 + ( e >> 1 )] >> ( 16 * ( e & 1 ) ) ) & 0xffff );
      if( en )
      {
#define AP( K, OFF ) ( ( ( OFF ) & 1 ) ? __builtin_amdgcn_alignbit( wv[K][( ( OFF ) + 1 ) >> 1], wv[K][( ( OFF ) - 1 ) >> 1], 16 ) : wv[K][( OFF ) >> 1] )
        uint32_t ck[6]; alf_s2 cpP[6], cpN[6];
#pragma unroll
        for( int j = 0; j < 6; j++ ) { ck[j] = (uint16_t) cf[j]; const short c = (short) cp[j]; cpP[j] = alf_s2{ c, c }; cpN[j] = alf_s2{ (short) -c, (short) -c }; }
#pragma unroll
        for( int p = 0; p < 2; p++ )
        {
          const alf_s2 cur = __builtin_bit_cast( alf_s2, wv[2][2 + p] );
          int s0 = 0, s1 = 0;
          // rows: 0 = r4, 1 = r2, 2 = ly, 3 = r1, 4 = r3
#define TAP( K, KA, DXA, KB, DXB ) { const alf_s2 a = __builtin_bit_cast( alf_s2, AP( KA, 2 * p + ( DXA ) + 4 ) ), b = __builtin_bit_cast( alf_s2, AP( KB, 2 * p + ( DXB ) + 4 ) ); \
            const alf_s2 d = __builtin_elementwise_min( __builtin_elementwise_max( a - cur, cpN[K] ), cpP[K] ) + __builtin_elementwise_min( __builtin_elementwise_max( b - cur, cpN[K] ), cpP[K] ); \
            s0 = __builtin_amdgcn_sdot2( d, __builtin_bit_cast( alf_s2, ck[K] ), s0, false ); s1 = __builtin_amdgcn_sdot2( d, __builtin_bit_cast( alf_s2, ck[K] << 16 ), s1, false ); }
          TAP( 0, 4, 0, 0, 0 )
          TAP( 1, 3, 1, 1, -1 )
          TAP( 2, 3, 0, 1, 0 )
          TAP( 3, 3, -1, 1, 1 )
          TAP( 4, 2, 2, 2, -2 )
          TAP( 5, 2, 1, 2, -1 )
#undef TAP
          s0 = nearVb ? ( s0 + 512 ) >> 10 : ( s0 + 64 ) >> 7;
          s1 = nearVb ? ( s1 + 512 ) >> 10 : ( s1 + 64 ) >> 7;
          o[2 * p] = clip_pel( s0 + (int) cur.x, bd ); o[2 * p + 1] = clip_pel( s1 + (int) cur.y, bd );
        }
#undef AP
      }
      if( ccf )
      {
        // the cross over the SAO-filtered luma tile: rows 2 ly + { o2, 0, o1, o3 }, columns 2 lx4 - 4 .. 2 lx4 + 7; the centre's weight is minus the sum of the others
        // ( sum of c * ( Y - centre ) ), two samples of a row per multiply-add (v_dot2_i32_i16)
        uint32_t lw[4][6];
        {
          const int rr[4] = { 2 * ly + o2, 2 * ly, 2 * ly + o1, 2 * ly + o3 };
#pragma unroll
          for( int j = 0; j < 4; j++ )
          {
            const uint2* rp = reinterpret_cast<const uint2*>( &sh.al[( rr[j] + 3 ) * SA_ALW + 2 * lx4] );
            const uint2 u0 = rp[0], u1 = rp[1], u2 = rp[2];
            lw[j][0] = u0.x; lw[j][1] = u0.y; lw[j][2] = u1.x; lw[j][3] = u1.y; lw[j][4] = u2.x; lw[j][5] = u2.y;
          }
        }
        const int csum = ccf[0] + ccf[1] + ccf[2] + ccf[3] + ccf[4] + ccf[5] + ccf[6];
#define PK( LO, HI ) __builtin_bit_cast( alf_s2, (uint32_t) (uint16_t) ( LO ) | ( (uint32_t) (uint16_t) ( HI ) << 16 ) )
        const alf_s2 q1 = PK( 0, ccf[1] ), q2 = PK( -csum, ccf[2] ), q3 = PK( 0, ccf[3] ), q45 = PK( ccf[4], ccf[5] ), q0 = PK( ccf[0], 0 ), q6 = PK( ccf[6], 0 );
#undef PK
        const int off = 1 << bd >> 1;
#pragma unroll
        for( int e = 0; e < 4; e++ )
        {
          // pixel e: luma columns ( 2 ( lx4 + e ), + 1 ) = register 2 + e of a row, ( - 2, - 1 ) = register 1 + e
          int sum = 0;
#define LP( R, J ) __builtin_bit_cast( alf_s2, lw[R][J] )
          sum = __builtin_amdgcn_sdot2( LP( 1, 1 + e ), q1, sum, false );
          sum = __builtin_amdgcn_sdot2( LP( 1, 2 + e ), q2, sum, false );
          sum = __builtin_amdgcn_sdot2( LP( 2, 1 + e ), q3, sum, false );
          sum = __builtin_amdgcn_sdot2( LP( 2, 2 + e ), q45, sum, false );
          sum = __builtin_amdgcn_sdot2( LP( 0, 2 + e ), q0, sum, false );
          sum = __builtin_amdgcn_sdot2( LP( 3, 2 + e ), q6, sum, false );
#undef LP
          sum = ( sum + 64 ) >> 7;
          sum = clip_pel( sum + off, bd ) - off;
          o[e] = clip_pel( o[e] + sum, bd );
        }
      }
      *reinterpret_cast<uint2*>( &D[(size_t) y * dst.stride[1] + cx0 + lx4] ) = make_uint2( (uint32_t) o[0] | ( (uint32_t) o[1] << 16 ), (uint32_t) o[2] | ( (uint32_t) o[3] << 16 ) );
    }
  }
#undef T
}

// the fused pass applies to pictures without picture-header virtual boundaries whose CTUs hold whole 64x64 regions
bool sao_alf_fused( const PicDev& pic ) { return !( pic.hdr.num_ver_vb | pic.hdr.num_hor_vb ) && pic.hdr.log2_ctu >= 6; }
void launch_sao_alf( hipStream_t s, const PicDev& pic, DevPlanes src, DevPlanes dst, bool sao, bool alf )
{
  const dim3 grid( ( src.w[0] + SA_T - 1 ) / SA_T, ( src.h[0] + SA_T - 1 ) / SA_T );
  if( sao && alf ) hipLaunchKernelGGL( ( k_sao_alf<true, true> ), grid, dim3( 256 ), 0, s, pic, src, dst );
  else if( sao )   hipLaunchKernelGGL( ( k_sao_alf<true, false> ), grid, dim3( 256 ), 0, s, pic, src, dst );
  else             hipLaunchKernelGGL( ( k_sao_alf<false, true> ), grid, dim3( 256 ), 0, s, pic, src, dst );
}

// plain plane copy (used when a stage is disabled for a picture)
// =====================================================================================================================
// k_lmcs — luma mapping with chroma scaling, the luma part: forward mapping of the inter prediction (Reshape::rspBufFwd,
// Reshape.cpp:413, rspFwdCore Buffer.cpp:321; DecCu.cpp:458-476) before the residual is added, and the inverse mapping of the
// whole reconstructed picture before the in-loop filters (Reshape::rspCtuBcw :376, applyLutCore Buffer.cpp:200).  The piecewise
// linear maps arrive as look-up tables (vvr_lmcs_params); one workgroup handles 2048 samples of a row, the table sits in LDS.
// =====================================================================================================================
__global__ __launch_bounds__( 256 ) void k_lmcs( PicDev pic, DevPlanes reco, int inverse )
{
  __shared__ int16_t lut[4096];
  const int n = 1 << pic.hdr.bit_depth;
  const int16_t* __restrict__ src = inverse ? pic.lmcs->inv_lut : pic.lmcs->fwd_lut;
  for( int i = threadIdx.x; i < n; i += 256 ) lut[i] = src[i];
  __syncthreads();
  const int y = blockIdx.y, x = ( blockIdx.x * 256 + threadIdx.x ) * 8;
  if( x >= reco.w[0] ) return;
  if( pic.slices && !( flags_at( pic, x, y ) & VVR_TOOL_LMCS ) ) return;       // (8 samples lie inside one CTU) the CTU's slice does not use LMCS (Reshape.cpp:385)
  pel_t* __restrict__ row = reco.p[0] + (size_t) y * reco.stride[0];
  const bool lo = true, hi = true;                 // (round 3: only the inverse pass is left; the forward map is applied by the motion-compensation kernels)
  uint4 v = *reinterpret_cast<const uint4*>( row + x );       // rows are padded to a multiple of 64 samples
  uint32_t* w = reinterpret_cast<uint32_t*>( &v );
  for( int k = 0; k < 4; k++ )
  {
    if( !( k < 2 ? lo : hi ) ) continue;
    const uint32_t a = (uint16_t) lut[( w[k] & 0xffff ) & ( n - 1 )], b = (uint16_t) lut[( w[k] >> 16 ) & ( n - 1 )];
    w[k] = a | ( b << 16 );
  }
  *reinterpret_cast<uint4*>( row + x ) = v;
}

void launch_lmcs( hipStream_t s, const PicDev& pic, DevPlanes reco, int inverse )
{
  hipLaunchKernelGGL( k_lmcs, dim3( ( reco.w[0] + 2047 ) / 2048, reco.h[0] ), dim3( 256 ), 0, s, pic, reco, inverse );
}

// k_copy — copy of one picture (all planes, rows with their padding: both pictures share one geometry), 16 bytes per lane and access.  Also
// the copy kernel the practical HBM ceiling is measured with (vvr_measure_copy_bandwidth).
__global__ __launch_bounds__( 256 ) void k_copy( const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16 )
{
  const size_t step = (size_t) gridDim.x * 256;
  for( size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += step ) dst[i] = src[i];
}
void launch_copy_planes( hipStream_t s, DevPlanes src, DevPlanes dst )
{
  const int ncomp = src.p[1] ? 3 : 1;
  for( int c = 0; c < ncomp; c++ )
  {
    const size_t n16 = (size_t) src.stride[c] * src.h[c] * sizeof( pel_t ) / 16;
    hipLaunchKernelGGL( k_copy, dim3( (unsigned) std::min<size_t>( ( n16 + 255 ) / 256, 256 * 16 ) ), dim3( 256 ), 0, s, (const uint4*) src.p[c], (uint4*) dst.p[c], n16 );
  }
}

void launch_copy_bytes( hipStream_t s, const void* src, void* dst, size_t bytes )
{
  const size_t n16 = bytes / 16;
  hipLaunchKernelGGL( k_copy, dim3( (unsigned) std::min<size_t>( ( n16 + 255 ) / 256, 256 * 32 ) ), dim3( 256 ), 0, s, (const uint4*) src, (uint4*) dst, n16 );
}

// =====================================================================================================================
// output stage: window of a plane packed to the bytes the application / the MD5 wants; per-row CRC and checksum pieces
//   VVDecImpl::copyComp (vvdecimpl.cpp:818-880), compCRC / compChecksum (PicYuvMD5.cpp:99-176)
// =====================================================================================================================
__global__ __launch_bounds__( 256 ) void k_output_window( const pel_t* __restrict__ src, int stride, int w, int h, int bytesPerSample, uint8_t* __restrict__ dst )
{
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if( x >= w ) return;
  const uint16_t v = (uint16_t) src[(size_t) y * stride + x];
  if( bytesPerSample == 2 ) ( (uint16_t*) dst )[(size_t) y * w + x] = v;
  else dst[(size_t) y * w + x] = (uint8_t) v;          // "only narrowing conversions" of 8-bit content (vvdecimpl.cpp:853)
}
void launch_output_window( hipStream_t s, const pel_t* src, int stride, int w, int h, int bytesPerSample, void* dst )
{
  hipLaunchKernelGGL( k_output_window, dim3( ( w + 255 ) / 256, h ), dim3( 256 ), 0, s, src, stride, w, h, bytesPerSample, (uint8_t*) dst );
}

// multiplication in GF(2)[x] / (x^16 + x^12 + x^5 + 1), the ring the CRC of the decoded picture hash lives in
__device__ __forceinline__ uint32_t crc_mul( uint32_t a, uint32_t b )
{
  uint32_t r = 0;
#pragma unroll
  for( int bit = 15; bit >= 0; bit-- ) { r <<= 1; r ^= ( ( r >> 16 ) & 1 ) * 0x11021u; r ^= ( ( b >> bit ) & 1 ) * a; }
  return r;
}
__device__ __forceinline__ uint32_t crc_xpow( uint32_t n ) { uint32_t r = 1, base = 2; while( n ) { if( n & 1 ) r = crc_mul( r, base ); base = crc_mul( base, base ); n >>= 1; } return r; }

// One wavefront per row.  Checksum: the row's share of the 32-bit sum.  CRC: the row's bytes as a polynomial reduced mod P (register value
// from 0); lane l folds samples l, l + 64, ... by Horner's rule ( acc = acc * x^(64 * bits) + sample ), shifts its share to the end of the
// row and the shares are XOR-ed together - the CRC is linear.  The host chains the rows (vvr_picture_hash).
__global__ __launch_bounds__( 64 ) void k_plane_hash_rows( const pel_t* __restrict__ plane, int stride, int w, int two, int crcMode, uint32_t* __restrict__ out )
{
  const int y = blockIdx.x, lane = threadIdx.x;
  const pel_t* __restrict__ row = plane + (size_t) y * stride;
  uint32_t acc = 0;
  if( !crcMode )
  {
    for( int x = lane; x < w; x += 64 )
    {
      const uint32_t v = (uint16_t) row[x], mask = ( ( x & 0xff ) ^ ( y & 0xff ) ^ ( x >> 8 ) ^ ( y >> 8 ) ) & 0xff;
      acc += ( v & 0xff ) ^ mask;
      if( two ) acc += ( v >> 8 ) ^ mask;
    }
    for( int o = 32; o; o >>= 1 ) acc += __shfl_down( acc, o, 64 );
  }
  else
  {
    const int bits = two ? 16 : 8;
    const uint32_t X = crc_xpow( 64 * bits );
    int last = -1;
    for( int x = lane; x < w; x += 64 )
    {
      const uint32_t v = (uint16_t) row[x];
      const uint32_t smp = two ? ( ( v & 0xff ) << 8 ) | ( v >> 8 ) : ( v & 0xff );      // low byte first, most significant bit first
      acc = crc_mul( acc, X ) ^ smp;
      last = x;
    }
    if( last >= 0 ) acc = crc_mul( acc, crc_xpow( (uint32_t) ( w - 1 - last ) * bits ) );
    for( int o = 32; o; o >>= 1 ) acc ^= __shfl_down( acc, o, 64 );
  }
  if( lane == 0 ) out[y] = acc;
}
void launch_plane_hash_rows( hipStream_t s, const pel_t* plane, int stride, int w, int h, int two, int crcMode, uint32_t* out )
{
  hipLaunchKernelGGL( k_plane_hash_rows, dim3( h ), dim3( 64 ), 0, s, plane, stride, w, two, crcMode, out );
}

#include "vvr_intra_cells.inc"

// =====================================================================================================================
// k_intra — intra prediction + reconstruction, one workgroup per (CTU, colour component).
//   DecCu::predAndReco intra branch (DecCu.cpp:271-401), IntraPrediction::xFillReferenceSamples (IntraPrediction.cpp:1072),
//   xFilterReferenceSamples (:1251), useFilteredIntraRefSamples (:1301), xPredIntraPlanarCore (:154), xGetPredValDc (:412),
//   xPredIntraAng (:592), IntraPredSampleFilterCore (:212), xPredIntraBDPCM (:850), AreaBuf::reconstruct (Buffer.cpp:482).
//
// Intra blocks depend on their already reconstructed neighbours, so the blocks of one CTU are processed in decoding
// order by one workgroup, and CTUs run as a wavefront (reference: INTRA state of DecLibRecon::ctuTask, DecLibRecon.cpp:876).
// MI355X mapping: the CTU (plus 3 reference lines above / left and 64 samples above-right) lives in LDS for the whole
// lifetime of the workgroup, so the serial block-to-block dependency never leaves the CU; the CTU-to-CTU dependency is a
// per-(component, CTU) flag in HBM published with an agent-scope release and consumed with one relaxed poll + one
// agent-scope acquire (cdna_hip_programming.md §6 Guideline 16).  Work is handed out through an atomic ticket in raster
// order, so a workgroup only ever waits for CTUs whose workgroups have already started: no residency assumption.
// =====================================================================================================================
#define IT_PAD    3      // reference lines above the CTU (multiRefIdx <= 2)
#define IT_PADX   8      // columns left of the CTU kept in LDS: 8 samples = 16 bytes, so that every tile row starts 16-byte aligned in HBM
#define IT_RIGHT 64
#define IT_TS   ( IT_PADX + 128 + IT_RIGHT + 8 )     // LDS row stride of the rows ABOVE the CTU (they reach 64 samples into the above-right CTU)
#define IT_TSB  ( IT_PADX + 128 )                    // row stride of the rows inside the CTU (nothing right of the CTU is ever available there); 68 dwords: a column read spreads over 16 banks
// index of sample (ox + dx, oy + dy) in the tile: IT_PAD long rows, then the CTU rows
__device__ __forceinline__ int tile_idx( int dx, int dy )
{
  return dy < 0 ? ( dy + IT_PAD ) * IT_TS + dx + IT_PADX : IT_PAD * IT_TS + dy * IT_TSB + dx + IT_PADX;
}
#define IT_MAXREF ( 2 * 64 + 8 )
#define LEAF_PUBLISH_BATCH 1

#define IT_BATCH ( IT_WAVES * 16 )   // IntraItems staged in LDS at a time (16 B each: one dword per thread)
// wavefronts of a workgroup (template parameter of k_intra): each predicts one item (a block of up to 256 samples, or a band of rows of a larger one) at a time.
// 8 for pictures of intra CTUs (the bands of a large block on four wavefronts while the other four prepare the next block), 4 for the scattered intra
// blocks of an inter picture (units of a few blocks: twice the workgroups per CU)
#define IT_NT ( IT_WAVES * 64 )

// scratch of one wavefront (one block at a time)
#define IT_NEG 72           // entries in front of a reference line: the side reference projected onto negative indices of the main reference
struct IntraWave {
  pel_t topB[IT_NEG + IT_MAXREF + 8], leftB[IT_NEG + IT_MAXREF + 8];     // reference lines of the block (index 0 = the corner sample; IT_NEG entries in front)
  pel_t auxT[IT_NEG + IT_MAXREF + 8], auxL[IT_NEG + IT_MAXREF + 8];    // the smoothed lines (same layout); ISP: the line of the whole CU; MIP: the reduced prediction
  int16_t resi[IT_PART_SAMPLES];                      // residual of the block (fetched while the blocks before it are predicted)
  int   lmSel[8];                                     // CCLM: the (luma, chroma) pairs of the selected template positions; MIP: the reduced boundary
};

template<int IT_WAVES>
struct IntraShared {
  pel_t tile[IT_PAD * IT_TS + 128 * IT_TSB];
  IntraWave wave[IT_WAVES];
  IntraItem items[IT_BATCH];                           // the unit's first block records (reference staging, residual-add units, write-back)
  int16_t angTab[32], invAngTab[32], cfilt[32][4];     // the small ROM tables the serial per-block path indexes: LDS latency instead of a memory round trip each
  uint8_t filtThr[8];
  int   ticket;
  int   csFac[4];                                       // LMCS chroma residual scaling factor of the CTU's VPDUs
  int   prog[IT_WAVES];                                 // blocks of the unit each wavefront has finished (wavefront w: blocks w, w + 4, ... in order)
};

__constant__ uint8_t c_intraFilterThr[8] = { 24, 24, 24, 14, 2, 0, 0, 0 };
__constant__ int16_t c_angTable[32]    = { 0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 23, 26, 29, 32, 35, 39, 45, 51, 57, 64, 73, 86, 102, 128, 171, 256, 341, 512, 1024 };
__constant__ int16_t c_invAngTable[32] = { 0, 16384, 8192, 5461, 4096, 2731, 2048, 1638, 1365, 1170, 1024, 910, 819, 712, 630, 565, 512, 468, 420, 364, 321, 287, 256, 224, 191, 161, 128, 96, 64, 48, 32, 16 };

__device__ __forceinline__ int intra_wide_angle( int w, int h, int mode )   // IntraPrediction::getWideAngle (:443)
{
  if( mode > 1 && mode <= 66 )
  {
    const int d = iabs( ilog2( w ) - ilog2( h ) );
    const int modeShift = (int) ( ( 0x0F0E0C0A0600ull >> ( 8 * d ) ) & 0xff );      // { 0, 6, 10, 12, 14, 15 }
    if( w > h && mode < 2 + modeShift ) mode += 65;
    else if( h > w && mode > 66 - modeShift ) mode -= 65;
  }
  return mode;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding vector-memory operation
// (vmcnt(0)), which would put an HBM/L2 round trip on the serial path of k_intra.
__device__ __forceinline__ void lds_barrier() { asm volatile( "s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory" ); }
// Ordering inside ONE wavefront: its LDS operations execute in issue order, so a value another lane of the same wavefront wrote is there
// once the write has been issued; the wait keeps the compiler from moving accesses across and covers the write's completion.
__device__ __forceinline__ void wave_lds_sync() { asm volatile( "s_waitcnt lgkmcnt(0)" ::: "memory" ); }

// sum over the 64 lanes (DPP row shifts inside the rows of 16, then the four row totals through scalar registers): no LDS round trips
__device__ __forceinline__ int wave_sum( int v )
{
  v += __builtin_amdgcn_update_dpp( 0, v, 0x111, 0xf, 0xf, false );     // row_shr:1
  v += __builtin_amdgcn_update_dpp( 0, v, 0x112, 0xf, 0xf, false );     // row_shr:2
  v += __builtin_amdgcn_update_dpp( 0, v, 0x114, 0xf, 0xf, false );     // row_shr:4
  v += __builtin_amdgcn_update_dpp( 0, v, 0x118, 0xf, 0xf, false );     // row_shr:8  -> lane 15 of every row holds the row's sum
  return __builtin_amdgcn_readlane( v, 15 ) + __builtin_amdgcn_readlane( v, 31 ) + __builtin_amdgcn_readlane( v, 47 ) + __builtin_amdgcn_readlane( v, 63 );
}

// the 16-byte record of block q, the same for every lane (one request); decoded into scalar registers by intra_item_of
__device__ __forceinline__ uint4 intra_load_item( const IntraItem* __restrict__ items, uint32_t q ) { return *reinterpret_cast<const uint4*>( &items[q] ); }
__device__ __forceinline__ IntraItem intra_item_of( const uint4 v )
{
  IntraItem it; uint32_t* op = reinterpret_cast<uint32_t*>( &it );
  op[0] = __builtin_amdgcn_readfirstlane( v.x ); op[1] = __builtin_amdgcn_readfirstlane( v.y ); op[2] = __builtin_amdgcn_readfirstlane( v.z ); op[3] = __builtin_amdgcn_readfirstlane( v.w );
  return it;
}

// Residual of a block (its row part) into registers: issued one block ahead of its use by the same wavefront, stored to the wavefront's LDS
// scratch when the block's turn comes.  Four samples (8 bytes) per lane and load where positions allow it (luma always; chroma unless the
// block sits at x = 2 mod 4: the 4-wide chroma of the middle part of a ternary split of 16), else one sample per load (blocks of <= 512 samples).
struct IntraResiRegs { uint2 v[4]; };
__device__ __forceinline__ bool intra_resi_vec4( const IntraItem& it ) { return it.lw >= 2 && !( it.x & 3 ); }
__device__ __forceinline__ int intra_part_rows( const IntraItem& it ) { return ( 1 << it.lh ) >> IT_LPARTS( it ); }
__device__ __forceinline__ void intra_load_resi( IntraResiRegs& R, const IntraItem& it, const pel_t* __restrict__ rs, int rstride, int lane )
{
  if( !( it.flags & IT_F_RESI ) ) return;
  const int lw = it.lw, rows = intra_part_rows( it ), wh = rows << lw, y0 = it.y + IT_PART( it ) * rows;
  if( wh > IT_PART_SAMPLES ) return;                 // (a block that is not split: read where it is added)
  if( intra_resi_vec4( it ) )
  {
    // a CIIP coding unit of several transform units (IntraItem::tu: bit 31; the split at a largest transform size of 32): the residual plane holds something only
    // where a unit was coded - bit k of the word: unit k in raster order, bit 4: two units per row; the others contribute nothing (DecCu.cpp:449-470)
    const bool perUnit = ( it.flags >> 6 ) != 0 && ( it.flags & IT_F_ISP ) != IT_F_ISP && ( it.tu >> 31 ) != 0;
    const int thr = IT_COMP( it ) ? 16 : 32, yBand = IT_PART( it ) * rows;
#pragma unroll
    for( int e = 0; e < 4; e++ )
    {
      const int i = ( e * 64 + lane ) << 2;
      if( e * 256 < wh )
      {
        const int ii = min( i, wh - 4 ), yy = ii >> lw, xx = ii & ( ( 1 << lw ) - 1 );
        R.v[e] = *reinterpret_cast<const uint2*>( &rs[(size_t) ( y0 + yy ) * rstride + it.x + xx] );
        if( perUnit && !( ( it.tu >> ( ( yBand + yy >= thr ? 1 + ( ( it.tu >> 4 ) & 1 ) : 0 ) + ( xx >= thr ? 1 : 0 ) ) ) & 1 ) ) R.v[e] = make_uint2( 0, 0 );
      }
    }
  }
  else
  {
#pragma unroll
    for( int e = 0; e < 8; e++ )
    {
      const int i = e * 64 + lane;
      if( e * 64 < wh ) { const int ii = min( i, wh - 1 ); const uint32_t s = (uint16_t) rs[(size_t) ( y0 + ( ii >> lw ) ) * rstride + it.x + ( ii & ( ( 1 << lw ) - 1 ) )]; if( e & 1 ) R.v[e >> 1].y = s; else R.v[e >> 1].x = s; }
    }
  }
}
__device__ __forceinline__ void intra_stash_resi( const IntraResiRegs& R, const IntraItem& it, int16_t* __restrict__ dst, int lane )
{
  if( !( it.flags & IT_F_RESI ) ) return;
  const int wh = intra_part_rows( it ) << it.lw;
  if( wh > IT_PART_SAMPLES ) return;
  if( intra_resi_vec4( it ) )
  {
#pragma unroll
    for( int e = 0; e < 4; e++ ) { const int i = ( e * 64 + lane ) << 2; if( e * 256 < wh && i < wh ) *reinterpret_cast<uint2*>( &dst[i] ) = R.v[e]; }
  }
  else
  {
#pragma unroll
    for( int e = 0; e < 8; e++ ) { const int i = e * 64 + lane; if( e * 64 < wh && i < wh ) dst[i] = (int16_t) ( ( e & 1 ) ? R.v[e >> 1].y : R.v[e >> 1].x ); }
  }
}

// CCLM / MDLM: the down-sampled luma a chroma block is predicted from (xGetLumaRecPixels, IntraPrediction.cpp:1403-1470; 4:2:0, both luma
// filters) - sample e * 64 + lane of a block of up to IT_CCLM_REGS samples, 16 bits each - and of this lane's template position
// (xGetLMParameters :1694-1800).  It only depends on luma that is final before the chroma unit starts, so it is fetched one block ahead of its
// use, like the residual.
#define IT_CCLM_REGS 256    /* samples of a CCLM block whose luma is fetched ahead (4 per lane); larger blocks fetch it when they are predicted */
struct IntraLumaRegs { uint32_t v[IT_CCLM_REGS / 128]; int tpl; };
// luma of chroma sample (x, y) of the block whose co-located luma block starts at (lx0, ly0); the pairs (2x, 2x + 1) are dword loads
template<bool SC1 = false>
__device__ __forceinline__ int intra_cclm_luma_at( const pel_t* __restrict__ Yp, int ys, int lx0, int ly0, int x, int y, bool bLeft, bool bAbove, bool colloc )
{
  const int xl = ( x == 0 && !bLeft ) ? 0 : 2 * x - 1;
  const pel_t* r0 = Yp + (size_t) ( ly0 + 2 * y ) * ys + lx0;
  const uint32_t m0 = ld_pel2<SC1>( r0 + 2 * x ), m1 = ld_pel2<SC1>( r0 + ys + 2 * x );
  const int a0 = m0 & 0xffff, b0 = m0 >> 16, a1 = m1 & 0xffff, b1 = m1 >> 16;
  if( colloc )
  {
    const int yu = ( y == 0 && !bAbove ) ? 0 : 2 * y - 1;
    return ( ld_pel<SC1>( &Yp[(size_t) ( ly0 + yu ) * ys + lx0 + 2 * x] ) + a0 * 4 + ld_pel<SC1>( &r0[xl] ) + b0 + a1 + 4 ) >> 3;
  }
  return ( a0 * 2 + b0 + ld_pel<SC1>( &r0[xl] ) + a1 * 2 + b1 + ld_pel<SC1>( &r0[ys + xl] ) + 4 ) >> 3;
}
template<bool SC1 = false>
__device__ __forceinline__ void intra_load_cclm_luma( IntraLumaRegs& R, const IntraItem& it, const IntraPic& pic, int lane )
{
  if( it.mode < 67 || it.mode > 69 ) return;
  const pel_t* __restrict__ Yp = pic.plane[0]; const int ys = pic.stride[0];
  const int lw = it.lw, w = 1 << lw, wh = 1 << ( it.lw + it.lh );
  const int lx0 = (int) it.x << 1, ly0 = (int) it.y << 1;
#define LU( xx, yy ) ld_pel<SC1>( &Yp[(size_t) ( ly0 + ( yy ) ) * ys + lx0 + ( xx )] )
  const uint32_t lm = it.tu;
  const int actualTop = lm & 0xff, actualLeft = ( lm >> 8 ) & 0xff;
  const bool aboveAvail = ( lm >> 16 ) & 1, leftAvail = ( lm >> 17 ) & 1, bLeft = ( lm >> 18 ) & 1, firstRow = ( lm >> 19 ) & 1, bAbove = ( lm >> 20 ) & 1;
  const bool colloc = pic.colloc != 0;      // sps_chroma_vertical_collocated_flag: 5-tap cross instead of the 6-tap filter
  const int aboveIs4 = leftAvail ? 0 : 1, leftIs4 = aboveAvail ? 0 : 1;
  const int cntT = aboveAvail ? min( actualTop, ( 1 + aboveIs4 ) << 1 ) : 0, cntL = leftAvail ? min( actualLeft, ( 1 + leftIs4 ) << 1 ) : 0;
  R.tpl = 0;
  if( lane < cntT + cntL )
  {
    int lv;
    if( lane < cntT )
    {
      const int i = ( actualTop >> ( 2 + aboveIs4 ) ) + lane * max( 1, actualTop >> ( 1 + aboveIs4 ) );
      const int xl = ( i == 0 && !bLeft ) ? 2 * i : 2 * i - 1;
      if( firstRow ) lv = ( LU( 2 * i, -1 ) * 2 + LU( xl, -1 ) + LU( 2 * i + 1, -1 ) + 2 ) >> 2;
      else if( colloc ) lv = ( LU( 2 * i, -3 ) + LU( 2 * i, -2 ) * 4 + LU( xl, -2 ) + LU( 2 * i + 1, -2 ) + LU( 2 * i, -1 ) + 4 ) >> 3;
      else           lv = ( LU( 2 * i, -2 ) * 2 + LU( xl, -2 ) + LU( 2 * i + 1, -2 ) + LU( 2 * i, -1 ) * 2 + LU( xl, -1 ) + LU( 2 * i + 1, -1 ) + 4 ) >> 3;
    }
    else
    {
      const int j = ( actualLeft >> ( 2 + leftIs4 ) ) + ( lane - cntT ) * max( 1, actualLeft >> ( 1 + leftIs4 ) );
      if( colloc ) { const int yu = ( j == 0 && !bAbove ) ? 2 * j : 2 * j - 1; lv = ( LU( -2, yu ) + LU( -2, 2 * j ) * 4 + LU( -3, 2 * j ) + LU( -1, 2 * j ) + LU( -2, 2 * j + 1 ) + 4 ) >> 3; }
      else lv = ( LU( -2, 2 * j ) * 2 + LU( -3, 2 * j ) + LU( -1, 2 * j ) + LU( -2, 2 * j + 1 ) * 2 + LU( -3, 2 * j + 1 ) + LU( -1, 2 * j + 1 ) + 4 ) >> 3;
    }
    R.tpl = lv;
  }
#undef LU
  if( wh > IT_CCLM_REGS ) return;
#pragma unroll
  for( int e = 0; e < IT_CCLM_REGS / 64; e++ )
  {
    if( e * 64 < wh )
    {
      const int i = min( e * 64 + lane, wh - 1 );
      const uint32_t t = (uint16_t) intra_cclm_luma_at<SC1>( Yp, ys, lx0, ly0, i & ( w - 1 ), i >> lw, bLeft, bAbove, colloc );
      if( e & 1 ) R.v[e >> 1] = ( R.v[e >> 1] & 0xffffu ) | ( t << 16 ); else R.v[e >> 1] = t;
    }
  }
}

// ---- per-block parameters, computed once per picture by a pass of its own (k_intra_setup: one thread per block) ---------------------------------
// Everything about a block that does not depend on sample values - geometry, where in the tile its reference samples come from, the mode's
// angle and filters, the loop bounds - is uniform scalar work, a few hundred instructions per block.  On a wavefront that predicts the block
// by itself that work sat on the serial path (or kept ~100 scalar registers alive across it: the compiler spilled them into vector lanes).  The
// pass below writes the values to a 64-dword record per block; k_intra loads the record with one instruction (lane k holds value k) and
// takes a value out with v_readlane where it is used.
enum {
  C_FLAGS = 0, C_POS, C_GEO, C_TILEBASE, C_N, C_TB0, C_TS0, C_TB1, C_TL1, C_LB0, C_LS0, C_LB1, C_LL1, C_DCT, C_DCL, C_DCDEN,
  C_TOPLEN, C_LEFTLEN, C_ANGLE, C_INVANGLE, C_REFEND, C_BH, C_PDPCLEV, C_ANGSCALE, C_PSCALE, C_NGROUPS, C_XXB, C_YYB, C_CSIDX, C_ORIGIN,
  C_COUNT
};
#define IT_CTX 64          // dwords per record
// C_FLAGS bits
#define CF_RESI    ( 1u << 0 )
#define CF_CSON    ( 1u << 1 )
#define CF_DC      ( 1u << 2 )
#define CF_FILT    ( 1u << 3 )
#define CF_ANG     ( 1u << 4 )
#define CF_TR      ( 1u << 5 )
#define CF_FRAC    ( 1u << 6 )
#define CF_CUBIC   ( 1u << 7 )
#define CF_VEC     ( 1u << 8 )
#define CF_ANY     ( 1u << 9 )
#define CF_ISP     ( 1u << 10 )
#define CF_MIP     ( 1u << 11 )
#define CF_IBC     ( 1u << 12 )
#define CF_CCLM    ( 1u << 13 )
#define CF_PDPC    ( 1u << 14 )
#define CF_STASHED ( 1u << 15 )
#define CF_PLANAR  ( 1u << 16 )
#define CF_NEG     ( 1u << 17 )
#define CF_ANG0    ( 1u << 18 )
// C_GEO: lw | lh << 4 | log2( rows of the band ) << 8 | first row of the band << 12 (7 bits) | mrl << 19 | wIntra << 21 | bdpcm << 23 | log2 group size << 25 | log2 groups per row << 27 | indep << 30 ... kept in C_WAIT instead
#define CG_LW( g )     ( ( g ) & 15 )
#define CG_LH( g )     ( ( ( g ) >> 4 ) & 15 )
#define CG_LROWS( g )  ( ( ( g ) >> 8 ) & 15 )
#define CG_YB( g )     ( ( ( g ) >> 12 ) & 127 )
#define CG_MRL( g )    ( ( ( g ) >> 19 ) & 3 )
#define CG_WINTRA( g ) ( ( ( g ) >> 21 ) & 3 )
#define CG_BDPCM( g )  ( ( ( g ) >> 23 ) & 3 )
#define CG_GL( g )     ( ( ( g ) >> 25 ) & 3 )
#define CG_LGPR( g )   ( ( ( g ) >> 27 ) & 7 )

__global__ __launch_bounds__( 256 ) void k_intra_setup( IntraPic pic, const IntraItem* __restrict__ items, int numItems, uint32_t* __restrict__ ctx )
{
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if( idx >= numItems ) return;
  const IntraItem it = items[idx];
  if( it.mode == IT_MODE_RESI_ADD ) return;                // (residual-add items never enter the block loop)
  uint32_t* __restrict__ c = ctx + (size_t) idx * IT_CTX;
  const int comp = IT_COMP( it ), cs = comp ? 1 : 0;
  const int l2 = pic.log2Ctu - cs;
  const int x0 = it.x, y0 = it.y, ox = ( x0 >> l2 ) << l2, oy = ( y0 >> l2 ) << l2;
  const int lw = it.lw, lh = it.lh, w = 1 << lw, h = 1 << lh;
  const int rows = h >> IT_LPARTS( it ), yb = IT_PART( it ) * rows, wh = rows << lw;
  const bool mip = !comp && ( it.flags & IT_F_MIP );      // (chroma: the same bit says LMCS chroma residual scaling)
  const int mrl = ( it.flags & IT_F_MIP ) ? 0 : ( it.flags >> 4 ) & 3;
  const bool isp = !comp && ( it.flags & IT_F_ISP ) == IT_F_ISP;
  const int wIntra = isp ? 0 : it.flags >> 6;              // CIIP: weight of the planar intra part, 0 = ordinary intra block
  const uint32_t ispw = isp ? it.tu : 0;
  const int ispDx = ispw & 63, ispDy = ( ispw >> 6 ) & 63, cuW = 1 << ( ( ispw >> 12 ) & 7 ), cuH = 1 << ( ( ispw >> 15 ) & 7 );
  const bool ispVer = ( ispw >> 18 ) & 1;
  const int bdpcm = isp ? 0 : ( it.flags & IT_F_BDPCM_H ) ? 1 : ( it.flags & IT_F_BDPCM_V ) ? 2 : 0;
  const int dirMode = it.mode;
  const bool ibc = dirMode == IT_MODE_IBC, cclm = comp && dirMode >= 67 && dirMode <= 69;
  // reference line lengths; ISP: CU size + partition size along the split, twice the CU size across (IntraPrediction.cpp:1000-1001).
  // f*: the block whose line is fetched from the picture (ISP: the whole CU, initIntraPatternChTypeISP :966-999)
  const int topLen = isp ? ( ispVer ? cuW + w : 2 * cuW ) : 2 * w, leftLen = isp ? ( ispVer ? 2 * cuH : cuH + h ) : 2 * h;
  const int fx0 = x0 - ispDx, fy0 = y0 - ispDy, fTopLen = isp ? 2 * cuW : topLen, fLeftLen = isp ? 2 * cuH : leftLen;
  const int unit = 4 >> cs;
  const int nTL = it.nTL & 1, nA = it.nA, nL = it.nL;
  const bool isDc = !bdpcm && dirMode == 1;
  // reference smoothing (useFilteredIntraRefSamples :1301), mode-specific set-up (xPredIntraAng :592-615)
  bool useFilt = false;
  if( !comp && !mrl && !bdpcm && dirMode != 1 && !isp && dirMode <= 66 && !mip )
  {
    if( dirMode == 0 ) useFilt = w * h > 32;
    else
    {
      const int pm = intra_wide_angle( w, h, dirMode );
      const int diff = min( iabs( pm - 18 ), iabs( pm - 50 ) );
      const int am = pm >= 34 ? pm - 50 : -( pm - 18 );
      useFilt = diff > c_intraFilterThr[( lw + lh ) >> 1] && ( ( c_angTable[iabs( am )] & 0x1F ) == 0 );
    }
  }
  const bool pdpcOK = ( w >= 4 && h >= 4 ) && mrl == 0;
  int predMode = 0, angle = 0, invAngle = 0, absAng = 0; bool isVer = true;
  const bool angular = !bdpcm && dirMode > 1 && dirMode <= 66 && !mip;
  if( angular )
  {
    predMode = isp ? intra_wide_angle( cuW, cuH, dirMode ) : intra_wide_angle( w, h, dirMode );     // ISP: the CU's shape (:502,604)
    isVer = predMode >= 34;
    const int am = isVer ? predMode - 50 : -( predMode - 18 );
    invAngle = c_invAngTable[iabs( am )]; absAng = c_angTable[iabs( am )]; angle = am < 0 ? -absAng : absAng;
  }
  const int bw = isVer ? w : h, bh = isVer ? h : w;       // angular modes predict in the transposed domain for horizontal modes
  bool cubic = false, doAngPdpc = false; int angScale = 0;
  if( angular )
  {
    if( !comp )
    {
      const int diff = min( iabs( predMode - 18 ), iabs( predMode - 50 ) );
      cubic = isp || !( diff > c_intraFilterThr[( ilog2( bw ) + ilog2( bh ) ) >> 1] ) || mrl > 0;
    }
    if( angle > 0 )
    {
      const int sideSize = predMode >= 34 ? h : w;
      angScale = min( 2, ilog2( sideSize ) - ( ilog2( 3 * invAngle - 2 ) - 8 ) );
      doAngPdpc = pdpcOK && angScale >= 0;
    }
  }
  const int pscale = ( lw - 2 + lh - 2 + 2 ) >> 2;
  const int pdpcLev = !angular ? 0 : angle == 0 ? ( pdpcOK ? min( pscale == 0 ? 3 : pscale == 1 ? 6 : pscale == 2 ? 12 : 24, bw ) : 0 ) : doAngPdpc ? min( 3 << angScale, bw ) : 0;
  // group loop: a lane predicts g = min( 4, extent ) neighbouring samples of a row of the prediction block; (xx, yy) are the coordinates
  // there: xx = x, yy = y, for horizontal angular modes transposed (xx = y, yy = x)
  const bool tr = angular && !isVer;
  const int nxl = tr ? ilog2( rows ) : lw;                                  // log2 extent of the item in xx
  const int gl = min( 2, nxl ), lgpr = nxl - gl;                             // log2 group size, log2 groups per row
  const int ngroups = ( tr ? w : rows ) << lgpr;
  const bool vec = !tr && gl == 2 && !( x0 & 3 );                           // 8-byte LDS accesses to the tile row and the residual
  const int tileBase = IT_PAD * IT_TS + ( y0 - oy ) * IT_TSB + ( x0 - ox ) + IT_PADX;       // block origin in the tile (rows inside the CTU)
  // reference fill (xFillReferenceSamples :1072-1250): index <= mrl: base0 + step0 * index; beyond: base1 + min( index - 1 - mrl, limit1 ) [* row stride]
  const int cx = fx0 - ( 1 + mrl ), cy = fy0 - ( 1 + mrl );                  // corner
  const int iCorner = tile_idx( cx - ox, cy - oy ), iPad = tile_idx( cx - ox, fy0 - oy ), iAbove = tile_idx( fx0 - ox, cy - oy );
  const int szL = min( nL * unit, fLeftLen ), szA = min( nA * unit, fTopLen );
  const int tb0 = nL ? ( nTL ? iCorner : iPad ) : iAbove, ts0 = ( nL && nTL ) ? 1 : 0;
  const int tb1 = nA ? iAbove : nTL ? iAbove - 1 : iPad, tl1 = nA ? szA - 1 : 0;
  const int lb0 = tb0, ls0 = ( nL && nTL ) ? IT_TSB : 0;        // (rows cy .. cy + mrl lie inside the CTU when mrl > 0)
  const int lb1 = nL ? iPad : iAbove, ll1 = nL ? szL - 1 : 0;
  const int denom = w == h ? w << 1 : max( w, h );
  uint32_t F = 0;
  if( it.flags & IT_F_RESI ) F |= CF_RESI;
  if( comp && ( it.flags & IT_F_CSCALE ) ) F |= CF_CSON;
  if( isDc ) F |= CF_DC;
  if( useFilt ) F |= CF_FILT;
  if( angular ) F |= CF_ANG;
  if( tr ) F |= CF_TR;
  if( absAng & 0x1F ) F |= CF_FRAC;
  if( cubic ) F |= CF_CUBIC;
  if( vec ) F |= CF_VEC;
  if( nTL | nA | nL ) F |= CF_ANY;
  if( isp ) F |= CF_ISP;
  if( mip ) F |= CF_MIP;
  if( ibc ) F |= CF_IBC;
  if( cclm ) F |= CF_CCLM;
  if( !bdpcm && pdpcOK ) F |= CF_PDPC;
  if( wh <= IT_PART_SAMPLES ) F |= CF_STASHED;
  if( dirMode == 0 ) F |= CF_PLANAR;
  if( angle < 0 ) F |= CF_NEG;
  if( angular && angle == 0 ) F |= CF_ANG0;
  c[C_FLAGS] = F;
  c[C_POS] = (uint32_t) x0 | ( (uint32_t) y0 << 16 );
  c[C_GEO] = (uint32_t) lw | ( lh << 4 ) | ( ilog2( rows ) << 8 ) | ( yb << 12 ) | ( mrl << 19 ) | ( wIntra << 21 ) | ( bdpcm << 23 ) | ( gl << 25 ) | ( lgpr << 27 );
  c[C_TILEBASE] = tileBase;
  c[C_N] = max( max( topLen, leftLen ), max( fTopLen, fLeftLen ) ) + mrl + 1;       // <= 131
  c[C_TB0] = tb0; c[C_TS0] = ts0; c[C_TB1] = tb1; c[C_TL1] = tl1; c[C_LB0] = lb0; c[C_LS0] = ls0; c[C_LB1] = lb1; c[C_LL1] = ll1;
  c[C_DCT] = w >= h ? mrl + w : -1; c[C_DCL] = w <= h ? mrl + h : -1;              // DC: the samples next to the longer side(s)
  c[C_DCDEN] = (uint32_t) ilog2( denom ) | ( ( denom >> 1 ) << 8 );
  c[C_TOPLEN] = topLen; c[C_LEFTLEN] = leftLen;
  c[C_ANGLE] = angle; c[C_INVANGLE] = invAngle; c[C_REFEND] = ( isVer ? topLen : leftLen ) + mrl; c[C_BH] = bh;
  c[C_PDPCLEV] = pdpcLev; c[C_ANGSCALE] = angScale; c[C_PSCALE] = pscale;
  c[C_NGROUPS] = ngroups; c[C_XXB] = tr ? yb : 0; c[C_YYB] = tr ? 0 : yb;
  const int csNv1 = pic.log2Ctu > pic.vpduLog2 ? 1 : 0;
  c[C_CSIDX] = ( ( ( y0 << 1 ) >> pic.vpduLog2 ) & csNv1 ) * 2 + ( ( ( x0 << 1 ) >> pic.vpduLog2 ) & csNv1 );
  c[C_ORIGIN] = (uint32_t) ox | ( (uint32_t) oy << 16 );
}

// FINE (round 5; a picture whose CUs are all intra CUs): the CTU-to-CTU dependency is resolved per BLOCK instead of per unit.  A unit waits for nobody when it
// starts; a block whose reference lines leave the CTU polls the per-cell words of exactly the cells it reads (vvr_intra_cells.inc), fetches those samples from
// the picture into the border of the tile and goes on as before; an extra wavefront of the workgroup (the publisher) follows the blocks in order, stores every
// finished block from the tile to the picture (device scope) and clears its cells.  The CTU wavefront of an I picture - 62 steps of a whole CTU at 4K - becomes
// the dependency chain of the blocks themselves: a CTU starts when the few blocks it reads first are done, not when its neighbours are.  Chroma blocks poll the
// luma they read (CCLM, the chroma scaling factor) the same way.
template<int IT_WAVES, bool FINE>
__global__ __launch_bounds__( IT_NT + ( FINE ? 64 : 0 ) ) void k_intra( IntraPic pic, const IntraItem* __restrict__ items, const uint32_t* __restrict__ ctx /* k_intra_setup */,
                                                  const IntraUnit* __restrict__ units, int numActive,
                                                  int* __restrict__ sync /* [0]: ticket, [1 + unit]: done flags */, LeafMaps M /* FINE: the per-cell words */, int* __restrict__ lsync /* FINE: [1] error word */
#ifdef VVR_INTRA_DEV
                                                  , int dbg, unsigned long long* __restrict__ trace /* developer timeline (VVR_INTRA_TRACE) or nullptr */,
                                                  unsigned long long* __restrict__ btrace /* per block: ready / go / filled / done (shader clock) */
#endif
                                                  )
{
#ifndef VVR_INTRA_DEV
  constexpr int dbg = 0; constexpr unsigned long long* trace = nullptr, * btrace = nullptr;      // (the timing experiments and the in-kernel timeline are developer builds only)
#endif
  __shared__ IntraShared<IT_WAVES> sh;
  constexpr bool WT = true;            // samples that cross workgroups of the launch: written through (sc1), read device scope - no agent-scope fences (round 6)
#define IT_TRACE( K ) if( trace && threadIdx.x == 0 ) trace[(size_t) 8 * tr_ticket + ( K )] = wall_clock64()
  int tr_ticket = 0;
  const int tid = threadIdx.x;
  const int wv = tid >> 6, lane = tid & 63;
  if( tid < 32 ) { sh.angTab[tid] = c_angTable[tid]; sh.invAngTab[tid] = c_invAngTable[tid]; }
  if( tid < 8 ) sh.filtThr[tid] = c_intraFilterThr[tid];
  if( tid < 128 ) sh.cfilt[tid >> 2][tid & 3] = d_chroma_filter[tid >> 2][tid & 3];
  // Persistent workgroups: the launch holds at most as many workgroups as the picture's dependency front can keep busy (launch_intra), and
  // every workgroup takes tickets until none is left.  A picture with a long dependency chain (an intra picture: one CTU wavefront) would
  // otherwise park a workgroup per unit on the device, nearly all of them spinning on their producers while holding 50 KB of LDS each - which
  // starves the kernels of every other picture in flight.  Forward progress is as before: every producer holds a lower ticket than its
  // consumers and tickets are taken in order, so whoever waits, waits for a unit that a running workgroup already holds.
  for( ;; )
  {
  __syncthreads();                    // the previous unit of this workgroup is done with the LDS
  if( tid == 0 ) sh.ticket = atomicAdd( &sync[0], 1 );
  if( tid < IT_WAVES ) sh.prog[tid] = 0;
  __syncthreads();
  const int ticket = sh.ticket;
  if( ticket >= numActive ) return;
  tr_ticket = ticket; IT_TRACE( 0 );
  const IntraUnit* __restrict__ un = &units[ticket];
  const uint32_t ent = un->ent;
  const int comp = ( ent >> 24 ) & 3, ctu = ent & 0xffffff;
  const bool borderOnly = ( ent >> 31 ) != 0;          // whole CTU, every sample intra: the interior is produced here, never read first
  const bool publish = !FINE && ( ( ent >> 30 ) & 1 ) != 0 && !( dbg & 0x100 );     // another unit waits for this one
  const int cxI = ctu % pic.ctusX, cyI = ctu / pic.ctusX;
  const int cs = comp ? 1 : 0;
  const int S = ( 1 << pic.log2Ctu ) >> cs;
  const int ox = cxI * S, oy = cyI * S;
  const int bd = pic.bitDepth;
  const int PW = pic.w[comp], PH = pic.h[comp], pstride = pic.stride[comp];
  pel_t* __restrict__ plane = pic.plane[comp];
  const pel_t* __restrict__ rs = pic.resi[comp];
  const int rstride = pic.rstride[comp];
  const uint32_t i0 = un->i0, i1 = un->i1, iA = un->iA;
  // the unit's first IT_BATCH block records are on their way while the producers' flags are polled (one dword per lane; stored to LDS behind the
  // wait): the reference staging below reads them from LDS instead of waiting for HBM / L2 again; so are the records of the first two blocks
  // of every wavefront
  const int nb0 = (int) min( (uint32_t) IT_BATCH, i1 - i0 );
  const uint32_t itemPre = tid < nb0 * 4 ? reinterpret_cast<const uint32_t*>( items + i0 )[tid] : 0u;
  uint4 recN = make_uint4( 0, 0, 0, 0 ), recNN = make_uint4( 0, 0, 0, 0 );
  if( wv < IT_WAVES && iA + wv < i1 ) recN = intra_load_item( items, iA + wv );
  if( wv < IT_WAVES && iA + wv + IT_WAVES < i1 ) recNN = intra_load_item( items, iA + wv + IT_WAVES );
  uint32_t ctxPre = 0;
  if( wv < IT_WAVES && iA + wv < i1 ) ctxPre = ctx[(size_t) ( iA + wv ) * IT_CTX + lane];
#define TILE( x, y ) sh.tile[tile_idx( ( x ) - ox, ( y ) - oy )]
  // block record q of this unit into scalar registers (uniform per wavefront)
#define IT_FETCH( IT, Q ) { const uint32_t* ip_ = ( Q ) - i0 < (uint32_t) IT_BATCH ? reinterpret_cast<const uint32_t*>( &sh.items[( Q ) - i0] ) : reinterpret_cast<const uint32_t*>( &items[Q] ); \
                            uint32_t* op_ = reinterpret_cast<uint32_t*>( &IT ); for( int e_ = 0; e_ < 4; e_++ ) op_[e_] = __builtin_amdgcn_readfirstlane( ip_[e_] ); }
  // ---- wait for the units that produce intra samples this one reads (same component: reference lines; luma: CCLM)
  {
    // one lane per producer (at most VVR_INTRA_MAX_DEPS of them): the polls overlap instead of queueing behind each other
    const uint32_t nd = ( FINE || ( dbg & 1 ) ) ? 0 : un->ndeps;      // (FINE: every block waits for exactly the cells it reads)
    if( (uint32_t) tid < nd )
    {
      int* flag = &sync[1 + un->deps[tid]];
      while( __hip_atomic_load( flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) == 0 ) __builtin_amdgcn_s_sleep( 8 );
    }
    if( nd )
    {
      __syncthreads();
      // (round 6: no agent-scope acquire here - 1.7 us per unit on the CTU wavefront, and a release fence of 6 - 10 us on the producer's side with a CTU of 32 KB
      // freshly written.  Every sample another workgroup of the launch produces is stored write-through (sc1) and read device scope (sc1 loads: ld_pel / ld_pel8),
      // the flag follows the stores behind s_waitcnt vmcnt(0): MI355X_MICROARCH.md, inter-workgroup visibility, "sc1 stores AND sc1 loads")
      if( !WT ) __builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "agent" );
    }
  }
  if( tid < nb0 * 4 ) reinterpret_cast<uint32_t*>( sh.items )[tid] = itemPre;
  __syncthreads();
  IT_TRACE( 1 );
  // ---- a unit of residual-add blocks only (inter blocks with LMCS chroma scaling): no neighbourhood is read, so the blocks go
  // straight from HBM to HBM, one block per wavefront, without staging the CTU
  if( iA == i1 && i1 > i0 )
  {
    const int csNv1 = pic.log2Ctu > pic.vpduLog2 ? 1 : 0;
    if( ( ent >> 29 ) & 1 )
    {
      const int lx = ( cxI << pic.log2Ctu ) + ( ( wv & csNv1 ) << pic.vpduLog2 ), ly = ( cyI << pic.log2Ctu ) + ( ( ( wv >> 1 ) & csNv1 ) << pic.vpduLog2 );
      if( wv < 4 && lx < pic.width && ly < pic.height && ( wv == 0 || csNv1 ) )
      {
        const int f = lmcs_cscale_factor_wave<WT>( pic, lx, ly, lane );
        if( lane == 0 ) sh.csFac[wv] = f;
      }
    }
    __syncthreads();
    for( uint32_t q = i0 + wv; q < i1; q += IT_WAVES )
    {
      IntraItem it;
      IT_FETCH( it, q )
      const int x0 = it.x, y0 = it.y, lw = it.lw, wh = 1 << ( it.lw + it.lh );
      const bool cs = ( it.flags & IT_F_CSCALE ) != 0;
      const int f = cs ? sh.csFac[( ( ( y0 << 1 ) >> pic.vpduLog2 ) & csNv1 ) * 2 + ( ( ( x0 << 1 ) >> pic.vpduLog2 ) & csNv1 )] : 0;
      if( lw >= 2 && !( x0 & 3 ) )
      {
        // four samples of a row per lane (8-byte accesses; x0 and the row strides are multiples of 4 samples - not so for the chroma of
        // an 8-wide inter CU that is the middle part of a ternary split of 16)
        for( int i = lane; i < ( wh >> 2 ); i += 64 )
        {
          const int x = x0 + ( ( i << 2 ) & ( ( 1 << lw ) - 1 ) ), y = y0 + ( ( i << 2 ) >> lw );
          const uint2 rv = *reinterpret_cast<const uint2*>( &rs[(size_t) y * rstride + x] );
          uint2* pp = reinterpret_cast<uint2*>( &plane[(size_t) y * pstride + x] );
          const uint2 pv = *pp;
          int r[4] = { (int16_t) ( rv.x & 0xffff ), (int16_t) ( rv.x >> 16 ), (int16_t) ( rv.y & 0xffff ), (int16_t) ( rv.y >> 16 ) };
          const int pr[4] = { (int) ( pv.x & 0xffff ), (int) ( pv.x >> 16 ), (int) ( pv.y & 0xffff ), (int) ( pv.y >> 16 ) };
          int o[4];
          for( int e = 0; e < 4; e++ ) o[e] = clip_pel( pr[e] + ( cs ? lmcs_scale_resi( r[e], f, bd ) : r[e] ), bd );
          st_pel4_sc1( &plane[(size_t) y * pstride + x], make_uint2( (uint32_t) o[0] | ( (uint32_t) o[1] << 16 ), (uint32_t) o[2] | ( (uint32_t) o[3] << 16 ) ) );
        }
      }
      else
        for( int i = lane; i < wh; i += 64 )
        {
          const int x = x0 + ( i & ( ( 1 << lw ) - 1 ) ), y = y0 + ( i >> lw );
          const int r = (int16_t) rs[(size_t) y * rstride + x];
          st_pel_sc1( &plane[(size_t) y * pstride + x], clip_pel( plane[(size_t) y * pstride + x] + ( cs ? lmcs_scale_resi( r, f, bd ) : r ), bd ) );
        }
    }
    IT_TRACE( 4 );
    if( trace && threadIdx.x == 0 ) { trace[(size_t) 8 * tr_ticket + 6] = ( (unsigned long long) ( i1 - i0 ) << 32 ) | ent; trace[(size_t) 8 * tr_ticket + 7] = un->ndeps | 0x100; }
    if( !publish ) continue;
    asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
    __syncthreads();
    if( tid == 0 )
    {
      if( !WT ) __builtin_amdgcn_fence( __ATOMIC_RELEASE, "agent" );      // (WT: the samples went out write-through, every wavefront has drained its stores in front of the barrier)
      asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
      __hip_atomic_store( &sync[1 + ticket], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    }
    IT_TRACE( 5 );
    continue;
  }
  const int csNv1 = pic.log2Ctu > pic.vpduLog2 ? 1 : 0;        // VPDUs per CTU side - 1
  // ---- block loop: ONE WAVEFRONT PER BLOCK.  Wavefront w takes the unit's blocks w, w + 4, ... in order.  Everything that does not depend on
  // neighbouring samples - the block's record, its residual, the co-located luma of a CCLM block, the mode set-up - is done before the block's
  // turn comes (records two blocks ahead, residual and luma one block ahead, all in registers); a block starts when the blocks before it (all
  // but the `indep` directly before it, which it does not read) are finished: progress counters in LDS, one per wavefront.  No workgroup
  // barrier on the block-to-block path, and one wait for the LDS per block: inside a wavefront LDS operations execute in order.  What is
  // left on that path is kept short: all reference samples of a block are read from the tile at once (one read per line and lane, the
  // substitution of unavailable samples is index arithmetic), smoothing happens in place, the side reference of a negative prediction
  // angle is projected onto negative indices of the main reference once per block, and a lane predicts four neighbouring samples of a row
  // of the (possibly transposed) block from seven reference samples with one filter.
  // The loop is software-pipelined without a copy of its loads in front of it: its first round has no block, it only starts the fetches of
  // the wavefront's first block - and stages the unit's part of the tile behind them.
  const int w4c = M.w4, h4c = M.h4, cu_ = comp ? 1 : 2;      // (FINE) the cell maps: row stride, rows, log2 samples per cell side of this component
  if( !FINE || wv < IT_WAVES )
  {
    IntraWave& W = sh.wave[wv];
    pel_t* const T = W.topB + IT_NEG;
    pel_t* const L = W.leftB + IT_NEG;
    int done = 0;                     // blocks this wavefront has finished
    IntraResiRegs RR;
    IntraLumaRegs LR;
    for( int e = 0; e < 4; e++ ) RR.v[e] = make_uint2( 0, 0 );
    for( int e = 0; e < IT_CCLM_REGS / 128; e++ ) LR.v[e] = 0;
    LR.tpl = 0;
    const int q0 = (int) iA + wv, qEnd = ( dbg & 4 ) ? (int) i0 : (int) i1;
    uint4 recC = make_uint4( 0, 0, 0, 0 );
    uint32_t ctxCur = 0, ctxN = ctxPre;
#define IT_CSYNC() asm volatile( "" ::: "memory" )      /* LDS accesses of one wavefront execute in order: only the compiler has to keep them in order */
    for( int q = q0 - IT_WAVES; ; q += IT_WAVES )
    {
      const bool cur = q >= q0;
      if( cur && q >= qEnd ) break;
      const IntraItem it = intra_item_of( recC );
      if( cur ) intra_stash_resi( RR, it, W.resi, lane );
      IntraLumaRegs LC = LR;                                                   // (CCLM) co-located luma of this block, fetched while the block before was predicted (FINE: when its turn comes, below)
      // the residual (and the luma of a CCLM block) of the wavefront's next block starts, the records move up
      if( q + IT_WAVES < qEnd ) { const IntraItem itN = intra_item_of( recN ); intra_load_resi( RR, itN, rs, rstride, lane ); if( comp && !FINE ) intra_load_cclm_luma<WT>( LR, itN, pic, lane ); }
      recC = recN; recN = recNN;
      if( q + 3 * IT_WAVES < qEnd ) recNN = intra_load_item( items, (uint32_t) ( q + 3 * IT_WAVES ) );
      const uint32_t ctxC = ctxCur;                                           // the block's parameter record (lane k: value k), fetched two blocks ahead
      ctxCur = ctxN;
      if( q + 2 * IT_WAVES < qEnd ) ctxN = ctx[(size_t) ( q + 2 * IT_WAVES ) * IT_CTX + lane];
      if( !cur )
      {
      // ---- stage the needed part of the CTU and its reference border in LDS
      {
        // 16-byte chunks (8 samples); plane rows are 128-byte aligned and padded to a multiple of 64 samples, so a chunk that
        // straddles the picture's right edge stays inside the row allocation (those samples are never used).
        // bbox (host glue): rows / chunks that hold reference samples of this CTU's blocks, relative to (oy - IT_PAD, ox - IT_PADX)
        const uint32_t bb = un->bbox;
        const int y0 = max( 0, oy - IT_PAD ), y1 = min( PH, oy + S );
        const int c0 = ox >= IT_PADX ? -1 : 0;                                  // first chunk relative to ox / 8
        const int c1 = ( min( PW, ox + S + IT_RIGHT ) - ox + 7 ) >> 3;           // one past the last chunk
        const int by0 = max( y0, oy - IT_PAD + (int) ( bb & 0xff ) ), by1 = min( y1, oy - IT_PAD + (int) ( ( bb >> 8 ) & 0xff ) );
        const int bc0 = max( c0, (int) ( ( bb >> 16 ) & 0xff ) - 1 ), bc1 = min( c1, (int) ( bb >> 24 ) - 1 );
        const int nch = bc1 - bc0;
        const int nTop = nch * max( 0, min( by1, oy ) - by0 );                   // chunks in the rows above the CTU
        const int rowsIn = max( 0, by1 - max( by0, oy ) );
        const bool perBlock = !borderOnly;
        const int total = ( FINE || ( dbg & 2 ) || perBlock ) ? 0 : nTop + ( bc0 < 0 ? rowsIn : 0 );      // (FINE: the blocks fetch what they read of the border themselves)
        if( perBlock && !( dbg & 2 ) )
        {
          // a unit that is not a whole intra CTU: only the reference lines its blocks read (one row above, one column left of every
          // block, as far as they are available) instead of the bounding box; one block per wavefront, 16-byte chunks.  The fetches of up to
          // four blocks of a wavefront are in flight together (a block's lines are 64 chunks at most, one per lane, in all but rare cases):
          // one memory round trip per unit instead of one per block
          auto stageItem = [&]( uint32_t q, uint4& sv, int& so )
          {
            IntraItem it;
            IT_FETCH( it, q )
            if( IT_PART( it ) ) return;                                              // (row parts of one block: fetched with the first)
            if( it.mode == IT_MODE_IBC )
            {
              // intra block copy: the part of the reference block that lies in this CTU is read from the tile (blocks of this unit write
              // their samples there first), so it is staged like a reference line; the part in a CTU further left is read from HBM
              const int bw = 1 << it.lw, bh = 1 << it.lh;
              const int rx = (int) it.x + (int16_t) ( it.tu & 0xffff ), ry = (int) it.y + (int16_t) ( it.tu >> 16 );
              const int cx0 = max( rx, ox ) & ~7, cch = max( 0, ( rx + bw - cx0 + 7 ) >> 3 );
              for( int i = lane; i < cch * bh; i += 64 )
              {
                const int y = ry + i / cch, x = cx0 + 8 * ( i % cch );
                const uint4 v = ld_pel8<WT>( &plane[(size_t) y * pstride + x] );
                *reinterpret_cast<uint4*>( &sh.tile[tile_idx( x - ox, y - oy )] ) = v;
              }
              return;
            }
            const bool isp = !comp && ( it.flags & IT_F_ISP ) == IT_F_ISP;
            if( isp && ( it.tu & 0xfff ) ) return;                                 // later ISP partitions: the CU's line was fetched with the first one
            const int bw = isp ? 1 << ( ( it.tu >> 12 ) & 7 ) : 1 << it.lw, bh = isp ? 1 << ( ( it.tu >> 15 ) & 7 ) : 1 << it.lh;
            const int mrl = ( isp || comp || ( it.flags & IT_F_MIP ) ) ? 0 : ( it.flags >> 4 ) & 3;
            const int unit = 4 >> cs;
            const int lx = (int) it.x - 1 - mrl, ty = (int) it.y - 1 - mrl;
            // row above: from the corner column to the last available sample above / above-right
            const int tx0 = max( 0, lx ) & ~7, tx1 = (int) it.x + max( (int) it.nA * unit, 1 );
            const int nT = ty >= 0 ? ( tx1 - tx0 + 7 ) >> 3 : 0;
            // column left: from the corner row to the last available sample left / below-left
            const int ly0 = max( 0, ty ), ly1 = (int) it.y + (int) it.nL * unit;
            const int nL = lx >= 0 ? max( 0, ly1 - ly0 ) : 0;
            // CIIP: the inter prediction of the block itself
            const int wIntra = isp ? 0 : it.flags >> 6;                             // (the two bits are zero for every other kind of block)
            const int cch = ( ( (int) it.x & 7 ) + bw + 7 ) >> 3, nC = wIntra ? cch * bh : 0;
            auto chunkAt = [&]( int i, int& x, int& y )
            {
              if( i < nT ) { x = tx0 + 8 * i; y = ty; }
              else if( i < nT + nL ) { x = lx & ~7; y = ly0 + ( i - nT ); }
              else { const int j = i - nT - nL; y = (int) it.y + j / cch; x = ( (int) it.x & ~7 ) + 8 * ( j % cch ); }
            };
            if( lane < nT + nL + nC )
            {
              int x, y; chunkAt( lane, x, y );
              sv = ld_pel8<WT>( &plane[(size_t) y * pstride + x] );
              so = tile_idx( x - ox, y - oy );
            }
            for( int i = lane + 64; i < nT + nL + nC; i += 64 )
            {
              int x, y; chunkAt( i, x, y );
              const uint4 v = ld_pel8<WT>( &plane[(size_t) y * pstride + x] );
              *reinterpret_cast<uint4*>( &sh.tile[tile_idx( x - ox, y - oy )] ) = v;
            }
          };
          for( uint32_t qb = i0 + wv; qb < i1; qb += 4 * IT_WAVES )
          {
            uint4 sv[4]; int so[4];
#pragma unroll
            for( int r = 0; r < 4; r++ ) { so[r] = -1; sv[r] = make_uint4( 0, 0, 0, 0 ); const uint32_t q = qb + r * IT_WAVES; if( q < i1 ) stageItem( q, sv[r], so[r] ); }
#pragma unroll
            for( int r = 0; r < 4; r++ ) if( so[r] >= 0 ) *reinterpret_cast<uint4*>( &sh.tile[so[r]] ) = sv[r];
          }
        }
        for( int base = 0; base < total; base += IT_NT * 4 )
        {
          // four 16-byte loads in flight per lane; the tail repeats the last chunk (same data to the same place) instead of branching
          uint4 v0, v1, v2, v3; int o0, o1, o2, o3;
#define IT_LD( V, O, U ) { const int i = min( base + U * IT_NT + tid, total - 1 ); int r, cidx; \
            if( i < nTop ) { r = i / nch; cidx = bc0 + ( i - r * nch ); } else { r = ( max( by0, oy ) - by0 ) + ( i - nTop ); cidx = -1; } \
            const int y = by0 + r, x = ox + cidx * 8; V = ld_pel8<WT>( &plane[(size_t) y * pstride + x] ); O = tile_idx( x - ox, y - oy ); }
          IT_LD( v0, o0, 0 ) IT_LD( v1, o1, 1 ) IT_LD( v2, o2, 2 ) IT_LD( v3, o3, 3 )
#undef IT_LD
          *reinterpret_cast<uint4*>( &sh.tile[o0] ) = v0; *reinterpret_cast<uint4*>( &sh.tile[o1] ) = v1;
          *reinterpret_cast<uint4*>( &sh.tile[o2] ) = v2; *reinterpret_cast<uint4*>( &sh.tile[o3] ) = v3;
        }
      }
    IT_TRACE( 2 );
    // ---- LMCS chroma residual scaling (DecCu.cpp:383-388,500-505): one factor per VPDU of the CTU, one wavefront each (the unit
    // has waited for the luma units that reconstruct the samples the factors are averaged over)
    if( ( ent >> 29 ) & 1 )
    {
      const int lx = ( cxI << pic.log2Ctu ) + ( ( wv & csNv1 ) << pic.vpduLog2 ), ly = ( cyI << pic.log2Ctu ) + ( ( ( wv >> 1 ) & csNv1 ) << pic.vpduLog2 );
      if( wv < 4 && lx < pic.width && ly < pic.height && ( wv == 0 || csNv1 ) )
      {
        if( FINE )
        {
          // the luma the factor is averaged over: left of / above the CU at the VPDU's origin (calculateChromaAdjVpduNei)
          const uint32_t d = pic.csVpdu[( ly >> pic.vpduLog2 ) * pic.vpdusX + ( lx >> pic.vpduLog2 )];
          const int xPos = d & 0x1fff, yPos = ( d >> 13 ) & 0x1fff, n = 1 << pic.vpduLog2;
          LeafSeg seg[2];
          seg[0] = ( ( d >> 26 ) & 1 ) ? leaf_seg( M.cell[0], w4c, h4c, ( xPos - 1 ) >> 2, yPos >> 2, ( xPos - 1 ) >> 2, min( yPos + n - 1, pic.height - 1 ) >> 2 ) : leaf_seg( M.cell[0], w4c, h4c, 0, 0, -1, -1 );
          seg[1] = ( ( d >> 27 ) & 1 ) ? leaf_seg( M.cell[0], w4c, h4c, xPos >> 2, ( yPos - 1 ) >> 2, min( xPos + n - 1, pic.width - 1 ) >> 2, ( yPos - 1 ) >> 2 ) : leaf_seg( M.cell[0], w4c, h4c, 0, 0, -1, -1 );
          leaf_wait( seg, w4c, lane, lsync );
        }
        const int f = lmcs_cscale_factor_wave<FINE || WT>( pic, lx, ly, lane );
        if( lane == 0 ) sh.csFac[wv] = f;
      }
    }
    lds_barrier();
        continue;
      }
      // ---- the block's parameters come from its record (k_intra_setup), a value at a time where it is needed
#define CP( K ) ( (int) __builtin_amdgcn_readlane( ctxC, K ) )
      const uint32_t F = (uint32_t) CP( C_FLAGS ), G = (uint32_t) CP( C_GEO );
      const int lw = CG_LW( G ), lh = CG_LH( G ), w = 1 << lw, h = 1 << lh;
      const int yb = CG_YB( G );                                               // first row of the band of rows this item predicts
      const int wh = 1 << ( lw + CG_LROWS( G ) );                              // samples of this item
      const int mrl = CG_MRL( G );
      const bool hasResi = ( F & CF_RESI ) != 0, csOn = ( F & CF_CSON ) != 0, stashed = ( F & CF_STASHED ) != 0;
      const int tileBase = CP( C_TILEBASE );                                   // block origin in the tile (rows inside the CTU)
      // LMCS chroma residual scaling factor of the block's VPDU (the unit has waited for the luma it is averaged over)
      int csScale = 0;
      if( csOn ) csScale = sh.csFac[CP( C_CSIDX )];
#define IT_BT( K ) if( btrace && lane == 0 ) btrace[(size_t) 8 * q + ( K )] = clock64()
      IT_BT( 0 );
      // ---- the block's turn: every block of the unit it may read from is finished
      {
        const int m = q - (int) iA - IT_INDEP( it );              // blocks 0 .. m - 1 of the unit
        if( m > 0 )
        {
          const int need = max( 0, ( m - ( lane & ( IT_WAVES - 1 ) ) + IT_WAVES - 1 ) / IT_WAVES );
          for( ;; )
          {
            const int p = __hip_atomic_load( &sh.prog[lane & ( IT_WAVES - 1 )], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP );      // (LDS-typed: a generic pointer made this a flat load that waited for every outstanding fetch)
            if( !__builtin_amdgcn_ballot_w64( p < need ) ) break;
            __builtin_amdgcn_s_sleep( 1 );
          }
        }
        IT_CSYNC();
      }
      IT_BT( 1 );
      // (the wait makes the block's samples visible to the other wavefronts before the counter moves)
#define IT_DONE() { wave_lds_sync(); done++; if( lane == 0 ) __hip_atomic_store( &sh.prog[wv], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP ); IT_BT( 3 ); }
      // residual of sample i of the item (row-major inside its band): from the scratch, or from the residual plane for a block that is not split
      auto RES = [&]( int i ) -> int { const int pos = CP( C_POS ); return stashed ? (int) W.resi[i] : (int) (int16_t) rs[(size_t) ( ( pos >> 16 ) + yb + ( i >> lw ) ) * rstride + ( pos & 0xffff ) + ( i & ( w - 1 ) )]; };
      // ---- intra block copy (InterPrediction::xIntraBlockCopy :1995, DecCu.cpp:442-470): copy of reconstructed samples of this picture
      // at the block vector (+ residual).  Samples of this CTU come from the tile, where the blocks of this unit have put theirs and
      // the others were staged after the wait for their producers; samples of a CTU further left come from HBM.
      if( F & CF_IBC )
      {
        const int x0 = it.x, y0 = it.y;
        const int qx = x0 + (int16_t) ( it.tu & 0xffff ), qy = y0 + (int16_t) ( it.tu >> 16 );
#pragma unroll 1
        for( int i = lane; i < wh; i += 64 )
        {
          const int x = i & ( w - 1 ), y = yb + ( i >> lw );
          const int sx = qx + x, sy = qy + y;
          int v = sx >= ox ? (int) TILE( sx, sy ) : ld_pel<WT>( &plane[(size_t) sy * pstride + sx] );
          if( hasResi ) { const int r = RES( i ); v = clip_pel( v + ( csOn ? lmcs_scale_resi( r, csScale, bd ) : r ), bd ); }
          sh.tile[tileBase + y * IT_TSB + x] = (pel_t) v;
        }
        IT_DONE()
        continue;
      }
      if( FINE )
      {
        // ---- what the block reads outside its CTU: wait for exactly those cells, then bring the samples into the border of the tile (the fill below reads them
        // there as ever).  A block at the CTU's top edge reads its whole top line (corner, above, above-right) outside, one at the left edge its left line.
        const bool ispB = ( F & CF_ISP ) != 0;
        if( ( F & CF_ANY ) && !( ispB && ( it.tu & 0xfff ) ) )      // (later ISP partitions: the coding unit's lines are in the tile since the first one)
        {
          const uint32_t ispw = ispB ? it.tu : 0;
          const int bx0 = it.x, by0 = it.y;
          const int fx0 = bx0 - (int) ( ispw & 63 ), fy0 = by0 - (int) ( ( ispw >> 6 ) & 63 );
          const int fTopLen = ispB ? 2 << ( ( ispw >> 12 ) & 7 ) : 2 * w, fLeftLen = ispB ? 2 << ( ( ispw >> 15 ) & 7 ) : 2 * h;
          const int unitS = 4 >> cs, nTL = it.nTL & 1, nA = it.nA, nL = it.nL;
          const int szA = min( nA * unitS, fTopLen ), szL = min( nL * unitS, fLeftLen );
          const int cxx = fx0 - 1 - mrl, cyy = fy0 - 1 - mrl;
          const bool topOut = fy0 == oy && ( nA || nTL ), leftOut = fx0 == ox && ( nL || nTL );
          // top rectangle: rows cyy .. oy - 1, columns tc0 .. tc1; left rectangle: columns cxx .. ox - 1, rows lr0 .. lr1
          const int tc0 = nTL ? cxx : fx0, tc1 = nA ? fx0 + szA - 1 : fx0 - 1, tn = topOut ? max( 0, tc1 - tc0 + 1 ) : 0;
          const int lr0 = nTL ? cyy : fy0, lr1 = nL ? fy0 + szL - 1 : fy0 - 1, ln = leftOut ? max( 0, lr1 - lr0 + 1 ) : 0;
          if( tn | ln )
          {
            LeafSeg seg[2];
            seg[0] = tn ? leaf_seg( M.cell[comp], w4c, h4c, tc0 >> cu_, cyy >> cu_, tc1 >> cu_, ( oy - 1 ) >> cu_ ) : leaf_seg( M.cell[comp], w4c, h4c, 0, 0, -1, -1 );
            seg[1] = ln ? leaf_seg( M.cell[comp], w4c, h4c, cxx >> cu_, lr0 >> cu_, ( ox - 1 ) >> cu_, lr1 >> cu_ ) : leaf_seg( M.cell[comp], w4c, h4c, 0, 0, -1, -1 );
            leaf_wait( seg, w4c, lane, lsync );
            const int rowsT = mrl + 1, total = rowsT * ( tn + ln );
#pragma unroll 1
            for( int k = lane; k < total; k += 64 )
            {
              int x, y;
              if( k < rowsT * tn ) { const int r = k / tn; x = tc0 + ( k - r * tn ); y = cyy + r; }
              else { const int k2 = k - rowsT * tn, c = k2 / ln; x = cxx + c; y = lr0 + ( k2 - c * ln ); }
              sh.tile[tile_idx( x - ox, y - oy )] = (pel_t) ld_pel<true>( &plane[(size_t) y * pstride + x] );
            }
            IT_CSYNC();
          }
        }
        if( F & CF_CCLM )
        {
          // the co-located luma and its template rows / columns (luma blocks of this and of the neighbouring CTUs: all in units with lower tickets)
          const uint32_t lm = it.tu;
          const int actualTop = lm & 0xff, actualLeft = ( lm >> 8 ) & 0xff;
          const int lx0 = (int) it.x << 1, ly0 = (int) it.y << 1;
          LeafSeg seg[3];
          seg[0] = leaf_seg( M.cell[0], w4c, h4c, lx0 >> 2, ly0 >> 2, ( lx0 + 2 * w - 1 ) >> 2, ( ly0 + 2 * h - 1 ) >> 2 );
          seg[1] = ly0 > 0 ? leaf_seg( M.cell[0], w4c, h4c, ( lx0 - 4 ) >> 2, ( ly0 - 4 ) >> 2, ( lx0 + 2 * max( w, actualTop ) - 1 ) >> 2, ( ly0 - 4 ) >> 2 ) : leaf_seg( M.cell[0], w4c, h4c, 0, 0, -1, -1 );
          seg[2] = lx0 > 0 ? leaf_seg( M.cell[0], w4c, h4c, ( lx0 - 4 ) >> 2, ly0 >> 2, ( lx0 - 4 ) >> 2, ( ly0 + 2 * max( h, actualLeft ) - 1 ) >> 2 ) : leaf_seg( M.cell[0], w4c, h4c, 0, 0, -1, -1 );
          leaf_wait( seg, w4c, lane, lsync );
          intra_load_cclm_luma<true>( LC, it, pic, lane );
        }
      }
      // ---- xFillReferenceSamples (:1072-1250): one lane per reference position, ONE read per line and lane - the substitution of samples that
      // are not available (the nearest available one along the line, the first left sample for an unavailable corner, mid grey if nothing is)
      // only moves the position that is read: index <= mrl (the corner and the extension of its row / column): base0 + step0 * index; beyond:
      // base1 + min( index - 1 - mrl, limit1 ) [* row stride], all from the record.  Lines: index 0 = the corner, then the samples above (left
      // of) the block.  ISP: the line of the whole CU first, the partition's own line is cut out of it below.
      int dcPart = 0;
      {
        const bool isp = ( F & CF_ISP ) != 0, isDc = ( F & CF_DC ) != 0;
        const int n = CP( C_N );
        pel_t* const dT = isp ? W.auxT : T;
        pel_t* const dL = isp ? W.auxL : L;
        const int dcT = CP( C_DCT ), dcL = CP( C_DCL );
        if( F & CF_ANY )
        {
          const int tb0 = CP( C_TB0 ), ts0 = CP( C_TS0 ), tb1 = CP( C_TB1 ), tl1 = CP( C_TL1 ), lb0 = CP( C_LB0 ), ls0 = CP( C_LS0 ), lb1 = CP( C_LB1 ), ll1 = CP( C_LL1 );
#pragma unroll 1
          for( int j = lane; j < n; j += 64 )
          {
            const int i = j - 1 - mrl;
            const bool c = j <= mrl;
            const int tv = sh.tile[c ? tb0 + ts0 * j : tb1 + min( i, tl1 )];
            const int lv = sh.tile[c ? lb0 + ls0 * j : lb1 + min( i, ll1 ) * IT_TSB];
            dT[j] = (pel_t) tv; dL[j] = (pel_t) lv;          // (entries past a line's end are written but never used; n <= 131 fits both arrays)
            if( isDc ) dcPart += ( ( j > mrl && j <= dcT ) ? tv : 0 ) + ( ( j > mrl && j <= dcL ) ? lv : 0 );
          }
        }
        else
        {
          const int dcv = 1 << ( bd - 1 );
          for( int j = lane; j < n; j += 64 ) { dT[j] = (pel_t) dcv; dL[j] = (pel_t) dcv; if( isDc ) dcPart += ( ( j > mrl && j <= dcT ) ? dcv : 0 ) + ( ( j > mrl && j <= dcL ) ? dcv : 0 ); }
        }
        if( isp )
        {
          // intra sub-partition (luma): the partition's line is cut out of the line of the whole CU
          const uint32_t ispw = it.tu;
          const int x0 = it.x, y0 = it.y;
          const int ispDx = ispw & 63, ispDy = ( ispw >> 6 ) & 63, cuW = 1 << ( ( ispw >> 12 ) & 7 ), cuH = 1 << ( ( ispw >> 15 ) & 7 );
          const bool ispVer = ( ispw >> 18 ) & 1;
          const int fTopLen = 2 * cuW, fLeftLen = 2 * cuH, topLen = CP( C_TOPLEN ), leftLen = CP( C_LEFTLEN );
          const int nA = it.nA, nL = it.nL;
          IT_CSYNC();
          dcPart = 0;
#pragma unroll 1
          for( int j = lane; j < n; j += 64 )
          {
            int tv, lv;
            // later partitions: the row above / column left comes from the reconstruction of the previous partition (padded with its
            // last sample), the other line continues the CU's line (:1003-1069)
            if( !ispDx && !ispDy ) { tv = W.auxT[min( j, fTopLen )]; lv = W.auxL[min( j, fLeftLen )]; }
            else if( !ispVer )
            {
              lv = nL ? W.auxL[min( ispDy + j, fLeftLen )] : TILE( x0, y0 - 1 );
              tv = j == 0 ? lv : TILE( x0 + min( j - 1, w - 1 ), y0 - 1 );
            }
            else
            {
              tv = nA ? W.auxT[min( ispDx + j, fTopLen )] : TILE( x0 - 1, y0 );
              lv = j == 0 ? tv : TILE( x0 - 1, y0 + min( j - 1, h - 1 ) );
            }
            if( j <= topLen + mrl ) T[j] = (pel_t) tv;
            if( j <= leftLen + mrl ) L[j] = (pel_t) lv;
            if( isDc ) dcPart += ( ( j > mrl && j <= dcT ) ? tv : 0 ) + ( ( j > mrl && j <= dcL ) ? lv : 0 );
          }
        }
      }
      int dcVal = 0;
      if( F & CF_DC ) { const int dd = CP( C_DCDEN ); dcVal = ( wave_sum( dcPart ) + ( dd >> 8 ) ) >> ( dd & 0xff ); }
      IT_CSYNC();
      IT_BT( 2 );
      // ---- MIP (PredictorMIP, MatrixIntraPrediction.cpp:68-330): boundary down-sampling, matrix-vector product, up-sampling
      if( F & CF_MIP )
      {
        const int dirMode = it.mode;
        const bool transp = ( it.flags & 0x10 ) != 0;
        const int sizeId = ( w == 4 && h == 4 ) ? 0 : ( w == 4 || h == 4 || ( w == 8 && h == 8 ) ) ? 1 : 2;
        const int bdry = sizeId == 0 ? 2 : 4, red = sizeId < 2 ? 4 : 8, l2red = sizeId < 2 ? 2 : 3;
        const int upH = w / red, upV = h / red, l2H = ilog2( upH ), l2V = ilog2( upV );
        if( lane < 2 * bdry )
        {
          const bool isL = lane >= bdry; const int qq = isL ? lane - bdry : lane;
          const pel_t* src = isL ? L : T; const int len = isL ? h : w;
          const int f = len / bdry;
          int sum = 0;
          for( int t2 = 0; t2 < f; t2++ ) sum += src[1 + qq * f + t2];
          W.lmSel[lane] = f > 1 ? ( sum + ( f >> 1 ) ) >> ilog2( f ) : sum;
        }
        IT_CSYNC();
        {
          const int inSize = 2 * bdry;
          int in[8];
          for( int qq = 0; qq < 8; qq++ ) in[qq] = qq < inSize ? ( transp ? ( qq < bdry ? W.lmSel[bdry + qq] : W.lmSel[qq - bdry] ) : W.lmSel[qq] ) : 0;
          const int inOff = in[0];
          in[0] = sizeId < 2 ? ( 1 << ( bd - 1 ) ) - inOff : 0;
          int sum = in[0];
          for( int qq = 1; qq < 8; qq++ ) if( qq < inSize ) { in[qq] = (int16_t) ( in[qq] - inOff ); sum += in[qq]; }
          const int offset = 32 - 32 * sum;
          const int redSize = sizeId == 2;
          if( lane < red * red )
          {
            const uint8_t* wt = ( sizeId == 0 ? &d_mip_matrix_4x4[dirMode][0][0] : sizeId == 1 ? &d_mip_matrix_8x8[dirMode][0][0] : &d_mip_matrix_16x16[dirMode][0][0] ) + lane * ( inSize - redSize );
            int acc = redSize ? 0 : in[0] * wt[0];
            for( int qq = 1; qq < 8; qq++ ) if( qq < inSize ) acc += in[qq] * wt[qq - redSize];
            const int v = clip_pel( ( ( acc + offset ) >> 6 ) + inOff, bd );
            const int py = lane >> l2red, px = lane & ( red - 1 );
            W.auxT[transp ? px * red + py : lane] = (pel_t) v;
          }
        }
        IT_CSYNC();
        // horizontal up-sampling into every upV-th row of the block (predictionUpsampling1D :196)
#pragma unroll 1
        for( int i = lane; i < red * w; i += 64 )
        {
          const int kk = i >> lw, x = i & ( w - 1 ), row = ( upV - 1 ) + kk * upV;
          int v;
          if( upH == 1 ) v = W.auxT[kk * red + x];
          else
          {
            const int j = x >> l2H, ii = ( x & ( upH - 1 ) ) + 1;
            const int before = j == 0 ? L[1 + row] : W.auxT[kk * red + j - 1], behind = W.auxT[kk * red + j];
            v = (int16_t) ( before * upH + ( upH >> 1 ) + ii * (int16_t) ( behind - before ) ) >> l2H;
          }
          sh.tile[tileBase + row * IT_TSB + x] = (pel_t) v;
        }
        IT_CSYNC();
        // vertical up-sampling (the rows that hold the horizontally up-sampled lines keep their values; the others only read those rows, so
        // writing in place is safe) and the residual
#pragma unroll 1
        for( int i = lane; i < w * h; i += 64 )
        {
          const int x = i & ( w - 1 ), y = i >> lw;
          const int j = y >> l2V, ii = ( y & ( upV - 1 ) ) + 1;
          const int behind = sh.tile[tileBase + ( ( upV - 1 ) + j * upV ) * IT_TSB + x];
          int v = behind;
          if( ii != upV )
          {
            const int before = j == 0 ? T[1 + x] : sh.tile[tileBase + ( ( upV - 1 ) + ( j - 1 ) * upV ) * IT_TSB + x];
            v = (int16_t) ( before * upV + ( upV >> 1 ) + ii * (int16_t) ( behind - before ) ) >> l2V;
          }
          if( !hasResi ) { if( ii != upV ) sh.tile[tileBase + y * IT_TSB + x] = (pel_t) v; }
          else if( ii != upV ) sh.tile[tileBase + y * IT_TSB + x] = (pel_t) clip_pel( v + RES( i ), bd );
        }
        if( hasResi )
        {
          // (the kept rows last: the rows between them were interpolated from their prediction values)
          IT_CSYNC();
#pragma unroll 1
          for( int i = lane; i < red * w; i += 64 )
          {
            const int kk = i >> lw, x = i & ( w - 1 ), y = ( upV - 1 ) + kk * upV;
            sh.tile[tileBase + y * IT_TSB + x] = (pel_t) clip_pel( sh.tile[tileBase + y * IT_TSB + x] + RES( ( y << lw ) + x ), bd );
          }
        }
        IT_DONE()
        continue;
      }
      // ---- CCLM / MDLM (xGetLumaRecPixels :1403, xGetLMParameters :1694, predIntraChromaLM :519; 4:2:0): the down-sampled luma of the block
      // (up to 256 samples) and of the template positions is in registers already (intra_load_cclm_luma)
      if( F & CF_CCLM )
      {
        const uint32_t lm = it.tu;
        const int actualTop = lm & 0xff, actualLeft = ( lm >> 8 ) & 0xff;
        const bool aboveAvail = ( lm >> 16 ) & 1, leftAvail = ( lm >> 17 ) & 1;
        const int aboveIs4 = leftAvail ? 0 : 1, leftIs4 = aboveAvail ? 0 : 1;
        const int cntT = aboveAvail ? min( actualTop, ( 1 + aboveIs4 ) << 1 ) : 0, cntL = leftAvail ? min( actualLeft, ( 1 + leftIs4 ) << 1 ) : 0;
        if( lane < cntT + cntL )
        {
          int cv;
          if( lane < cntT ) cv = T[1 + ( actualTop >> ( 2 + aboveIs4 ) ) + lane * max( 1, actualTop >> ( 1 + aboveIs4 ) )];
          else              cv = L[1 + ( actualLeft >> ( 2 + leftIs4 ) ) + ( lane - cntT ) * max( 1, actualLeft >> ( 1 + leftIs4 ) )];
          W.lmSel[lane] = (int16_t) LC.tpl; W.lmSel[4 + lane] = cv;
        }
        IT_CSYNC();
        int selL[4] = { 0, 0, 0, 0 }, selC[4] = { 0, 0, 0, 0 };
        const int cnt = cntT + cntL;
        for( int qq = 0; qq < 4; qq++ ) if( qq < cnt ) { selL[qq] = W.lmSel[qq]; selC[qq] = W.lmSel[4 + qq]; }
        if( cnt == 2 )
        {
          selL[3] = selL[0]; selC[3] = selC[0]; selL[2] = selL[1]; selC[2] = selC[1];
          selL[0] = selL[1]; selC[0] = selC[1]; selL[1] = selL[3]; selC[1] = selC[3];
        }
        int mn0 = 0, mn1 = 2, mx0 = 1, mx1 = 3;
        if( selL[mn0] > selL[mn1] ) { const int t = mn0; mn0 = mn1; mn1 = t; }
        if( selL[mx0] > selL[mx1] ) { const int t = mx0; mx0 = mx1; mx1 = t; }
        if( selL[mn0] > selL[mx1] ) { int t = mn0; mn0 = mx0; mx0 = t; t = mn1; mn1 = mx1; mx1 = t; }
        if( selL[mn1] > selL[mx0] ) { const int t = mn1; mn1 = mx0; mx0 = t; }
        const int minL = ( selL[mn0] + selL[mn1] + 1 ) >> 1, minC = ( selC[mn0] + selC[mn1] + 1 ) >> 1;
        const int maxL = ( selL[mx0] + selL[mx1] + 1 ) >> 1, maxC = ( selC[mx0] + selC[mx1] + 1 ) >> 1;
        int a = 0, b = 1 << ( bd - 1 ), shift = 0;
        if( leftAvail || aboveAvail )
        {
          const int diff = maxL - minL;
          b = minC;
          if( diff > 0 )
          {
            const int diffC = maxC - minC;
            int x = 31 - __clz( diff );
            const int normDiff = ( diff << 4 >> x ) & 15;
            const int v = (int) ( ( 0x0111122334455670ull >> ( 4 * normDiff ) ) & 15 ) | 8;      // DivSigTable { 0,7,6,5,5,4,4,3,3,2,2,1,1,1,1,0 }
            x += normDiff != 0;
            const int y = diffC == 0 ? 0 : 32 - __clz( iabs( diffC ) );
            const int add = 1 << y >> 1;
            a = ( diffC * v + add ) >> y;
            shift = 3 + x - y;
            if( shift < 1 ) { shift = 1; a = a == 0 ? 0 : a < 0 ? -15 : 15; }
            b = minC - ( ( a * minL ) >> shift );
          }
        }
        if( wh <= IT_CCLM_REGS )
        {
#pragma unroll
          for( int e = 0; e < IT_CCLM_REGS / 64; e++ )
          {
            const int i = e * 64 + lane;
            if( e * 64 < wh && i < wh )
            {
              const int t = (int16_t) ( ( e & 1 ) ? LC.v[e >> 1] >> 16 : LC.v[e >> 1] & 0xffff );
              int v = clip_pel( ( ( a * t ) >> shift ) + b, bd );
              if( hasResi ) { const int r = W.resi[i]; v = clip_pel( v + ( csOn ? lmcs_scale_resi( r, csScale, bd ) : r ), bd ); }
              sh.tile[tileBase + ( i >> lw ) * IT_TSB + ( i & ( w - 1 ) )] = (pel_t) v;
            }
          }
        }
        else
        {
          // a large block (512 / 1024 samples): the luma is fetched here
          const bool bLeft = ( lm >> 18 ) & 1, bAbove = ( lm >> 20 ) & 1, colloc = pic.colloc != 0;
          const int x0 = it.x, y0 = it.y;
#pragma unroll 2
          for( int i = lane; i < wh; i += 64 )
          {
            const int x = i & ( w - 1 ), y = i >> lw;
            const int t = (int16_t) intra_cclm_luma_at<FINE || WT>( pic.plane[0], pic.stride[0], x0 << 1, y0 << 1, x, y, bLeft, bAbove, colloc );
            int v = clip_pel( ( ( a * t ) >> shift ) + b, bd );
            if( hasResi ) { const int r = W.resi[i]; v = clip_pel( v + ( csOn ? lmcs_scale_resi( r, csScale, bd ) : r ), bd ); }
            sh.tile[tileBase + y * IT_TSB + x] = (pel_t) v;
          }
        }
        IT_DONE()
        continue;
      }
      // ---- reference smoothing (xFilterReferenceSamples :1251): into the second pair of lines
      pel_t* Tp = T; pel_t* Lp = L;
      if( F & CF_FILT )
      {
        pel_t* const fT = W.auxT + IT_NEG; pel_t* const fL = W.auxL + IT_NEG;
        const int topLen = CP( C_TOPLEN ), leftLen = CP( C_LEFTLEN );
        const int mx = max( topLen, leftLen );
#pragma unroll 1
        for( int j = lane; j <= mx; j += 64 )
        {
          const int jm = max( j - 1, 0 );
          const int t1 = T[j + 1], t0_ = T[j], tm = j ? (int) T[jm] : (int) L[1];
          const int l1 = L[j + 1], l0_ = L[j], lm_ = j ? (int) L[jm] : (int) T[1];
          fT[j] = (pel_t) ( j < topLen ? ( t1 + 2 * t0_ + tm + 2 ) >> 2 : t0_ );          // (the last sample of a line stays as it is; index 0 is the same corner value in both)
          fL[j] = (pel_t) ( j < leftLen ? ( l1 + 2 * l0_ + lm_ + 2 ) >> 2 : l0_ );
        }
        Tp = fT; Lp = fL;
        IT_CSYNC();
      }
      // the main / side reference of xPredIntraAng (:640-690): index j is relative to the block (after the multi-reference-line offset);
      // negative indices are the side reference projected with invAngle - written in front of the main reference once per block -,
      // indices beyond the end replicate the last sample
      const bool tr = ( F & CF_TR ) != 0;
      pel_t* const Mn = tr ? Lp : Tp;
      const pel_t* const Sd = tr ? Tp : Lp;
      if( F & CF_NEG )
      {
        const int k = 1 + lane, bh = CP( C_BH ), invAngle = CP( C_INVANGLE );   // (the lowest index read is ( angle * ( mrl + bh ) >> 5 ) + mrl >= -bh >= -64: one round)
        if( k <= bh ) Mn[-k] = Sd[min( ( k * invAngle + 256 ) >> 9, bh )];
        IT_CSYNC();
      }
      {
        // ---- prediction, one group of samples per lane and round.  Planar (:154), DC (:541), BDPCM (:850) with their position-dependent
        // combination (IntraPredSampleFilterCore :212); angular (xPredIntraAng :592) in one 4-tap form for every kind - c = the cubic / Gauss
        // filter of the row's fraction (luma), { 0, 64 - 2 f, 2 f, 0 } for the 2-tap chroma interpolation, { 0, 64, 0, 0 } for whole-sample
        // angles - over 7 neighbouring reference samples
        IT_BT( 4 );
        const bool angular = ( F & CF_ANG ) != 0, vec = ( F & CF_VEC ) != 0;
        const int ngroups = CP( C_NGROUPS ), lgpr = CG_LGPR( G ), gl = CG_GL( G ), g = 1 << gl;
        const int xxb = CP( C_XXB ), yyb = CP( C_YYB ), pscale = CP( C_PSCALE );
        const int maxv = ( 1 << bd ) - 1;
        const int wIntra = CG_WINTRA( G );
        // ISP: partitions narrower than 4 are predicted in groups of width 4 (1 = two 2-wide, 2 = four 1-wide); residual flags of the group's partitions
        const int ispGrp = ( F & CF_ISP ) ? ( it.tu >> 23 ) & 3 : 0, ispResi = ( F & CF_ISP ) ? ( it.tu >> 19 ) & 15 : 0;
        // (uniform values of the two kinds of prediction, fetched before the loop)
        int tR = 0, lB = 0, t0 = 0, angle = 0, refEnd = 0, pdpcLev = 0, angScale = 0, invAngle = 0, bdpcm = 0;
        if( angular ) { angle = CP( C_ANGLE ); refEnd = CP( C_REFEND ); pdpcLev = CP( C_PDPCLEV ); angScale = CP( C_ANGSCALE ); invAngle = CP( C_INVANGLE ); t0 = Tp[0]; }
        else { tR = Tp[w + 1]; lB = Lp[h + 1]; bdpcm = CG_BDPCM( G ); }
        IT_BT( 5 );
#pragma unroll 1
        for( int gi = lane; gi < ngroups; gi += 64 )
        {
          const int yy = yyb + ( gi >> lgpr ), xx0 = xxb + ( ( gi & ( ( 1 << lgpr ) - 1 ) ) << gl );
          int v[4];
          if( !angular )
          {
            const int lft = Lp[yy + 1];
            int tp[4];
            for( int e = 0; e < 4; e++ ) tp[e] = Tp[xx0 + e + 1];
            const int wTp = 32 >> min( 31, ( yy << 1 ) >> pscale );
            for( int e = 0; e < 4; e++ )
            {
              const int x = xx0 + e;
              if( bdpcm ) v[e] = bdpcm == 1 ? lft : tp[e];
              else if( F & CF_PLANAR )
              {
                const int hor = ( lft << lw ) + __mul24( x + 1, tR - lft );
                const int ver = ( tp[e] << lh ) + __mul24( yy + 1, lB - tp[e] );
                v[e] = (int16_t) ( ( ( hor << lh ) + ( ver << lw ) + ( 1 << ( lw + lh ) ) ) >> ( 1 + lw + lh ) );
              }
              else v[e] = dcVal;
              if( F & CF_PDPC )
              {
                const int wLp = 32 >> min( 31, ( x << 1 ) >> pscale );
                v[e] = (int16_t) ( v[e] + ( ( __mul24( wLp, lft - v[e] ) + __mul24( wTp, tp[e] - v[e] ) + 32 ) >> 6 ) );
              }
            }
          }
          else
          {
            const int deltaPos = __mul24( angle, 1 + mrl + yy );
            const int di = deltaPos >> 5, df = deltaPos & 31;
            const int kb = di + xx0 + mrl;
            int r[7];
            for( int t = 0; t < 7; t++ ) r[t] = Mn[min( kb + t, refEnd )];
            int c0 = 0, c1 = 64, c2 = 0, c3 = 0;
            if( F & CF_FRAC )
            {
              if( comp ) { c1 = 64 - 2 * df; c2 = 2 * df; }
              else if( F & CF_CUBIC ) { const uint2 cv = *reinterpret_cast<const uint2*>( sh.cfilt[df] ); c0 = (int16_t) ( cv.x & 0xffff ); c1 = (int16_t) ( cv.x >> 16 ); c2 = (int16_t) ( cv.y & 0xffff ); c3 = (int16_t) ( cv.y >> 16 ); }
              else { c0 = 16 - ( df >> 1 ); c1 = 32 - ( df >> 1 ); c2 = 16 + ( df >> 1 ); c3 = df >> 1; }     // g_intraGaussFilter (:96)
            }
            for( int e = 0; e < 4; e++ ) v[e] = min( max( ( __mul24( c0, r[e] ) + __mul24( c1, r[e + 1] ) + __mul24( c2, r[e + 2] ) + __mul24( c3, r[e + 3] ) + 32 ) >> 6, 0 ), maxv );
            if( xx0 < pdpcLev )
            {
              // (the weights of positions at and beyond the level are zero by themselves: 32 >> 6)
              if( F & CF_ANG0 )
              {
                const int sd = Sd[yy + 1] - t0;
                for( int e = 0; e < 4; e++ ) { const int wLp = 32 >> min( 31, ( ( xx0 + e ) << 1 ) >> pscale ); v[e] = min( max( ( __mul24( wLp, sd ) + ( v[e] << 6 ) + 32 ) >> 6, 0 ), maxv ); }
              }
              else
                for( int e = 0; e < 4; e++ )
                {
                  const int xx = xx0 + e;
                  const int wLp = 32 >> min( 31, 2 * xx >> angScale );
                  const int sv = Sd[min( yy + ( ( 256 + ( xx + 1 ) * invAngle ) >> 9 ) + 1, IT_MAXREF )];
                  v[e] = (int16_t) ( v[e] + ( ( __mul24( wLp, sv - v[e] ) + 32 ) >> 6 ) );
                }
            }
          }
          // store of the group: CIIP blend, residual, clipping, the tile (8-byte accesses where the positions allow it)
          if( vec )
          {
            const int to = tileBase + yy * IT_TSB + xx0, ri = ( ( yy - yb ) << lw ) + xx0;
            if( wIntra )
            {
              // predBlendIntraCiip (:935-944): the tile holds the inter prediction
              const uint2 tv = *reinterpret_cast<const uint2*>( &sh.tile[to] ); const int ip[4] = { (int) ( tv.x & 0xffff ), (int) ( tv.x >> 16 ), (int) ( tv.y & 0xffff ), (int) ( tv.y >> 16 ) };
              for( int e = 0; e < 4; e++ ) v[e] = ( ( 4 - wIntra ) * ip[e] + wIntra * v[e] + 2 ) >> 2;
            }
            if( hasResi )
            {
              const uint2 rv = *reinterpret_cast<const uint2*>( &W.resi[ri] ); const int r[4] = { (int16_t) ( rv.x & 0xffff ), (int16_t) ( rv.x >> 16 ), (int16_t) ( rv.y & 0xffff ), (int16_t) ( rv.y >> 16 ) };
              for( int e = 0; e < 4; e++ ) if( !ispGrp || ( ( ispResi >> ( ( xx0 + e ) >> ( 2 - ispGrp ) ) ) & 1 ) ) v[e] = clip_pel( v[e] + ( csOn ? lmcs_scale_resi( r[e], csScale, bd ) : r[e] ), bd );
            }
            *reinterpret_cast<uint2*>( &sh.tile[to] ) = make_uint2( ( v[0] & 0xffff ) | ( (uint32_t) v[1] << 16 ), ( v[2] & 0xffff ) | ( (uint32_t) v[3] << 16 ) );
          }
          else
          {
            for( int e = 0; e < 4; e++ ) if( e < g )
            {
              const int x = tr ? yy : xx0 + e, y = tr ? xx0 + e : yy;
              const int to = tileBase + y * IT_TSB + x;
              int vv = v[e];
              if( wIntra ) vv = ( ( 4 - wIntra ) * sh.tile[to] + wIntra * vv + 2 ) >> 2;
              if( hasResi && ( !ispGrp || ( ( ispResi >> ( x >> ( 2 - ispGrp ) ) ) & 1 ) ) ) { const int r = W.resi[( ( y - yb ) << lw ) + x]; vv = clip_pel( vv + ( csOn ? lmcs_scale_resi( r, csScale, bd ) : r ), bd ); }
              sh.tile[to] = (pel_t) vv;
            }
          }
        }
        IT_BT( 6 );
      }
      IT_DONE()
    }
#undef IT_DONE
#undef IT_BT
#undef IT_CSYNC
#undef CP
  }
  else
  {
    // ---- FINE: the publisher.  It follows the unit's items in order: when item q is in the tile (progress counter of the wavefront that predicts it), its
    // samples go to the picture with device-scope stores, and behind them (s_waitcnt vmcnt(0)) its cells are cleared - that is what the blocks of the CTUs
    // to the right and below (and the chroma blocks of this CTU: CCLM, the chroma scaling factor) wait for.  The cells of an ISP coding unit are cleared with
    // its last partition (partitions share cells).
    lds_barrier();                    // (the barrier behind the compute wavefronts' first round)
    const uint32_t qEndP = ( dbg & 4 ) ? i0 : i1;
    auto itemDone = [&]( uint32_t q ) { const int k = (int) ( q - iA ); return __hip_atomic_load( &sh.prog[k & ( IT_WAVES - 1 )], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP ) >= k / IT_WAVES + 1; };
#pragma unroll 1
    for( uint32_t q = iA; q < qEndP; )
    {
      // the items that are in the tile by now, in order, at most LEAF_PUBLISH_BATCH at a time: their stores go out together, one wait for all of them, then
      // their cells.  Measured at 4K (I picture alone): batches of 1 - 4231 us, of up to 8 - 4531 us: what a neighbouring CTU waits for is the FIRST item of a
      // batch, and it is held back by the stores of the others
      while( !itemDone( q ) ) __builtin_amdgcn_s_sleep( 1 );
      uint32_t n = 1;
      while( n < LEAF_PUBLISH_BATCH && q + n < qEndP && itemDone( q + n ) ) n++;
      asm volatile( "" ::: "memory" );
#pragma unroll 1
      for( uint32_t j = 0; j < n; j++ )
      {
        IntraItem it;
        IT_FETCH( it, q + j )
        const int lw = it.lw, rows = intra_part_rows( it ), wh = rows << lw, yb = IT_PART( it ) * rows;
        if( lw >= 2 && !( it.x & 3 ) )
        {
          for( int i = lane; i < ( wh >> 2 ); i += 64 )
          {
            const int x = it.x + ( ( i << 2 ) & ( ( 1 << lw ) - 1 ) ), y = it.y + yb + ( ( i << 2 ) >> lw );
            st_pel4_sc1( &plane[(size_t) y * pstride + x], *reinterpret_cast<const uint2*>( &TILE( x, y ) ) );
          }
        }
        else
          for( int i = lane; i < wh; i += 64 )
          {
            const int x = it.x + ( i & ( ( 1 << lw ) - 1 ) ), y = it.y + yb + ( i >> lw );
            st_pel_sc1( &plane[(size_t) y * pstride + x], TILE( x, y ) );
          }
      }
      asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
#pragma unroll 1
      for( uint32_t j = 0; j < n; j++ )
      {
        IntraItem it;
        IT_FETCH( it, q + j )
        const int rows = intra_part_rows( it ), yb = IT_PART( it ) * rows;
        const bool ispIt = !comp && it.mode <= 66 && ( it.flags & IT_F_ISP ) == IT_F_ISP && !( it.flags & IT_F_MIP );
        int cx0 = it.x, cy0 = it.y + yb, cw = 1 << it.lw, ch = rows;
        if( ispIt )
        {
          // the last partition of its coding unit? (the partitions follow each other in the list; they share cells: cleared with the last one)
          bool more = false;
          if( q + j + 1 < i1 ) { IntraItem nx; IT_FETCH( nx, q + j + 1 ) more = ( nx.flags & IT_F_ISP ) == IT_F_ISP && !( nx.flags & IT_F_MIP ) && nx.mode <= 66 && ( nx.tu & 0xfff ) != 0; }
          if( more ) continue;
          cx0 = it.x - (int) ( it.tu & 63 ); cy0 = it.y - (int) ( ( it.tu >> 6 ) & 63 ); cw = 1 << ( ( it.tu >> 12 ) & 7 ); ch = 1 << ( ( it.tu >> 15 ) & 7 );
        }
        leaf_set_cells( M.cell[comp], w4c, cu_, cx0, cy0, cw, ch, 0u, lane );
      }
      q += n;
    }
  }
  __syncthreads();                    // every block of the unit is in the tile
  IT_TRACE( 3 );
  // ---- write the reconstructed intra samples back to HBM (deferred so that the block loop never waits for a store)
  if( FINE ) { /* the publisher has stored every block */ }
  else if( borderOnly )
  {
    const int rows = min( PH, oy + S ) - oy, nch = ( min( PW, ox + S ) - ox + 7 ) >> 3;     // a chunk past the picture edge lands in the row padding
    for( int i = tid; i < rows * nch; i += IT_NT )
    {
      const int r = i / nch, c = i - r * nch;
      {
        const uint4 v = *reinterpret_cast<const uint4*>( &TILE( ox + c * 8, oy + r ) );
        pel_t* d = &plane[(size_t) ( oy + r ) * pstride + ox + c * 8];
        if( WT ) { st_pel4_sc1( d, make_uint2( v.x, v.y ) ); st_pel4_sc1( d + 4, make_uint2( v.z, v.w ) ); } else *reinterpret_cast<uint4*>( d ) = v;
      }
    }
  }
  else
  {
    // one block per wavefront
    for( uint32_t q = i0 + wv; q < ( ( dbg & 4 ) ? i0 : i1 ); q += IT_WAVES )
    {
      IntraItem it;
      IT_FETCH( it, q )
      const int lw = it.lw, rows = intra_part_rows( it ), wh = rows << lw, yb = IT_PART( it ) * rows;
      if( lw >= 2 && !( it.x & 3 ) )
      {
        // four samples (8 bytes) per lane: block positions and widths are multiples of 4 samples
        for( int i = lane; i < ( wh >> 2 ); i += 64 )
        {
          const int x = it.x + ( ( i << 2 ) & ( ( 1 << lw ) - 1 ) ), y = it.y + yb + ( ( i << 2 ) >> lw );
          st_pel4_sc1( &plane[(size_t) y * pstride + x], *reinterpret_cast<const uint2*>( &TILE( x, y ) ) );
        }
      }
      else
        for( int i = lane; i < wh; i += 64 )
        {
          const int x = it.x + ( i & ( ( 1 << lw ) - 1 ) ), y = it.y + yb + ( i >> lw );
          st_pel_sc1( &plane[(size_t) y * pstride + x], TILE( x, y ) );
        }
    }
  }
#undef TILE
#undef IT_FETCH
  // ---- publish: all stores of the workgroup drained, one agent-scope release, then the flag
  IT_TRACE( 4 );
  if( trace && threadIdx.x == 0 ) { trace[(size_t) 8 * tr_ticket + 6] = ( (unsigned long long) ( i1 - i0 ) << 32 ) | ent; trace[(size_t) 8 * tr_ticket + 7] = un->ndeps; }
  if( !publish ) continue;
  asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
  __syncthreads();
  if( tid == 0 )
  {
    if( !WT ) __builtin_amdgcn_fence( __ATOMIC_RELEASE, "agent" );
    asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
    __hip_atomic_store( &sync[1 + ticket], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
  }
  }     // next ticket
#undef IT_TRACE
}

// ints of the per-lane buffer of the intra stage: ticket + one flag per unit, then (64-dword aligned) one parameter record per block
size_t intra_ctx_offset( int numUnits ) { return ( (size_t) 1 + (size_t) numUnits + 63 ) & ~(size_t) 63; }
size_t intra_sync_ints( int numUnits, int numItems ) { return intra_ctx_offset( numUnits ) + (size_t) numItems * IT_CTX; }

// =====================================================================================================================
// k_resi_add — the residual of the inter-predicted chroma blocks of a picture with LMCS chroma residual scaling (DecCu::finishLMCSAndReco,
// DecCu.cpp:483-520; Reshape::calculateChromaAdjVpduNei, Reshape.cpp:192-274): the factor of the block's VPDU from the reconstructed luma left
// of and above the VPDU's first CU, the scaled residual (k_itrans stored it) onto the prediction (AreaBuf::scaleSignal, Buffer.cpp:412).
// Runs between the luma units and the chroma units of the intra stage: every luma sample of the picture is final by then, and the chroma
// blocks of the intra stage that read these blocks come afterwards - so the blocks need no place in the stage's dependency graph.
// One wavefront per block.
// =====================================================================================================================
__global__ __launch_bounds__( 256 ) void k_resi_add( IntraPic pic, const IntraItem* __restrict__ items, int numItems )
{
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + ( threadIdx.x >> 6 );
  if( q >= numItems ) return;
  uint4 rec = reinterpret_cast<const uint4*>( items )[q];
  IntraItem it;
  { uint32_t* op = reinterpret_cast<uint32_t*>( &it ); op[0] = __builtin_amdgcn_readfirstlane( rec.x ); op[1] = __builtin_amdgcn_readfirstlane( rec.y ); op[2] = __builtin_amdgcn_readfirstlane( rec.z ); op[3] = __builtin_amdgcn_readfirstlane( rec.w ); }
  const int comp = IT_COMP( it ), bd = pic.bitDepth;
  const int x0 = it.x, y0 = it.y, lw = it.lw, wh = 1 << ( it.lw + it.lh );
  pel_t* __restrict__ plane = pic.plane[comp];
  const pel_t* __restrict__ rs = pic.resi[comp];
  const int pstride = pic.stride[comp], rstride = pic.rstride[comp];
  const bool cs = ( it.flags & IT_F_CSCALE ) != 0;
  const int f = cs ? lmcs_cscale_factor_wave( pic, x0 << 1, y0 << 1, lane ) : 0;
  if( lw >= 2 && !( x0 & 3 ) )
  {
    // four samples of a row per lane (8-byte accesses; x0 and the row strides are multiples of 4 samples - not so for the chroma of
    // an 8-wide inter CU that is the middle part of a ternary split of 16)
    for( int i = lane; i < ( wh >> 2 ); i += 64 )
    {
      const int x = x0 + ( ( i << 2 ) & ( ( 1 << lw ) - 1 ) ), y = y0 + ( ( i << 2 ) >> lw );
      const uint2 rv = *reinterpret_cast<const uint2*>( &rs[(size_t) y * rstride + x] );
      uint2* pp = reinterpret_cast<uint2*>( &plane[(size_t) y * pstride + x] );
      const uint2 pv = *pp;
      int r[4] = { (int16_t) ( rv.x & 0xffff ), (int16_t) ( rv.x >> 16 ), (int16_t) ( rv.y & 0xffff ), (int16_t) ( rv.y >> 16 ) };
      const int pr[4] = { (int) ( pv.x & 0xffff ), (int) ( pv.x >> 16 ), (int) ( pv.y & 0xffff ), (int) ( pv.y >> 16 ) };
      int o[4];
      for( int e = 0; e < 4; e++ ) o[e] = clip_pel( pr[e] + ( cs ? lmcs_scale_resi( r[e], f, bd ) : r[e] ), bd );
      *pp = make_uint2( (uint32_t) o[0] | ( (uint32_t) o[1] << 16 ), (uint32_t) o[2] | ( (uint32_t) o[3] << 16 ) );
    }
  }
  else
    for( int i = lane; i < wh; i += 64 )
    {
      const int x = x0 + ( i & ( ( 1 << lw ) - 1 ) ), y = y0 + ( i >> lw );
      const int r = (int16_t) rs[(size_t) y * rstride + x];
      plane[(size_t) y * pstride + x] = (pel_t) clip_pel( plane[(size_t) y * pstride + x] + ( cs ? lmcs_scale_resi( r, f, bd ) : r ), bd );
    }
}

static IntraPic intra_pic( const PicDev& pic, const DevPlanes& reco, const DevPlanes& resi )
{
  IntraPic ip;
  for( int c = 0; c < 3; c++ ) { ip.plane[c] = reco.p[c]; ip.resi[c] = resi.p[c]; ip.stride[c] = reco.stride[c]; ip.rstride[c] = resi.stride[c]; ip.w[c] = reco.w[c]; ip.h[c] = reco.h[c]; }
  ip.csVpdu = pic.csVpdu; ip.lmcs = pic.lmcs; ip.vpdusX = pic.vpdusX; ip.vpduLog2 = pic.vpduLog2; ip.ctusX = pic.ctus_x; ip.log2Ctu = pic.hdr.log2_ctu;
  ip.bitDepth = pic.hdr.bit_depth; ip.width = pic.hdr.width; ip.height = pic.hdr.height; ip.colloc = ( pic.hdr.tool_flags & VVR_TOOL_CCLM_COLLOC ) ? 1 : 0;
  return ip;
}

#include "vvr_intra_leaf.inc"

// =====================================================================================================================
// k_prep - the picture's first launch: three passes over its records that depend on nothing but the uploaded image and on each other not at all,
// as sections of one grid (each was a launch of 6 to 22 us that could not fill the device):
//   * prep_expand_mc   the motion-compensation tiles of plain, BDOF and DMVR CUs from the CU records
//   * prep_lf_maps     the per-cell records and motion of the deblocking edge derivation (k_lf_init, the next launch of that stage, reads them)
//   * prep_intra_mark  the cells the scattered intra blocks of the picture are going to produce, and their ticket (k_intra_leaf)
// =====================================================================================================================
struct PrepArgs
{
  const McCuRef* mcCus; int numMcCus; McItem *plain, *bdof, *dmvr;
  int numCu, numTu, numSb, dbg; LfCell *cell, *cellC; LfMv* mv; uint32_t* ref; const LfSbCell* sb;
  const IntraItem *items, *resi; int numItems, numResi; LeafMaps M; int* sync;
  int blocksExpand, blocksMaps;
};
__global__ __launch_bounds__( 256 ) void k_prep( PicDev pic, PrepArgs a )
{
  int bid = blockIdx.x;
  // (the longest section first: the maps' waves walk lists, the other two are a handful of stores)
  if( bid < a.blocksMaps ) { prep_lf_maps( bid, pic, a.numCu, a.numTu, a.cell, a.cellC, a.mv, a.ref, a.sb, a.numSb, a.dbg ); return; }
  bid -= a.blocksMaps;
  if( bid < a.blocksExpand ) { prep_expand_mc( bid, pic.cu, a.mcCus, a.numMcCus, a.plain, a.bdof, a.dmvr ); return; }
  bid -= a.blocksExpand;
  prep_intra_mark( bid, a.items, a.numItems, a.resi, a.numResi, a.M, a.sync );
}

void launch_prep( hipStream_t s, const PicDev& pic, const PrepWork& w )
{
  PrepArgs a = {};
  if( w.numMcCus ) { a.mcCus = w.mcCus; a.numMcCus = w.numMcCus; a.plain = w.plain; a.bdof = w.bdof; a.dmvr = w.dmvr; a.blocksExpand = ( w.numMcCus + 3 ) / 4; }
  if( w.lfMaps && w.numCu && w.numTu )
  {
    a.numCu = (int) w.numCu; a.numTu = (int) w.numTu; a.cell = w.cell; a.cellC = w.cellC; a.mv = w.mv; a.ref = w.ref; a.sb = w.sb; a.numSb = w.numSb;
#ifdef VVR_DEV_ENV
    static const int dbgEnv = getenv( "VVR_LFM_DBG" ) ? atoi( getenv( "VVR_LFM_DBG" ) ) : 0;      // developer build: 1 affine cells like plain ones, 2 no motion stores, 4 no stores (timing only)
    a.dbg = dbgEnv;
#endif
    a.blocksMaps = (int) ( ( 8 * w.numTu + w.numSb + 255 ) / 256 );
  }
  int blocksMark = 0;
  if( w.numItems )
  {
    a.items = w.items; a.numItems = w.numItems; a.resi = w.resi; a.numResi = w.numResi;
    a.M = leaf_maps_of( pic, w.maps, w.mapW4, w.mapH4 ); a.sync = reinterpret_cast<int*>( w.maps + w.mapInts - 64 );
    blocksMark = ( w.numItems + w.numResi + 15 ) / 16;
  }
  const int blocks = a.blocksMaps + a.blocksExpand + blocksMark;
  if( blocks ) hipLaunchKernelGGL( k_prep, dim3( blocks ), dim3( 256 ), 0, s, pic, a );
}

void launch_resi_add( hipStream_t s, const PicDev& pic, DevPlanes reco, DevPlanes resi, const IntraItem* items, int numItems )
{
  if( !numItems ) return;
  hipLaunchKernelGGL( k_resi_add, dim3( ( numItems + 3 ) / 4 ), dim3( 256 ), 0, s, intra_pic( pic, reco, resi ), items, numItems );
}

// The units [ticket0, ticket1) of the table.  A picture whose inter blocks carry scaled chroma residuals runs the stage in two launches - the luma
// units, then (behind k_resi_add) the chroma units: the flags of the first launch stay set, so a chroma unit that names a luma producer finds it done.
void launch_intra( hipStream_t s, const PicDev& pic, DevPlanes reco, DevPlanes resi, const IntraItem* items, int numItems, const IntraUnit* units, int numActive, int ticket0, int ticket1,
                   int numWorkgroups, int* sync, int wide, uint32_t* maps, size_t mapInts, int mapW4, int mapH4, int* errWord )
{
  if( !numActive || ticket1 <= ticket0 ) return;
  numWorkgroups = std::max( 1, std::min( numWorkgroups, ticket1 - ticket0 ) );
  const IntraPic ip = intra_pic( pic, reco, resi );
  uint32_t* ctx = reinterpret_cast<uint32_t*>( sync ) + intra_ctx_offset( numActive );
  if( ticket0 == 0 )
  {
    hipMemsetAsync( sync, 0, sizeof( int ) * ( 1 + (size_t) numActive ), s );
    // the blocks' parameter records live behind the flags (intra_sync_ints): one pass over all blocks writes them
    hipLaunchKernelGGL( k_intra_setup, dim3( ( numItems + 255 ) / 256 ), dim3( 256 ), 0, s, ip, items, numItems, ctx );
  }
  else hipMemsetD32Async( (hipDeviceptr_t) sync, ticket0, 1, s );      // the ticket counter of the second launch starts where the first ended
  numActive = ticket1;
  // an I picture the stream waits for (`wide`: it has the priority lane, or the context has one lane): workgroups of eight wavefronts - the bands of a block side by
  // side - and as many workgroups as the host stage counted; I pictures among I pictures (all-intra: a dozen in flight fill the device, what counts is how many
  // units are resident) and the scattered intra blocks of an inter picture: four wavefronts, and half the workgroups for the I picture (all-intra at 4K, 12
  // pictures in flight: 1065 against 810 pictures/s, profiles/round4_lanes_and_host_threads.txt)
  int waves = wide ? 8 : 4;
  if( pic.hdr.slice_type == 2 && !wide ) numWorkgroups = std::max( 1, numWorkgroups / 2 );
  // FINE (maps != nullptr: a picture whose units are all whole CTUs of intra CUs, launched in one go): every cell of the picture is pending until the publisher of
  // its CTU has stored it; with the units waiting for nobody more of them are worth having resident (the front of the block-level dependency chain spans
  // several CTU diagonals)
  LeafMaps M = {};
  int* lsync = nullptr;
#ifdef VVR_INTRA_DEV
  const bool fine = maps != nullptr && ticket0 == 0 && ticket1 == numActive;
#else
  const bool fine = false;
#endif
  if( fine )
  {
    const size_t cellsPerMap = (size_t) mapW4 * mapH4;
    for( int k = 0; k < 3; k++ ) M.cell[k] = maps + (size_t) k * cellsPerMap;
    M.w4 = pic.w4; M.h4 = pic.h4;
    lsync = errWord;
    for( int k = 0; k < ( pic.hdr.chroma_format ? 3 : 1 ); k++ ) hipMemsetD32Async( (hipDeviceptr_t) M.cell[k], 1, (size_t) pic.w4 * pic.h4, s );
    static const int fineWg = getenv( "VVR_INTRA_FINE_WG" ) ? atoi( getenv( "VVR_INTRA_FINE_WG" ) ) : 0;
    numWorkgroups = std::max( 1, std::min( ticket1 - ticket0, fineWg > 0 ? fineWg : ( wide ? 256 : 4 * numWorkgroups ) ) );
  }
#ifndef VVR_INTRA_DEV
  // (the block-by-block variant k_intra<.., FINE> - measured in round 5, slower under load - exists in the developer build only: `maps` is never set here)
  if( waves == 8 )      hipLaunchKernelGGL( ( k_intra<8, false> ), dim3( numWorkgroups ), dim3( 512 ), 0, s, ip, items, ctx, units, numActive, sync, M, lsync );
  else                  hipLaunchKernelGGL( ( k_intra<4, false> ), dim3( numWorkgroups ), dim3( 256 ), 0, s, ip, items, ctx, units, numActive, sync, M, lsync );
#else
  if( const char* e = getenv( "VVR_INTRA_WAVES" ) ) waves = atoi( e ) == 8 ? 8 : 4;
  static const int dbg = getenv( "VVR_INTRA_DBG" ) ? atoi( getenv( "VVR_INTRA_DBG" ) ) : 0;     // timing experiments only (results are wrong with any bit set)
  static const bool tr = getenv( "VVR_INTRA_TRACE" ) != nullptr;
  unsigned long long* trace = nullptr, * btrace = nullptr;
  const size_t nItems = 1 << 20;      // (block timeline: sized generously, indexed by item)
  if( tr ) { hipMalloc( (void**) &trace, sizeof( unsigned long long ) * 8 * (size_t) numActive ); hipMemsetAsync( trace, 0, sizeof( unsigned long long ) * 8 * (size_t) numActive, s );
             hipMalloc( (void**) &btrace, sizeof( unsigned long long ) * 8 * nItems ); hipMemsetAsync( btrace, 0, sizeof( unsigned long long ) * 8 * nItems, s ); }
  if( fine )
  {
    if( waves == 8 ) hipLaunchKernelGGL( ( k_intra<8, true> ), dim3( numWorkgroups ), dim3( 576 ), 0, s, ip, items, ctx, units, numActive, sync, M, lsync, dbg, trace, btrace );
    else             hipLaunchKernelGGL( ( k_intra<4, true> ), dim3( numWorkgroups ), dim3( 320 ), 0, s, ip, items, ctx, units, numActive, sync, M, lsync, dbg, trace, btrace );
  }
  else if( waves == 8 ) hipLaunchKernelGGL( ( k_intra<8, false> ), dim3( numWorkgroups ), dim3( 512 ), 0, s, ip, items, ctx, units, numActive, sync, M, lsync, dbg, trace, btrace );
  else                  hipLaunchKernelGGL( ( k_intra<4, false> ), dim3( numWorkgroups ), dim3( 256 ), 0, s, ip, items, ctx, units, numActive, sync, M, lsync, dbg, trace, btrace );
  if( tr )
  {
    // developer timeline: ticket, phase time stamps (100 MHz), block count / unit word, number of producers; per block four shader-clock stamps
    std::vector<unsigned long long> h( 8 * (size_t) numActive ), hb( 8 * nItems );
    hipStreamSynchronize( s );
    hipMemcpy( h.data(), trace, h.size() * sizeof( unsigned long long ), hipMemcpyDeviceToHost );
    hipMemcpy( hb.data(), btrace, hb.size() * sizeof( unsigned long long ), hipMemcpyDeviceToHost );
    hipFree( trace ); hipFree( btrace );
    std::vector<IntraUnit> hu( numActive ); hipMemcpy( hu.data(), units, sizeof( IntraUnit ) * (size_t) numActive, hipMemcpyDeviceToHost );
    size_t maxItem = 0; for( auto& u : hu ) maxItem = std::max<size_t>( maxItem, u.i1 );
    std::vector<IntraItem> hi( maxItem ); hipMemcpy( hi.data(), items, sizeof( IntraItem ) * maxItem, hipMemcpyDeviceToHost );
    char name[128]; snprintf( name, sizeof( name ), "gpurun_out/intra_trace_poc%d.bin", pic.hdr.poc );
    if( FILE* f = fopen( name, "wb" ) ) { fwrite( h.data(), sizeof( unsigned long long ), h.size(), f ); fclose( f ); }
    snprintf( name, sizeof( name ), "gpurun_out/intra_btrace_poc%d.bin", pic.hdr.poc );
    if( FILE* f = fopen( name, "wb" ) ) { fwrite( hb.data(), sizeof( unsigned long long ), 8 * maxItem, f ); fclose( f ); }
    snprintf( name, sizeof( name ), "gpurun_out/intra_units_poc%d.bin", pic.hdr.poc );
    if( FILE* f = fopen( name, "wb" ) ) { fwrite( hu.data(), sizeof( IntraUnit ), hu.size(), f ); fclose( f ); }
    snprintf( name, sizeof( name ), "gpurun_out/intra_items_poc%d.bin", pic.hdr.poc );
    if( FILE* f = fopen( name, "wb" ) ) { fwrite( hi.data(), sizeof( IntraItem ), hi.size(), f ); fclose( f ); }
  }
#endif
}
