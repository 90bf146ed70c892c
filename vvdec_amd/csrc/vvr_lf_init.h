// vvdec_amd/csrc/vvr_lf_init.h — edge parameters of the deblocking filter derived from the CU / TU records (VVR_TOOL_LFP_ON_DEVICE).
//
// The reference fills one LoopFilterParam per 4x4 unit and edge direction in its LF_INIT task (LoopFilter::calcFilterStrengthsCTU,
// LoopFilter.cpp:495-1360: per CU, per transform unit, then per sub-block edge); the back-end takes that table as input (vvr_picture.lfp) or, with
// VVR_TOOL_LFP_ON_DEVICE, derives it here: ONE function per cell and direction that looks at nothing but the two transform units on either side of
// the cell's left / top edge, their CUs and - for edges decided by motion - the motion of the two cells.  No order between cells, so the device
// runs one thread per cell (k_lf_init); the same source compiles for the host, where the tests compare its table with the reference's own
// (tests/test_lf_init.py: generated pictures through oracle/_ref, parser-fed pictures through the drop-in).
//
// What the filter kernels read of a table entry - and therefore what is derived: the boundary strengths (bs), the luma filter lengths of both sides,
// the averaged QPs of the components that are filtered, the long-chroma-filter flag.  Fields of an edge that is not filtered are left zero (the
// reference's table keeps lengths and QPs there; nothing reads them).
#pragma once
#include <stdint.h>
#include "../../include/vvr.h"

#ifdef __HIPCC__
#define VVR_HD __host__ __device__ __forceinline__
#else
#define VVR_HD inline
#endif

struct LfInitView {
  const vvr_pic_header*   hdr;
  const vvr_cu*           cu;
  const vvr_tu*           tu;
  const int32_t*          tuOf4;       // transform unit of the luma tree (or of the joint tree) that covers the cell
  const int32_t*          tuOf4C;      // transform unit of the chroma tree: valid where the cell's luma CU belongs to a luma-only tree (dual tree, local dual tree)
  const vvr_motion*       sbMotion;    // motion of the cells of CUs whose motion varies inside the CU (affine, SbTMVP, GPM); picture raster, other cells undefined
  const uint16_t*         ctuSlice;    // NULL: one slice
  const uint16_t*         ctuTile;     // NULL: one tile
  const uint16_t*         ctuSubpic;   // NULL: one sub-picture
  const vvr_subpic*       subpics;
  const vvr_slice_header* slices;      // NULL: the picture header's values
  int                     w4, h4, ctusX;
  int                     numTu, numCu;
};

// one cell of a CU whose motion varies inside the CU, as the host hands it over (SbTMVP and GPM CUs; affine CUs too unless the back-end spans their
// sub-block vectors itself, VVR_TOOL_AFFINE_MV_ON_DEVICE): scattered into LfInitView::sbMotion before the cells are derived
struct LfSbCell { uint32_t cell; vvr_motion m; };

// the cells a transform unit covers, into the map of its tree
VVR_HD bool lfi_tu_owns_cells( const vvr_tu& t );
VVR_HD void lfi_map_tu( const vvr_tu& t, int tuIdx, const vvr_cu& c, int32_t* tuOf4, int32_t* tuOf4C, int w4, int h4 );

// does a transform unit own the cells it touches?  Partitions narrower (lower) than a cell share it: the last one in decoding order - the one that ends
// on the cell's far side - is what a later look-up of the cell finds (the reference's edge pass visits the partitions in order, LoopFilter.cpp:543-567)
VVR_HD bool lfi_tu_owns_cells( const vvr_tu& t ) { return !( ( t.w < 4 && ( ( t.x + t.w ) & 3 ) ) || ( t.h < 4 && ( ( t.y + t.h ) & 3 ) ) ); }

VVR_HD int lfi_idx( int i, int n ) { return i < 0 ? 0 : i >= n ? n - 1 : i; }

struct LfiMotion { int32_t mv[2][2]; int32_t poc[2]; };      // poc: the reference picture of the list, INT32_MIN = list not used

VVR_HD bool lfi_sub_block_cu( const vvr_cu& c ) { return c.pred_mode == VVR_PRED_INTER && ( c.flags & ( VVR_CU_AFFINE | VVR_CU_SBTMVP | VVR_CU_GEO ) ) != 0; }

VVR_HD void lfi_map_tu( const vvr_tu& t, int tuIdx, const vvr_cu& c, int32_t* tuOf4, int32_t* tuOf4C, int w4, int h4 )
{
  if( !lfi_tu_owns_cells( t ) ) return;
  int32_t* map = c.tree == VVR_TREE_CHROMA ? tuOf4C : tuOf4;
  const int x0 = t.x >> 2, y0 = t.y >> 2;
  int x1 = ( t.x + t.w + 3 ) >> 2, y1 = ( t.y + t.h + 3 ) >> 2;
  if( x1 > w4 ) x1 = w4;
  if( y1 > h4 ) y1 = h4;
  for( int y = y0; y < y1; y++ ) for( int x = x0; x < x1; x++ ) map[(size_t) y * w4 + x] = tuIdx;
}

VVR_HD int lfi_ilog2( int v ) { int l = 0; while( ( 1 << l ) < v ) l++; return l; }
VVR_HD int lfi_max( int a, int b ) { return a > b ? a : b; }
VVR_HD int lfi_min( int a, int b ) { return a < b ? a : b; }
VVR_HD bool lfi_affine_spread_over_limit( int a, int b, int c, int d, int predType )      // InterPrediction::isSubblockVectorSpreadOverLimit (InterPrediction.cpp:892)
{
  const int s4 = 4 << 11, ft = 6;
  if( predType == 3 )
  {
    int rw = lfi_max( lfi_max( 0, 4 * a + s4 ), lfi_max( 4 * c, 4 * a + 4 * c + s4 ) ) - lfi_min( lfi_min( 0, 4 * a + s4 ), lfi_min( 4 * c, 4 * a + 4 * c + s4 ) );
    int rh = lfi_max( lfi_max( 0, 4 * b ), lfi_max( 4 * d + s4, 4 * b + 4 * d + s4 ) ) - lfi_min( lfi_min( 0, 4 * b ), lfi_min( 4 * d + s4, 4 * b + 4 * d + s4 ) );
    rw = ( rw >> 11 ) + ft + 3; rh = ( rh >> 11 ) + ft + 3;
    return rw * rh > ( ft + 9 ) * ( ft + 9 );
  }
  int rw = lfi_max( 0, 4 * a + s4 ) - lfi_min( 0, 4 * a + s4 ), rh = lfi_max( 0, 4 * b ) - lfi_min( 0, 4 * b );
  rw = ( rw >> 11 ) + ft + 3; rh = ( rh >> 11 ) + ft + 3;
  if( rw * rh > ( ft + 9 ) * ( ft + 5 ) ) return true;
  rw = lfi_max( 0, 4 * c ) - lfi_min( 0, 4 * c ); rh = lfi_max( 0, 4 * d + s4 ) - lfi_min( 0, 4 * d + s4 );
  rw = ( rw >> 11 ) + ft + 3; rh = ( rh >> 11 ) + ft + 3;
  return rw * rh > ( ft + 5 ) * ( ft + 9 );
}
// the stored motion vector of the 4x4 sub-block (wx, wy) of an affine CU (PU::setAllAffineMv, UnitTools.cpp:2689-2810; the same arithmetic k_mc_affine predicts with)
VVR_HD void lfi_affine_mv( const vvr_cu& cu, int l, int wx, int wy, int32_t mv[2] )
{
  const int lw = lfi_ilog2( cu.w ), lh = lfi_ilog2( cu.h );
  const int dHX = ( cu.mv[l][1][0] - cu.mv[l][0][0] ) * ( 1 << ( 7 - lw ) ), dHY = ( cu.mv[l][1][1] - cu.mv[l][0][1] ) * ( 1 << ( 7 - lw ) );
  int dVX, dVY;
  if( cu.flags & VVR_CU_AFFINE_6P ) { dVX = ( cu.mv[l][2][0] - cu.mv[l][0][0] ) * ( 1 << ( 7 - lh ) ); dVY = ( cu.mv[l][2][1] - cu.mv[l][0][1] ) * ( 1 << ( 7 - lh ) ); }
  else { dVX = -dHY; dVY = dHX; }
  const bool over = lfi_affine_spread_over_limit( dHX, dHY, dVX, dVY, cu.inter_dir );
  const int px = over ? cu.w >> 1 : 2 + 4 * wx, py = over ? cu.h >> 1 : 2 + 4 * wy;
  int mx = cu.mv[l][0][0] * 128 + dHX * px + dVX * py, my = cu.mv[l][0][1] * 128 + dHY * px + dVY * py;
  mx = ( mx + 64 - ( mx >= 0 ) ) >> 7; my = ( my + 64 - ( my >= 0 ) ) >> 7;
  mv[0] = lfi_min( ( 1 << 17 ) - 1, lfi_max( -( 1 << 17 ), mx ) ); mv[1] = lfi_min( ( 1 << 17 ) - 1, lfi_max( -( 1 << 17 ), my ) );
}

VVR_HD LfiMotion lfi_motion( const LfInitView& V, int x4, int y4, const vvr_cu& c )
{
  LfiMotion m;
  const vvr_pic_header& h = *V.hdr;
  if( c.pred_mode == VVR_PRED_INTER && ( c.flags & VVR_CU_AFFINE ) && ( h.tool_flags & VVR_TOOL_AFFINE_MV_ON_DEVICE ) )
  {
    for( int l = 0; l < 2; l++ )
    {
      m.mv[l][0] = m.mv[l][1] = 0; m.poc[l] = INT32_MIN;
      if( c.ref_idx[l] < 0 ) continue;
      lfi_affine_mv( c, l, x4 - ( c.x >> 2 ), y4 - ( c.y >> 2 ), m.mv[l] );
      m.poc[l] = h.ref_poc[l][c.ref_idx[l] & ( VVR_MAX_REFS - 1 )];
    }
    return m;
  }
  if( lfi_sub_block_cu( c ) )
  {
    const vvr_motion& s = V.sbMotion[(size_t) y4 * V.w4 + x4];
    for( int l = 0; l < 2; l++ ) { m.mv[l][0] = s.mv[l][0]; m.mv[l][1] = s.mv[l][1]; m.poc[l] = s.ref_idx[l] >= 0 ? h.ref_poc[l][s.ref_idx[l] & ( VVR_MAX_REFS - 1 )] : INT32_MIN; }
    return m;
  }
  for( int l = 0; l < 2; l++ )
  {
    const bool on = c.pred_mode == VVR_PRED_INTER && c.ref_idx[l] >= 0;
    m.mv[l][0] = on ? c.mv[l][0][0] : 0; m.mv[l][1] = on ? c.mv[l][0][1] : 0;
    m.poc[l] = on ? h.ref_poc[l][c.ref_idx[l] & ( VVR_MAX_REFS - 1 )] : INT32_MIN;
  }
  return m;
}

VVR_HD bool lfi_far( const int32_t a[2], const int32_t b[2] )
{
  const int dx = a[0] - b[0], dy = a[1] - b[1];
  return ( dx < 0 ? -dx : dx ) >= 8 || ( dy < 0 ? -dy : dy ) >= 8;      // half a luma sample in 1/16 units
}

// boundary strength of an edge between two inter-predicted cells without coded residual (LoopFilter.cpp:1222-1345): 1 when they predict from
// different pictures, from a different number of pictures, or with motion vectors half a sample or more apart
VVR_HD int lfi_motion_bs( const LfiMotion& q, const LfiMotion& p )
{
  const int nq = ( q.poc[0] != INT32_MIN ) + ( q.poc[1] != INT32_MIN ), np = ( p.poc[0] != INT32_MIN ) + ( p.poc[1] != INT32_MIN );
  if( nq != np ) return 1;
  if( nq == 1 )
  {
    const int lq = q.poc[0] != INT32_MIN ? 0 : 1, lp = p.poc[0] != INT32_MIN ? 0 : 1;
    return ( q.poc[lq] != p.poc[lp] || lfi_far( q.mv[lq], p.mv[lp] ) ) ? 1 : 0;
  }
  if( nq == 0 ) return 0;
  if( !( ( q.poc[0] == p.poc[0] && q.poc[1] == p.poc[1] ) || ( q.poc[0] == p.poc[1] && q.poc[1] == p.poc[0] ) ) ) return 1;
  if( p.poc[0] != p.poc[1] )
  {
    if( q.poc[0] == p.poc[0] ) return ( lfi_far( q.mv[0], p.mv[0] ) || lfi_far( q.mv[1], p.mv[1] ) ) ? 1 : 0;
    return ( lfi_far( q.mv[0], p.mv[1] ) || lfi_far( q.mv[1], p.mv[0] ) ) ? 1 : 0;
  }
  return ( ( lfi_far( q.mv[0], p.mv[0] ) || lfi_far( q.mv[1], p.mv[1] ) ) && ( lfi_far( q.mv[0], p.mv[1] ) || lfi_far( q.mv[1], p.mv[0] ) ) ) ? 1 : 0;
}

// may the edge between CTU a (the cell's) and CTU b (its left / upper neighbour's) be deblocked?  pps_loop_filter_across_slices / _tiles_enabled_flag;
// between two sub-pictures the flag of both (xGetLoopfilterParam, LoopFilter.cpp:1062-1091)
VVR_HD bool lfi_may_cross( const LfInitView& V, int a, int b )
{
  if( a == b ) return true;
  const uint32_t f = V.hdr->tool_flags;
  if( ( f & VVR_TOOL_NO_LF_ACROSS_SLICES ) && V.ctuSlice && V.ctuSlice[a] != V.ctuSlice[b] ) return false;
  if( ( f & VVR_TOOL_NO_LF_ACROSS_TILES ) && V.ctuTile && V.ctuTile[a] != V.ctuTile[b] ) return false;
  if( V.ctuSubpic && V.ctuSubpic[a] != V.ctuSubpic[b] && !( V.subpics[V.ctuSubpic[a]].lf_across && V.subpics[V.ctuSubpic[b]].lf_across ) ) return false;
  return true;
}

// is the left (d = 0) / top (d = 1) edge of the cell on a virtual boundary of the picture header?  (xDeriveEdgefilterParam, LoopFilter.cpp:669-690)
VVR_HD bool lfi_on_virtual_boundary( const vvr_pic_header& h, int d, int x4, int y4 )
{
  if( d == 0 ) { for( int i = 0; i < h.num_ver_vb && i < 3; i++ ) if( h.vb_pos_x[i] == ( x4 << 2 ) ) return true; }
  else         { for( int i = 0; i < h.num_hor_vb && i < 3; i++ ) if( h.vb_pos_y[i] == ( y4 << 2 ) ) return true; }
  return false;
}

// the table entry of cell (x4, y4) for the edges of direction d (0: the cell's left edge, vertical edges; 1: its top edge)
VVR_HD vvr_lfp lf_init_cell( const LfInitView& V, int d, int x4, int y4 )
{
  vvr_lfp L; L.qp[0] = L.qp[1] = L.qp[2] = 0; L.bs = 0; L.side_max_filt_length = 0; L.flags = 0; L.pad[0] = L.pad[1] = 0;
  const int px4 = d == 0 ? x4 - 1 : x4, py4 = d == 0 ? y4 : y4 - 1;
  if( px4 < 0 || py4 < 0 ) return L;                          // picture boundary
  const vvr_pic_header& h = *V.hdr;
  const int l2c = h.log2_ctu - 2;
  const int ctuQ = ( y4 >> l2c ) * V.ctusX + ( x4 >> l2c ), ctuP = ( py4 >> l2c ) * V.ctusX + ( px4 >> l2c );
  // a slice that switches deblocking off in a picture that deblocks: the edges of its CTUs are left alone (LoopFilter.cpp:366,423)
  if( V.slices && V.ctuSlice && ( V.slices[V.ctuSlice[ctuQ]].tool_flags & VVR_TOOL_DEBLOCK_OFF ) ) return L;
  const int step = d == 0 ? 1 : V.w4;
  const int iq = y4 * V.w4 + x4, ip = iq - step;
  // (indices are kept inside the arrays whatever the maps hold: a description whose transform units do not cover the picture is refused by the host's checks,
  // but a kernel must not depend on that)
  const int tq = lfi_idx( V.tuOf4[iq], V.numTu ), tp = lfi_idx( V.tuOf4[ip], V.numTu );
  const vvr_tu& TQ = V.tu[tq]; const vvr_tu& TP = V.tu[tp];
  const vvr_cu& CQ = V.cu[lfi_idx( (int) TQ.cu, V.numCu )]; const vvr_cu& CP = V.cu[lfi_idx( (int) TP.cu, V.numCu )];
  const bool onVb = lfi_on_virtual_boundary( h, d, x4, y4 );
  const bool open = !onVb && lfi_may_cross( V, ctuQ, ctuP );    // the edge may be filtered at all
  const int posPerp = ( d == 0 ? x4 : y4 ) << 2;
  int bsY = 0, bsCb = 0, bsCr = 0, lenP = 0, lenQ = 0, qpY = 0;
  bool te = false;
  if( tq != tp )
  {
    // ---- an edge of the luma transform grid: filter lengths from the transform sizes across the edge (:905-922)
    te = true;
    const int sizeQ = d == 0 ? TQ.w : TQ.h, sizeP = d == 0 ? TP.w : TP.h;
    if( sizeP <= 4 || sizeQ <= 4 ) lenP = lenQ = 1;
    // (:920 cuP->affineFlag(): a CU in sub-block merge mode carries that flag whether its candidate was an affine one or the SbTMVP one, DecCu.cpp:746-767)
    else { lenP = sizeP >= 32 ? ( ( CP.flags & ( VVR_CU_AFFINE | VVR_CU_SBTMVP ) ) && CP.pred_mode == VVR_PRED_INTER ? 5 : 7 ) : 3; lenQ = sizeQ >= 32 ? 7 : 3; }
    // boundary strength (:1094-1360): 2 next to an intra (or CIIP) block, 1 next to a coded residual, else by prediction mode and motion
    const bool ciip = ( ( CQ.pred_mode == VVR_PRED_INTER && ( CQ.flags & VVR_CU_CIIP ) ) || ( CP.pred_mode == VVR_PRED_INTER && ( CP.flags & VVR_CU_CIIP ) ) );
    if( CQ.pred_mode == VVR_PRED_INTRA || CP.pred_mode == VVR_PRED_INTRA || ciip ) bsY = ( CQ.bdpcm[0] && CP.bdpcm[0] && CQ.pred_mode == VVR_PRED_INTRA && CP.pred_mode == VVR_PRED_INTRA ) ? 0 : 2;
    else if( ( TQ.cbf & 1 ) || ( TP.cbf & 1 ) ) bsY = 1;
    else if( TQ.cu != TP.cu )
    {
      if( CQ.pred_mode != CP.pred_mode ) bsY = 1;
      else if( CQ.pred_mode == VVR_PRED_IBC ) bsY = lfi_far( CQ.mv[0][0], CP.mv[0][0] ) ? 1 : 0;      // two block vectors into the same picture (:1346-1360)
      else bsY = lfi_motion_bs( lfi_motion( V, x4, y4, CQ ), lfi_motion( V, px4, py4, CP ) );
    }
    qpY = ( CQ.qp + CP.qp + 1 ) >> 1;
  }
  bool large = false; int qpCb = 0, qpCr = 0;
  if( h.chroma_format && ( posPerp & 15 ) == 0 )
  {
    // ---- chroma edges lie on the 8x8 chroma sample grid; the blocks that own the chroma on either side: the chroma-tree CU where the luma CU has none
    const int tqc = CQ.tree == VVR_TREE_LUMA ? lfi_idx( V.tuOf4C[iq], V.numTu ) : tq, tpc = CP.tree == VVR_TREE_LUMA ? lfi_idx( V.tuOf4C[ip], V.numTu ) : tp;
    if( tqc != tpc )
    {
      const vvr_cu& CQc = V.cu[lfi_idx( (int) V.tu[tqc].cu, V.numCu )]; const vvr_cu& CPc = V.cu[lfi_idx( (int) V.tu[tpc].cu, V.numCu )];
      // (the unsplit chroma blocks of an ISP CU - their coded block flags, their QPs - belong to its last transform unit, :1121-1123)
      const vvr_tu& TQc = V.tu[CQc.isp_mode ? lfi_idx( (int) ( CQc.first_tu + CQc.num_tu ) - 1, V.numTu ) : tqc];
      const vvr_tu& TPc = V.tu[CPc.isp_mode ? lfi_idx( (int) ( CPc.first_tu + CPc.num_tu ) - 1, V.numTu ) : tpc];
      if( !( V.tu[tqc].cu == V.tu[tpc].cu && CQc.isp_mode ) )              // (the chroma block of an ISP CU is not split)
      {
        const int sizeQc = ( CQc.isp_mode ? ( d == 0 ? CQc.w : CQc.h ) : ( d == 0 ? TQc.w : TQc.h ) ) >> 1, sizePc = ( CPc.isp_mode ? ( d == 0 ? CPc.w : CPc.h ) : ( d == 0 ? TPc.w : TPc.h ) ) >> 1;
        large = sizePc >= 8 && sizeQc >= 8;
        const bool ciipC = ( ( CQc.pred_mode == VVR_PRED_INTER && ( CQc.flags & VVR_CU_CIIP ) ) || ( CPc.pred_mode == VVR_PRED_INTER && ( CPc.flags & VVR_CU_CIIP ) ) );
        if( CQc.pred_mode == VVR_PRED_INTRA || CPc.pred_mode == VVR_PRED_INTRA || ciipC )
          bsCb = bsCr = ( CQc.pred_mode == VVR_PRED_INTRA && CQc.bdpcm[1] && CPc.pred_mode == VVR_PRED_INTRA && CPc.bdpcm[1] ) ? 0 : 2;      // (:1132)
        else
        {
          const bool joint = TQc.joint_cbcr || TPc.joint_cbcr;      // (:1180-1184)
          bsCb = ( ( TQc.cbf & 2 ) || ( TPc.cbf & 2 ) || joint ) ? 1 : 0;
          bsCr = ( ( TQc.cbf & 4 ) || ( TPc.cbf & 4 ) || joint ) ? 1 : 0;
        }
        const int qpBd2 = 12 * ( h.bit_depth - 8 );
        qpCb = ( TQc.qp[1] + TPc.qp[1] - qpBd2 + 1 ) >> 1;
        qpCr = ( TQc.qp[2] + TPc.qp[2] - qpBd2 + 1 ) >> 1;
      }
    }
  }
  if( !open ) bsY = bsCb = bsCr = 0;
  // ---- sub-block edges of affine and SbTMVP CUs: the 8x8 grid inside the CU (xSetEdgeFilterInsidePu :1032, xSetMaxFilterLengthPQForCodingSubBlocks :707).
  // Luma only; strength from the motion of the two sub-blocks, lengths limited by the distance to the next transform edge
  if( CQ.pred_mode == VVR_PRED_INTER && ( CQ.flags & ( VVR_CU_AFFINE | VVR_CU_SBTMVP ) ) && CQ.tree != VVR_TREE_CHROMA )
  {
    const int perp = d == 0 ? CQ.w : CQ.h, pp = posPerp - ( d == 0 ? CQ.x : CQ.y );
    if( ( pp & 7 ) == 0 )
    {
      // is the cell k cells further along the perpendicular direction (inside this CU) at a transform edge?
      auto isTe = [&]( int k ) -> bool
      {
        const int q = pp + 4 * k;
        if( q < 0 || q >= perp || posPerp + 4 * k == 0 ) return false;
        return V.tuOf4[iq + k * step] != V.tuOf4[iq + ( k - 1 ) * step];
      };
      if( te )
      {
        if( lenQ > 5 ) lenQ = 5;
        if( pp > 0 )
        {
          if( lenP > 5 ) lenP = 5;
          // a transform edge inside the CU that is a sub-block edge too: without a coded block on either side the motion decides
          if( !onVb && !bsY ) bsY = lfi_motion_bs( lfi_motion( V, x4, y4, CQ ), lfi_motion( V, px4, py4, CQ ) );
        }
      }
      else
      {
        if( isTe( -1 ) || pp + 4 >= perp || isTe( 1 ) ) lenP = lenQ = 1;
        else if( pp == 8 || isTe( -2 ) || pp + 8 >= perp || isTe( 2 ) ) lenP = lenQ = 2;
        else lenP = lenQ = 3;
        if( !onVb ) { bsY = lfi_motion_bs( lfi_motion( V, x4, y4, CQ ), lfi_motion( V, px4, py4, CQ ) ); qpY = CQ.qp; }
      }
    }
  }
  L.bs = (uint8_t) ( bsY | ( bsCb << 2 ) | ( bsCr << 4 ) );
  if( bsY ) { L.qp[0] = (int8_t) qpY; L.side_max_filt_length = (uint8_t) ( ( lenP << 4 ) | lenQ ); L.flags |= 1; }
  if( te ) L.side_max_filt_length |= 0x80;
  if( bsCb | bsCr ) { L.qp[1] = (int8_t) qpCb; L.qp[2] = (int8_t) qpCr; L.flags |= (uint8_t) ( 2 | ( large ? 0x20 : 0 ) ); }
  return L;
}
