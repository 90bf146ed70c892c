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

// One 16-byte record per 4x4 cell and tree: everything the derivation reads of the transform unit and the CU that cover the cell (k_lf_maps writes it from
// the records, one thread per transform unit; k_lf_init then reads two records per edge instead of chasing cell -> transform unit -> CU through HBM)
struct LfCell { uint32_t a, b, c, d; };
//  a: transform unit << 4 | LFI_CELL_TREE_L | LFI_CELL_SUB          (the word that says whether two cells are separated by an edge at all)
//  b: CU (22 bits) | pred_mode << 22 | CIIP << 24 | luma BDPCM << 25 | chroma BDPCM << 26 | ISP << 27 | cbf Y / Cb / Cr << 28 | joint Cb-Cr << 31
//  c: transform unit width (7 bits) | height << 7 | log2 CU width << 14 | log2 CU height << 17 | cell column inside the CU << 20 | cell row inside the CU << 25
//  d: CU QP (8 bits) | Cb QP << 8 | Cr QP << 16                     (chroma QPs and chroma cbf of an ISP CU: of its last transform unit, LoopFilter.cpp:1121-1123)
#define LFI_CELL_TREE_L 1      /* the CU belongs to a luma-only tree: the chroma of the cell is in the chroma tree's record */
#define LFI_CELL_SUB    2      /* the CU has sub-block edges and counts as "affine" on the P side of an edge (affine and SbTMVP CUs: LoopFilter.cpp:535,920) */
VVR_HD uint32_t lfc_tu( const LfCell& q )      { return q.a >> 4; }
VVR_HD uint32_t lfc_cu( const LfCell& q )      { return q.b & 0x3fffffu; }
VVR_HD int      lfc_pred( const LfCell& q )    { return (int) ( ( q.b >> 22 ) & 3 ); }
VVR_HD bool     lfc_ciip( const LfCell& q )    { return ( q.b >> 24 ) & 1; }
VVR_HD bool     lfc_bdpcm( const LfCell& q, int chroma ) { return ( q.b >> ( 25 + chroma ) ) & 1; }
VVR_HD bool     lfc_isp( const LfCell& q )     { return ( q.b >> 27 ) & 1; }
VVR_HD int      lfc_cbf( const LfCell& q )     { return (int) ( ( q.b >> 28 ) & 7 ); }
VVR_HD bool     lfc_joint( const LfCell& q )   { return ( q.b >> 31 ) & 1; }
VVR_HD int      lfc_tu_size( const LfCell& q, int d ) { return (int) ( ( q.c >> ( d == 0 ? 0 : 7 ) ) & 127 ); }
VVR_HD int      lfc_cu_size( const LfCell& q, int d ) { return 1 << ( ( q.c >> ( d == 0 ? 14 : 17 ) ) & 7 ); }
VVR_HD int      lfc_off( const LfCell& q, int d )     { return (int) ( ( q.c >> ( d == 0 ? 20 : 25 ) ) & 31 ); }
VVR_HD int      lfc_qp( const LfCell& q, int comp )   { return (int) (int8_t) ( ( q.d >> ( 8 * comp ) ) & 0xff ); }

struct LfMv { int32_t v[2][2]; };
VVR_HD uint32_t lfi_pack_refs( const vvr_motion& m ) { return (uint32_t) (uint8_t) m.ref_idx[0] | ( (uint32_t) (uint8_t) m.ref_idx[1] << 8 ); }
VVR_HD LfMv lfi_pack_mv( const vvr_motion& m ) { LfMv r; r.v[0][0] = m.mv[0][0]; r.v[0][1] = m.mv[0][1]; r.v[1][0] = m.mv[1][0]; r.v[1][1] = m.mv[1][1]; return r; }
struct LfInitView {
  const vvr_pic_header*   hdr;
  const LfCell*           cell;        // luma tree (or joint tree)
  const LfCell*           cellC;       // chroma tree: valid where the cell's luma CU belongs to a luma-only tree (dual tree, local dual tree)
  const LfMv*             mv;          // motion of the cells of inter and IBC CUs (the deblocking filter's view: unrefined), picture raster: the vectors (one 16-byte store) ...
  const uint32_t*         ref;         // ... and the reference indices, (uint8_t) ref_idx[0] | (uint8_t) ref_idx[1] << 8
  const uint16_t*         ctuSlice;    // NULL: one slice
  const uint16_t*         ctuTile;     // NULL: one tile
  const uint16_t*         ctuSubpic;   // NULL: one sub-picture
  const vvr_subpic*       subpics;
  const vvr_slice_header* slices;      // NULL: the picture header's values
  int                     w4, h4, ctusX;
};

// one cell of a CU whose motion varies inside the CU, as the host hands it over (SbTMVP and GPM CUs; affine CUs too unless the back-end spans their
// sub-block vectors itself, VVR_TOOL_AFFINE_MV_ON_DEVICE): scattered into LfInitView::mv / ref
struct LfSbCell { uint32_t cell; vvr_motion m; };

VVR_HD int lfi_idx( int i, int n ) { return i < 0 ? 0 : i >= n ? n - 1 : i; }
VVR_HD int lfi_ilog2( int v ) { int l = 0; while( ( 1 << l ) < v ) l++; return l; }
VVR_HD int lfi_max( int a, int b ) { return a > b ? a : b; }
VVR_HD int lfi_min( int a, int b ) { return a < b ? a : b; }

// does a transform unit own the cells it touches?  Partitions narrower (lower) than a cell share it: the last one in decoding order - the one that ends
// on the cell's far side - is what a later look-up of the cell finds (the reference's edge pass visits the partitions in order, LoopFilter.cpp:543-567)
VVR_HD bool lfi_tu_owns_cells( const vvr_tu& t ) { return !( ( t.w < 4 && ( ( t.x + t.w ) & 3 ) ) || ( t.h < 4 && ( ( t.y + t.h ) & 3 ) ) ); }

// the record of the cells of transform unit t (index tuIdx) of CU c; last = the CU's last transform unit; the cell's place inside the CU is added per cell (lfi_cell_at)
VVR_HD LfCell lfi_pack_cell( const vvr_tu& t, int tuIdx, const vvr_cu& c, const vvr_tu& last )
{
  LfCell q;
  const bool sub = c.pred_mode == VVR_PRED_INTER && ( c.flags & ( VVR_CU_AFFINE | VVR_CU_SBTMVP ) ) && c.tree != VVR_TREE_CHROMA;
  const vvr_tu& tc = c.isp_mode ? last : t;         // (who holds the chroma of the block)
  q.a = ( (uint32_t) tuIdx << 4 ) | ( c.tree == VVR_TREE_LUMA ? LFI_CELL_TREE_L : 0 ) | ( sub ? LFI_CELL_SUB : 0 );
  q.b = ( t.cu & 0x3fffffu ) | ( (uint32_t) ( c.pred_mode & 3 ) << 22 ) | ( c.pred_mode == VVR_PRED_INTER && ( c.flags & VVR_CU_CIIP ) ? 1u << 24 : 0 )
      | ( c.pred_mode == VVR_PRED_INTRA && c.bdpcm[0] ? 1u << 25 : 0 ) | ( c.pred_mode == VVR_PRED_INTRA && c.bdpcm[1] ? 1u << 26 : 0 ) | ( c.isp_mode ? 1u << 27 : 0 )
      | ( (uint32_t) ( ( t.cbf & 1 ) | ( tc.cbf & 6 ) ) << 28 ) | ( tc.joint_cbcr ? 1u << 31 : 0 );
  q.c = (uint32_t) ( t.w & 127 ) | ( (uint32_t) ( t.h & 127 ) << 7 ) | ( (uint32_t) lfi_ilog2( c.w ) << 14 ) | ( (uint32_t) lfi_ilog2( c.h ) << 17 );
  q.d = (uint32_t) (uint8_t) c.qp | ( (uint32_t) (uint8_t) tc.qp[1] << 8 ) | ( (uint32_t) (uint8_t) tc.qp[2] << 16 );
  return q;
}
VVR_HD LfCell lfi_cell_at( LfCell q, int cuX4, int cuY4, int x4, int y4 ) { q.c |= ( (uint32_t) ( ( x4 - cuX4 ) & 31 ) << 20 ) | ( (uint32_t) ( ( y4 - cuY4 ) & 31 ) << 25 ); return q; }

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

// what the motion field holds for cell (x4, y4) of CU c, as far as the CU's record says: 0 = nothing to write (an intra CU, a chroma-tree CU), 1 = m is the motion,
// 2 = the motion of this CU varies inside it and comes from the host's list (SbTMVP, GPM; affine without VVR_TOOL_AFFINE_MV_ON_DEVICE)
VVR_HD int lfi_cell_motion( const vvr_pic_header& h, const vvr_cu& c, int x4, int y4, vvr_motion& m )
{
  if( c.tree == VVR_TREE_CHROMA || c.pred_mode == VVR_PRED_INTRA ) return 0;
  for( int l = 0; l < 2; l++ ) { m.mv[l][0] = m.mv[l][1] = 0; m.ref_idx[l] = -1; }
  m.pad[0] = m.pad[1] = 0;
  if( c.pred_mode == VVR_PRED_IBC ) { m.mv[0][0] = c.mv[0][0][0]; m.mv[0][1] = c.mv[0][0][1]; return 1; }      // the block vector, no reference picture (UnitTools.cpp:3018)
  const bool affine = ( c.flags & VVR_CU_AFFINE ) != 0;
  if( ( c.flags & ( VVR_CU_SBTMVP | VVR_CU_GEO ) ) || ( affine && !( h.tool_flags & VVR_TOOL_AFFINE_MV_ON_DEVICE ) ) ) return 2;
  for( int l = 0; l < 2; l++ )
  {
    if( c.ref_idx[l] < 0 ) continue;
    m.ref_idx[l] = c.ref_idx[l];
    if( affine ) lfi_affine_mv( c, l, x4 - ( c.x >> 2 ), y4 - ( c.y >> 2 ), m.mv[l] );
    else { m.mv[l][0] = c.mv[l][0][0]; m.mv[l][1] = c.mv[l][0][1]; }
  }
  return 1;
}

struct LfiMotion { int32_t mv[2][2]; int32_t poc[2]; };      // poc: the reference picture of the list, INT32_MIN = list not used
VVR_HD LfiMotion lfi_motion( const LfInitView& V, int cell )
{
  LfiMotion m;
  const LfMv s = V.mv[cell]; const uint32_t r = V.ref[cell];
  for( int l = 0; l < 2; l++ ) { const int ri = (int8_t) ( ( r >> ( 8 * l ) ) & 0xff ); m.mv[l][0] = s.v[l][0]; m.mv[l][1] = s.v[l][1]; m.poc[l] = ri >= 0 ? V.hdr->ref_poc[l][ri & ( VVR_MAX_REFS - 1 )] : INT32_MIN; }
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

// the table entry of cell (x4, y4) for the edges of direction d (0: the cell's left edge, vertical edges; 1: its top edge); Q = the cell's record
// (P: the record of the cell on the other side of the edge - the cell before this one in direction d; the device kernel reads it together with Q and the
// other direction's, before anything is decided: one memory round trip for the three of them.  Not looked at on the picture boundary.)
VVR_HD vvr_lfp lf_init_cell( const LfInitView& V, int d, int x4, int y4, const LfCell& Q, const LfCell& P )
{
  vvr_lfp L; L.qp[0] = L.qp[1] = L.qp[2] = 0; L.bs = 0; L.side_max_filt_length = 0; L.flags = 0; L.pad[0] = L.pad[1] = 0;
  const int px4 = d == 0 ? x4 - 1 : x4, py4 = d == 0 ? y4 : y4 - 1;
  if( px4 < 0 || py4 < 0 ) return L;                          // picture boundary
  const vvr_pic_header& h = *V.hdr;
  const int step = d == 0 ? 1 : V.w4;
  const int iq = y4 * V.w4 + x4, ip = iq - step;
  const int posPerp = ( d == 0 ? x4 : y4 ) << 2;
  // nothing to derive where both cells lie in one transform unit of a CU without sub-block edges and without a chroma tree of its own (a chroma edge inside a luma
  // transform unit needs one): nine cells in ten
  const bool chromaGrid = h.chroma_format && ( posPerp & 15 ) == 0;
  if( lfc_tu( Q ) == lfc_tu( P ) && !( Q.a & LFI_CELL_SUB ) && !( ( Q.a & LFI_CELL_TREE_L ) && chromaGrid ) ) return L;
  const int l2c = h.log2_ctu - 2;
  const int ctuQ = ( y4 >> l2c ) * V.ctusX + ( x4 >> l2c ), ctuP = ( py4 >> l2c ) * V.ctusX + ( px4 >> l2c );
  // a slice that switches deblocking off in a picture that deblocks: the edges of its CTUs are left alone (LoopFilter.cpp:366,423)
  if( V.slices && V.ctuSlice && ( V.slices[V.ctuSlice[ctuQ]].tool_flags & VVR_TOOL_DEBLOCK_OFF ) ) return L;
  const bool onVb = lfi_on_virtual_boundary( h, d, x4, y4 );
  const bool open = !onVb && lfi_may_cross( V, ctuQ, ctuP );    // the edge may be filtered at all
  int bsY = 0, bsCb = 0, bsCr = 0, lenP = 0, lenQ = 0, qpY = 0;
  bool te = false;
  if( lfc_tu( Q ) != lfc_tu( P ) )
  {
    // ---- an edge of the luma transform grid: filter lengths from the transform sizes across the edge (:905-922)
    te = true;
    const int sizeQ = lfc_tu_size( Q, d ), sizeP = lfc_tu_size( P, d );
    if( sizeP <= 4 || sizeQ <= 4 ) lenP = lenQ = 1;
    // (:920 cuP->affineFlag(): a CU in sub-block merge mode carries that flag whether its candidate was an affine one or the SbTMVP one, DecCu.cpp:746-767)
    else { lenP = sizeP >= 32 ? ( ( P.a & LFI_CELL_SUB ) ? 5 : 7 ) : 3; lenQ = sizeQ >= 32 ? 7 : 3; }
    // boundary strength (:1094-1360): 2 next to an intra (or CIIP) block, 1 next to a coded residual, else by prediction mode and motion
    const int pmQ = lfc_pred( Q ), pmP = lfc_pred( P );
    if( pmQ == VVR_PRED_INTRA || pmP == VVR_PRED_INTRA || lfc_ciip( Q ) || lfc_ciip( P ) ) bsY = ( lfc_bdpcm( Q, 0 ) && lfc_bdpcm( P, 0 ) ) ? 0 : 2;      // (:1146; the bit is set for intra CUs only)
    else if( ( lfc_cbf( Q ) | lfc_cbf( P ) ) & 1 ) bsY = 1;
    else if( lfc_cu( Q ) != lfc_cu( P ) )
    {
      if( pmQ != pmP ) bsY = 1;
      else if( pmQ == VVR_PRED_IBC ) bsY = lfi_far( V.mv[iq].v[0], V.mv[ip].v[0] ) ? 1 : 0;      // two block vectors into the same picture (:1346-1360)
      else bsY = lfi_motion_bs( lfi_motion( V, iq ), lfi_motion( V, ip ) );
    }
    qpY = ( lfc_qp( Q, 0 ) + lfc_qp( P, 0 ) + 1 ) >> 1;
  }
  bool large = false; int qpCb = 0, qpCr = 0;
  if( chromaGrid )
  {
    // ---- chroma edges lie on the 8x8 chroma sample grid; the blocks that own the chroma on either side: the chroma-tree CU where the luma CU has none
    const LfCell Qc = ( Q.a & LFI_CELL_TREE_L ) ? V.cellC[iq] : Q, Pc = ( P.a & LFI_CELL_TREE_L ) ? V.cellC[ip] : P;
    if( lfc_tu( Qc ) != lfc_tu( Pc ) && !( lfc_cu( Qc ) == lfc_cu( Pc ) && lfc_isp( Qc ) ) )              // (the chroma block of an ISP CU is not split)
    {
      const int sizeQc = ( lfc_isp( Qc ) ? lfc_cu_size( Qc, d ) : lfc_tu_size( Qc, d ) ) >> 1, sizePc = ( lfc_isp( Pc ) ? lfc_cu_size( Pc, d ) : lfc_tu_size( Pc, d ) ) >> 1;
      large = sizePc >= 8 && sizeQc >= 8;
      if( lfc_pred( Qc ) == VVR_PRED_INTRA || lfc_pred( Pc ) == VVR_PRED_INTRA || lfc_ciip( Qc ) || lfc_ciip( Pc ) ) bsCb = bsCr = ( lfc_bdpcm( Qc, 1 ) && lfc_bdpcm( Pc, 1 ) ) ? 0 : 2;      // (:1132)
      else
      {
        const int cbf = lfc_cbf( Qc ) | lfc_cbf( Pc ); const bool joint = lfc_joint( Qc ) || lfc_joint( Pc );      // (:1180-1184)
        bsCb = ( ( cbf & 2 ) || joint ) ? 1 : 0;
        bsCr = ( ( cbf & 4 ) || joint ) ? 1 : 0;
      }
      const int qpBd2 = 12 * ( h.bit_depth - 8 );
      qpCb = ( lfc_qp( Qc, 1 ) + lfc_qp( Pc, 1 ) - qpBd2 + 1 ) >> 1;
      qpCr = ( lfc_qp( Qc, 2 ) + lfc_qp( Pc, 2 ) - qpBd2 + 1 ) >> 1;
    }
  }
  if( !open ) bsY = bsCb = bsCr = 0;
  // ---- sub-block edges of affine and SbTMVP CUs: the 8x8 grid inside the CU (xSetEdgeFilterInsidePu :1032, xSetMaxFilterLengthPQForCodingSubBlocks :707).
  // Luma only; strength from the motion of the two sub-blocks, lengths limited by the distance to the next transform edge
  if( Q.a & LFI_CELL_SUB )
  {
    const int perp = lfc_cu_size( Q, d ), pp = lfc_off( Q, d ) << 2;
    if( ( pp & 7 ) == 0 )
    {
      // is the cell k cells further along the perpendicular direction (inside this CU) MARKED as a transform edge?  In a CU with sub-block edges the reference
      // writes marker and lengths of a transform edge only where that edge may be filtered (LoopFilter.cpp:960-981 under bValue): not on a virtual boundary, not on
      // a slice / tile / sub-picture boundary the filter may not cross (the CU's own border).  Found by tools/fuzz_lf_init.py.
      auto isTe = [&]( int k ) -> bool
      {
        const int q = pp + 4 * k;
        if( q < 0 || q >= perp || posPerp + 4 * k == 0 ) return false;
        if( ( V.cell[iq + k * step].a >> 4 ) == ( V.cell[iq + ( k - 1 ) * step].a >> 4 ) ) return false;
        const int cx = d == 0 ? x4 + k : x4, cy = d == 0 ? y4 : y4 + k;
        if( lfi_on_virtual_boundary( h, d, cx, cy ) ) return false;
        if( q > 0 ) return true;
        const int l2 = h.log2_ctu - 2, qx = d == 0 ? cx - 1 : cx, qy = d == 0 ? cy : cy - 1;
        return lfi_may_cross( V, ( cy >> l2 ) * V.ctusX + ( cx >> l2 ), ( qy >> l2 ) * V.ctusX + ( qx >> l2 ) );
      };
      if( te )
      {
        if( lenQ > 5 ) lenQ = 5;
        if( pp > 0 )
        {
          if( lenP > 5 ) lenP = 5;
          // a transform edge inside the CU that is a sub-block edge too: without a coded block on either side the motion decides
          if( !onVb && !bsY ) bsY = lfi_motion_bs( lfi_motion( V, iq ), lfi_motion( V, ip ) );
        }
      }
      else
      {
        if( isTe( -1 ) || pp + 4 >= perp || isTe( 1 ) ) lenP = lenQ = 1;
        else if( pp == 8 || isTe( -2 ) || pp + 8 >= perp || isTe( 2 ) ) lenP = lenQ = 2;
        else lenP = lenQ = 3;
        if( !onVb ) { bsY = lfi_motion_bs( lfi_motion( V, iq ), lfi_motion( V, ip ) ); qpY = lfc_qp( Q, 0 ); }
      }
    }
  }
  L.bs = (uint8_t) ( bsY | ( bsCb << 2 ) | ( bsCr << 4 ) );
  if( bsY ) { L.qp[0] = (int8_t) qpY; L.side_max_filt_length = (uint8_t) ( ( lenP << 4 ) | lenQ ); L.flags |= 1; }
  if( te ) L.side_max_filt_length |= 0x80;
  if( bsCb | bsCr ) { L.qp[1] = (int8_t) qpCb; L.qp[2] = (int8_t) qpCr; L.flags |= (uint8_t) ( 2 | ( large ? 0x20 : 0 ) ); }
  return L;
}

VVR_HD vvr_lfp lf_init_cell( const LfInitView& V, int d, int x4, int y4, const LfCell& Q )
{
  const int px4 = d == 0 ? x4 - 1 : x4, py4 = d == 0 ? y4 : y4 - 1;
  const LfCell P = ( px4 < 0 || py4 < 0 ) ? Q : V.cell[(size_t) py4 * V.w4 + px4];
  return lf_init_cell( V, d, x4, y4, Q, P );
}

// ---- the two passes as plain loops (the tests' stand-in runtime, the drop-in's self-check): what k_lf_maps and k_lf_init do with one thread per transform unit / cell
inline void lf_init_maps_host( const vvr_pic_header& h, const vvr_cu* cu, uint32_t numCu, const vvr_tu* tu, uint32_t numTu, LfCell* cell, LfCell* cellC, LfMv* mv, uint32_t* ref, int w4, int h4 )
{
  for( uint32_t t = 0; t < numTu; t++ )
  {
    const vvr_tu& T = tu[t];
    if( !lfi_tu_owns_cells( T ) ) continue;
    const vvr_cu& C = cu[lfi_idx( (int) T.cu, (int) numCu )];
    if( C.tree == VVR_TREE_CHROMA && !cellC ) continue;
    const LfCell rec = lfi_pack_cell( T, (int) t, C, tu[lfi_idx( (int) ( C.first_tu + C.num_tu ) - 1, (int) numTu )] );
    const int x1 = lfi_min( ( T.x + T.w + 3 ) >> 2, w4 ), y1 = lfi_min( ( T.y + T.h + 3 ) >> 2, h4 );
    for( int y = T.y >> 2; y < y1; y++ ) for( int x = T.x >> 2; x < x1; x++ )
    {
      ( C.tree == VVR_TREE_CHROMA ? cellC : cell )[(size_t) y * w4 + x] = lfi_cell_at( rec, C.x >> 2, C.y >> 2, x, y );
      vvr_motion m;
      if( mv && lfi_cell_motion( h, C, x, y, m ) == 1 ) { mv[(size_t) y * w4 + x] = lfi_pack_mv( m ); ref[(size_t) y * w4 + x] = lfi_pack_refs( m ); }
    }
  }
}
inline void lf_init_tables_host( const LfInitView& V, vvr_lfp* out0, vvr_lfp* out1 )
{
  for( int y = 0; y < V.h4; y++ ) for( int x = 0; x < V.w4; x++ )
  {
    const LfCell Q = V.cell[(size_t) y * V.w4 + x];
    out0[(size_t) y * V.w4 + x] = lf_init_cell( V, 0, x, y, Q ); out1[(size_t) y * V.w4 + x] = lf_init_cell( V, 1, x, y, Q );
  }
}
