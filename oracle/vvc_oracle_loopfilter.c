/* oracle/vvc_oracle_loopfilter.c — CPU restatement (TEST INFRASTRUCTURE): deblocking, SAO, ALF, CC-ALF.
 *
 * Follows  CommonLib/LoopFilter.cpp:106-335 (filters), :419-493 (xDeblockCtuArea), :1391-1462 (decisions),
 *          :1464-1617 (xEdgeFilterLuma), :1620-1732 (xEdgeFilterChroma),
 *          CommonLib/SampleAdaptiveOffset.cpp:64-343 (offsetBlock_core), :548-572, :661-739,
 *          CommonLib/AdaptiveLoopFilter.cpp:453-463 (prepareCTU), :466-480, :498-610, :664-745 (filterCTU),
 *          :969-1174 (deriveClassificationBlk), :1176-1346 (filterBlk), :1348-1445 (filterBlkCcAlf). */
#include "vvc_oracle_common.h"
#include "../tables/vvc_tables.inc"

/* ================================================================================================================= */
/* deblocking                                                                                                        */
/* ================================================================================================================= */
#define BS_GET( v, c ) ( ( ( v ) >> ( ( c ) << 1 ) ) & 3 )

static int calc_dp( const pel* s, ptrdiff_t o ) { return vvo_abs( s[-o * 3] - 2 * s[-o * 2] + s[-o] ); }          /* xCalcDP  (:1392) */
static int calc_dp_ctb( const pel* s, ptrdiff_t o ) { return vvo_abs( s[-o * 2] - 2 * s[-o * 2] + s[-o] ); }      /* <isChromaHorCTBBoundary> */
static int calc_dq( const pel* s, ptrdiff_t o ) { return vvo_abs( s[0] - 2 * s[o] + s[o * 2] ); }                  /* xCalcDQ  (:1404) */

static int use_strong( const pel* s, ptrdiff_t o, int d, int beta, int tc, int pLarge, int qLarge, int lenP, int lenQ, int chromaCtb )   /* xUseStrongFiltering (:1410) */
{
  const int m3 = s[-o], m4 = s[0];
  if( !( d < ( beta >> 2 ) && vvo_abs( m3 - m4 ) < ( ( tc * 5 + 1 ) >> 1 ) ) ) return 0;
  const int m0 = s[-4 * o], m7 = s[3 * o], m2 = s[-2 * o];
  int sp3 = vvo_abs( m0 - m3 );
  if( chromaCtb ) sp3 = vvo_abs( m2 - m3 );
  int sq3 = vvo_abs( m7 - m4 );
  const int d_strong = sp3 + sq3;
  if( pLarge || qLarge )
  {
    if( pLarge )
    {
      const int mP4 = s[-o * lenP - o];
      if( lenP == 7 ) sp3 = sp3 + vvo_abs( s[-o * 5] - s[-o * 6] - s[-o * 7] + mP4 );
      sp3 = ( sp3 + vvo_abs( m0 - mP4 ) + 1 ) >> 1;
    }
    if( qLarge )
    {
      const int m11 = s[o * lenQ];
      if( lenQ == 7 ) sq3 = sq3 + vvo_abs( s[o * 4] - s[o * 5] - s[o * 6] + m11 );
      sq3 = ( sq3 + vvo_abs( m11 - m7 ) + 1 ) >> 1;
    }
    return ( ( sp3 + sq3 ) < ( beta * 3 >> 5 ) ) && ( d < ( beta >> 4 ) ) && ( vvo_abs( m3 - m4 ) < ( ( tc * 5 + 1 ) >> 1 ) );
  }
  return d_strong < ( beta >> 3 );
}

static void filter_long( pel* src, ptrdiff_t step, ptrdiff_t o, int nP, int nQ, int tc )   /* xFilteringPandQCore (:129) + xBilinearFilter (:106) */
{
  static const int c7[7] = { 59, 50, 41, 32, 23, 14, 5 }, c5[5] = { 58, 45, 32, 19, 6 }, c3[3] = { 53, 32, 11 };
  static const int8_t tc7[7] = { 6, 5, 4, 3, 2, 1, 1 }, tc3[3] = { 6, 4, 2 };
  const int* cP = nP == 7 ? c7 : nP == 5 ? c5 : c3;
  const int* cQ = nQ == 7 ? c7 : nQ == 5 ? c5 : c3;
  for( int i = 0; i < 4; i++ )
  {
    pel* sP = src + step * i - o; pel* sQ = src + step * i;
    const int refP = ( sP[-( nP - 1 ) * o] + sP[-nP * o] + 1 ) >> 1;
    const int refQ = ( sQ[( nQ - 1 ) * o] + sQ[nQ * o] + 1 ) >> 1;
    int refM;
    if( nP == nQ )
    {
      if( nP == 5 ) refM = ( 2 * ( sP[0] + sQ[0] + sP[-o] + sQ[o] + sP[-2 * o] + sQ[2 * o] ) + sP[-3 * o] + sQ[3 * o] + sP[-4 * o] + sQ[4 * o] + 8 ) >> 4;
      else          refM = ( 2 * ( sP[0] + sQ[0] ) + sP[-o] + sQ[o] + sP[-2 * o] + sQ[2 * o] + sP[-3 * o] + sQ[3 * o] + sP[-4 * o] + sQ[4 * o] + sP[-5 * o] + sQ[5 * o] + sP[-6 * o] + sQ[6 * o] + 8 ) >> 4;
    }
    else
    {
      pel *pt = sP, *qt = sQ; ptrdiff_t oP = -o, oQ = o; int nnP = nP, nnQ = nQ;
      if( nQ > nP ) { pel* t = pt; pt = qt; qt = t; oP = o; oQ = -o; nnQ = nP; nnP = nQ; }
      if( nnP == 7 && nnQ == 5 ) refM = ( 2 * ( sP[0] + sQ[0] + sP[-o] + sQ[o] ) + sP[-2 * o] + sQ[2 * o] + sP[-3 * o] + sQ[3 * o] + sP[-4 * o] + sQ[4 * o] + sP[-5 * o] + sQ[5 * o] + 8 ) >> 4;
      else if( nnP == 7 && nnQ == 3 ) refM = ( 2 * ( pt[0] + qt[0] ) + qt[0] + 2 * ( qt[oQ] + qt[2 * oQ] ) + pt[oP] + qt[oQ] + pt[2 * oP] + pt[3 * oP] + pt[4 * oP] + pt[5 * oP] + pt[6 * oP] + 8 ) >> 4;
      else refM = ( sP[0] + sQ[0] + sP[-o] + sQ[o] + sP[-2 * o] + sQ[2 * o] + sP[-3 * o] + sQ[3 * o] + 4 ) >> 3;
    }
    const int8_t* tP = nP == 3 ? tc3 : tc7; const int8_t* tQ = nQ == 3 ? tc3 : tc7;
    for( int p = 0; p < nP; p++ ) { const int v = sP[-o * p], cv = ( tc * tP[p] ) >> 1; sP[-o * p] = (pel) vvo_clip3( v - cv, v + cv, ( refM * cP[p] + refP * ( 64 - cP[p] ) + 32 ) >> 6 ); }
    for( int p = 0; p < nQ; p++ ) { const int v = sQ[o * p],  cv = ( tc * tQ[p] ) >> 1; sQ[o * p]  = (pel) vvo_clip3( v - cv, v + cv, ( refM * cQ[p] + refQ * ( 64 - cQ[p] ) + 32 ) >> 6 ); }
  }
}

static void filter_luma_pel( pel* s, ptrdiff_t o, int tc, int sw, int thrCut, int fP, int fQ, int bd )   /* xPelFilterLumaCorePel (:213) */
{
  const int m1 = s[-3 * o], m2 = s[-2 * o], m3 = s[-o], m4 = s[0], m5 = s[o], m6 = s[2 * o];
  if( sw )
  {
    const int m0 = s[-4 * o], m7 = s[3 * o];
    s[-3 * o] = (pel) vvo_clip3( m1 - 1 * tc, m1 + 1 * tc, ( 2 * m0 + 3 * m1 + m2 + m3 + m4 + 4 ) >> 3 );
    s[-2 * o] = (pel) vvo_clip3( m2 - 2 * tc, m2 + 2 * tc, ( m1 + m2 + m3 + m4 + 2 ) >> 2 );
    s[-1 * o] = (pel) vvo_clip3( m3 - 3 * tc, m3 + 3 * tc, ( m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4 ) >> 3 );
    s[0]      = (pel) vvo_clip3( m4 - 3 * tc, m4 + 3 * tc, ( m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4 ) >> 3 );
    s[o]      = (pel) vvo_clip3( m5 - 2 * tc, m5 + 2 * tc, ( m3 + m4 + m5 + m6 + 2 ) >> 2 );
    s[2 * o]  = (pel) vvo_clip3( m6 - 1 * tc, m6 + 1 * tc, ( m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4 ) >> 3 );
  }
  else
  {
    int delta = ( 9 * ( m4 - m3 ) - 3 * ( m5 - m2 ) + 8 ) >> 4;
    if( vvo_abs( delta ) < thrCut )
    {
      delta = vvo_clip3( -tc, tc, delta );
      const int tc2 = tc >> 1;
      s[-o] = (pel) vvo_clip_pel( m3 + delta, bd );
      if( fP ) s[-2 * o] = (pel) vvo_clip_pel( m2 + vvo_clip3( -tc2, tc2, ( ( ( m1 + m3 + 1 ) >> 1 ) - m2 + delta ) >> 1 ), bd );
      s[0] = (pel) vvo_clip_pel( m4 - delta, bd );
      if( fQ ) s[o] = (pel) vvo_clip_pel( m5 + vvo_clip3( -tc2, tc2, ( ( ( m6 + m4 + 1 ) >> 1 ) - m5 - delta ) >> 1 ), bd );
    }
  }
}

static void filter_chroma_pel( pel* s, ptrdiff_t o, int tc, int sw, int bd, int ctb )   /* xPelFilterChroma (:281) */
{
  const int m2 = s[-2 * o], m3 = s[-o], m4 = s[0], m5 = s[o];
  if( sw )
  {
    const int m6 = s[2 * o], m7 = s[3 * o];
    if( ctb )
    {
      s[-o]    = (pel) vvo_clip3( m3 - tc, m3 + tc, ( 3 * m2 + 2 * m3 + m4 + m5 + m6 + 4 ) >> 3 );
      s[0]     = (pel) vvo_clip3( m4 - tc, m4 + tc, ( 2 * m2 + m3 + 2 * m4 + m5 + m6 + m7 + 4 ) >> 3 );
      s[o]     = (pel) vvo_clip3( m5 - tc, m5 + tc, ( m2 + m3 + m4 + 2 * m5 + m6 + 2 * m7 + 4 ) >> 3 );
      s[2 * o] = (pel) vvo_clip3( m6 - tc, m6 + tc, ( m3 + m4 + m5 + 2 * m6 + 3 * m7 + 4 ) >> 3 );
    }
    else
    {
      const int m0 = s[-4 * o], m1 = s[-3 * o];
      s[-3 * o] = (pel) vvo_clip3( m1 - tc, m1 + tc, ( 3 * m0 + 2 * m1 + m2 + m3 + m4 + 4 ) >> 3 );
      s[-2 * o] = (pel) vvo_clip3( m2 - tc, m2 + tc, ( 2 * m0 + m1 + 2 * m2 + m3 + m4 + m5 + 4 ) >> 3 );
      s[-o]     = (pel) vvo_clip3( m3 - tc, m3 + tc, ( m0 + m1 + m2 + 2 * m3 + m4 + m5 + m6 + 4 ) >> 3 );
      s[0]      = (pel) vvo_clip3( m4 - tc, m4 + tc, ( m1 + m2 + m3 + 2 * m4 + m5 + m6 + m7 + 4 ) >> 3 );
      s[o]      = (pel) vvo_clip3( m5 - tc, m5 + tc, ( m2 + m3 + m4 + 2 * m5 + m6 + 2 * m7 + 4 ) >> 3 );
      s[2 * o]  = (pel) vvo_clip3( m6 - tc, m6 + tc, ( m3 + m4 + m5 + 2 * m6 + 3 * m7 + 4 ) >> 3 );
    }
  }
  else
  {
    const int delta = vvo_clip3( -tc, tc, ( ( ( m4 - m3 ) * 4 ) + m2 - m5 + 4 ) >> 3 );
    s[-o] = (pel) vvo_clip_pel( m3 + delta, bd );
    s[0]  = (pel) vvo_clip_pel( m4 - delta, bd );
  }
}

static int tc_value( int idx, int bd ) { const int t = vvc_db_tc_table[idx]; return bd < 10 ? ( t + ( 1 << ( 9 - bd ) ) ) >> ( 10 - bd ) : t << ( bd - 10 ); }

static void edge_luma( const vvr_pic_header* H, vvo_planes* r, int x, int y, const vvr_lfp* lfp, int dir )   /* xEdgeFilterLuma (:1464) */
{
  const int bd = H->bit_depth, stride = r->stride[0];
  pel* src = r->p[0] + (size_t) y * stride + x;
  const ptrdiff_t o = dir == 0 ? 1 : stride, step = dir == 0 ? stride : 1;
  const int bs = BS_GET( lfp->bs, 0 );
  if( !bs ) return;
  int qp = lfp->qp[0];
  if( H->ladf_num_intervals )
  {
    /* deriveLADFShift (LoopFilter.cpp:1363-1386): EDGE_VER (src[0] + src[3*stride] + src[-1] + src[3*stride - 1]) >> 2, EDGE_HOR likewise with the roles swapped */
    const int level = ( src[0] + src[3 * step] + src[-o] + src[3 * step - o] ) >> 2;
    int shift = H->ladf_qp_offset[0];
    for( int k = 1; k < H->ladf_num_intervals; k++ ) { if( level > H->ladf_lower_bound[k] ) shift = H->ladf_qp_offset[k]; else break; }
    qp += shift;
  }
  const int lenP = ( lfp->side_max_filt_length >> 4 ) & 7, lenQ = lfp->side_max_filt_length & 7;
  int pLarge = lenP > 3, qLarge = lenQ > 3;
  if( dir == 1 && ( y & ( ( 1 << H->log2_ctu ) - 1 ) ) == 0 ) pLarge = 0;
  const int idxTC = vvo_clip3( 0, 65, qp + 2 * ( bs - 1 ) + 2 * H->deblock_tc_offset_div2[0] );
  const int idxB  = vvo_clip3( 0, 63, qp + 2 * H->deblock_beta_offset_div2[0] );
  const int tc = tc_value( idxTC, bd ), beta = vvc_db_beta_table[idxB] << ( bd - 8 );
  const int sideThr = ( beta + ( beta >> 1 ) ) >> 3, thrCut = tc * 10;
  const pel* s0 = src; const pel* s3 = src + 3 * step;
  const int dp0 = calc_dp( s0, o ), dq0 = calc_dq( s0, o ), dp3 = calc_dp( s3, o ), dq3 = calc_dq( s3, o );
  const int d0 = dp0 + dq0, d3 = dp3 + dq3;
  if( pLarge || qLarge )
  {
    const ptrdiff_t o3 = 3 * o;
    const int dp0L = pLarge ? ( dp0 + calc_dp( s0 - o3, o ) + 1 ) >> 1 : dp0;
    const int dq0L = qLarge ? ( dq0 + calc_dq( s0 + o3, o ) + 1 ) >> 1 : dq0;
    const int dp3L = pLarge ? ( dp3 + calc_dp( s3 - o3, o ) + 1 ) >> 1 : dp3;
    const int dq3L = qLarge ? ( dq3 + calc_dq( s3 + o3, o ) + 1 ) >> 1 : dq3;
    const int d0L = dp0L + dq0L, d3L = dp3L + dq3L, dL = d0L + d3L;
    if( dL < beta )
    {
      const int swL = use_strong( s0, o, 2 * d0L, beta, tc, pLarge, qLarge, lenP, lenQ, 0 ) && use_strong( s3, o, 2 * d3L, beta, tc, pLarge, qLarge, lenP, lenQ, 0 );
      if( swL ) { filter_long( src, step, o, pLarge ? lenP : 3, qLarge ? lenQ : 3, tc ); return; }
    }
  }
  {
    const int dp = dp0 + dp3, dq = dq0 + dq3, d = d0 + d3;
    if( d < beta )
    {
      int fP = 0, fQ = 0, sw = 0;
      if( lenP > 1 && lenQ > 1 ) { fP = dp < sideThr; fQ = dq < sideThr; }
      if( lenP > 2 && lenQ > 2 ) sw = use_strong( s0, o, 2 * d0, beta, tc, 0, 0, 7, 7, 0 ) && use_strong( s3, o, 2 * d3, beta, tc, 0, 0, 7, 7, 0 );
      for( int i = 0; i < 4; i++ ) filter_luma_pel( src + step * i, o, tc, sw, thrCut, fP, fQ, bd );
    }
  }
}

static void edge_chroma( const vvr_pic_header* H, vvo_planes* r, int cx, int cy, const vvr_lfp* lfp, int dir )   /* xEdgeFilterChroma (:1620), 4:2:0 */
{
  const int bd = H->bit_depth, stride = r->stride[1];
  const ptrdiff_t o = dir == 0 ? 1 : stride, step = dir == 0 ? stride : 1;
  const int loopLen = 2;                                       /* minCU (4) >> chroma scale */
  const int bS[2] = { BS_GET( lfp->bs, 1 ), BS_GET( lfp->bs, 2 ) };
  if( !bS[0] && !bS[1] ) return;
  const int large = ( lfp->flags >> 5 ) & 1;
  const int ctb = dir == 1 && ( cy & ( ( ( 1 << H->log2_ctu ) - 1 ) >> 1 ) ) == 0;
  for( int c = 0; c < 2; c++ )
  {
    if( !( bS[c] == 2 || ( large && bS[c] == 1 ) ) ) continue;
    pel* src = r->p[c + 1] + (size_t) cy * stride + cx;
    const int qp = lfp->qp[c + 1];
    const int idxTC = vvo_clip3( 0, 65, qp + 2 * ( bS[c] - 1 ) + 2 * H->deblock_tc_offset_div2[c + 1] );
    const int tc = tc_value( idxTC, bd );
    if( large )
    {
      const int idxB = vvo_clip3( 0, 63, qp + 2 * H->deblock_beta_offset_div2[c + 1] );
      const int beta = vvc_db_beta_table[idxB] * ( 1 << ( bd - 8 ) );
      const int dp0 = ctb ? calc_dp_ctb( src, o ) : calc_dp( src, o ), dq0 = calc_dq( src, o );
      const int dp3 = ctb ? calc_dp_ctb( src + step, o ) : calc_dp( src + step, o ), dq3 = calc_dq( src + step, o );   /* subSamplingShift == 1 */
      const int d0 = dp0 + dq0, d3 = dp3 + dq3, d = d0 + d3;
      if( d < beta )
      {
        const int sw = use_strong( src, o, 2 * d0, beta, tc, 0, 0, 7, 7, ctb ) && use_strong( src + step, o, 2 * d3, beta, tc, 0, 0, 7, 7, ctb );
        for( int i = 0; i < loopLen; i++ ) filter_chroma_pel( src + step * i, o, tc, sw, bd, ctb );
        continue;
      }
    }
    for( int i = 0; i < loopLen; i++ ) filter_chroma_pel( src + step * i, o, tc, 0, bd, ctb );
  }
}

/* the deblocking offsets are those of the slice the deblocked CTU belongs to - the CTU that holds the edge's position, i.e. its Q side
 * (LoopFilter::xDeblockCtuArea :421-422 passes ctuData.slice down to xEdgeFilterLuma :1473 / xEdgeFilterChroma :1637) */
static vvr_pic_header deblock_header_at( const vvr_picture* pic, int lx, int ly )
{
  vvr_pic_header h = pic->hdr;
  const vvr_slice_header* s = vvo_slice_at( pic, lx, ly );
  if( s ) for( int c = 0; c < 3; c++ ) { h.deblock_beta_offset_div2[c] = s->deblock_beta_offset_div2[c]; h.deblock_tc_offset_div2[c] = s->deblock_tc_offset_div2[c]; }
  return h;
}

void vvo_deblock( const vvr_picture* pic, vvo_planes* r, int dir )   /* xDeblockCtuArea (:419) for every CTU; order inside a direction is immaterial for valid streams */
{
  const vvr_pic_header* H = &pic->hdr;
  const int w4 = ( H->width + 3 ) >> 2, h4 = ( H->height + 3 ) >> 2;
  if( H->tool_flags & VVR_TOOL_DEBLOCK_OFF ) return;
  const vvr_lfp* T = pic->lfp[dir];
  for( int y4 = 0; y4 < h4; y4++ ) for( int x4 = 0; x4 < w4; x4++ )
  {
    const vvr_lfp* l = &T[(size_t) y4 * w4 + x4];
    if( BS_GET( l->bs, 0 ) ) { const vvr_pic_header Hs = deblock_header_at( pic, x4 * 4, y4 * 4 ); edge_luma( &Hs, r, x4 * 4, y4 * 4, l, dir ); }
  }
  if( !H->chroma_format ) return;
  /* chroma: edges on the 8-chroma-sample grid across, 2 chroma rows (one 4x4 luma unit) along (:457-489) */
  for( int y4 = 0; y4 < h4; y4++ ) for( int x4 = 0; x4 < w4; x4++ )
  {
    if( dir == 0 ? ( x4 & 3 ) : ( y4 & 3 ) ) continue;
    const vvr_lfp* l = &T[(size_t) y4 * w4 + x4];
    if( BS_GET( l->bs, 1 ) | BS_GET( l->bs, 2 ) ) { const vvr_pic_header Hs = deblock_header_at( pic, x4 * 4, y4 * 4 ); edge_chroma( &Hs, r, x4 * 2, y4 * 2, l, dir ); }
  }
}

/* ================================================================================================================= */
/* SAO: offsetBlock_core (SampleAdaptiveOffset.cpp:64) restated per sample.  With one slice/tile and no virtual      */
/* boundaries the eight availability flags (:741) reduce to "neighbour sample lies inside the picture".              */
/* ================================================================================================================= */
void vvo_sao( const vvr_picture* pic, const vvo_planes* src, vvo_planes* dst )
{
  const vvr_pic_header* H = &pic->hdr;
  const int ctu = 1 << H->log2_ctu, ctusX = ( H->width + ctu - 1 ) / ctu, bd = H->bit_depth;
  static const int dxy[4][2] = { { 1, 0 }, { 0, 1 }, { 1, 1 }, { -1, 1 } };   /* EO_0, EO_90, EO_135 (\), EO_45 (/): (dx,dy) of neighbour 'b'; 'a' is the opposite */
  for( int c = 0; c < src->ncomp; c++ )
  {
    const int cs = c ? 1 : 0, cw = src->w[c], chh = src->h[c], cctu = ctu >> cs;
    for( int y = 0; y < chh; y++ ) for( int x = 0; x < cw; x++ )
    {
      const vvr_sao_ctu* s = pic->sao ? &pic->sao[( y / cctu ) * ctusX + ( x / cctu )] : 0;
      const int v = src->p[c][(size_t) y * src->stride[c] + x];
      int out = v;
      if( s && s->mode[c] && ( H->tool_flags & ( c ? VVR_TOOL_SAO_CHROMA : VVR_TOOL_SAO_LUMA ) ) )
      {
        if( s->type[c] == 4 )
        {
          const int band = v >> ( bd - 5 ), k = ( band - s->band_pos[c] ) & 31;
          if( k < 4 ) out = vvo_clip_pel( v + s->offset[c][k], bd );
        }
        else
        {
          const int dx = dxy[s->type[c]][0], dy = dxy[s->type[c]][1];
          const int ax = x - dx, ay = y - dy, bx = x + dx, by = y + dy;
          /* neighbours outside the picture, or in a CTU of another slice / tile the loop filters may not cross, are not available
             (deriveLoopFilterBoundaryAvailibility, SampleAdaptiveOffset.cpp:741-805): the sample is left alone */
          /* picture-header virtual boundaries: the sample column (row) on either side of a boundary is skipped by the classes that look across
             it (isProcessDisabled :823; EO_0 only tests vertical boundaries :112, EO_90 only horizontal ones :156) */
          int atVb = 0;
          if( dx ) for( int i = 0; i < H->num_ver_vb; i++ ) { const int vb = H->vb_pos_x[i] >> cs; if( x == vb || x == vb - 1 ) atVb = 1; }
          if( dy ) for( int i = 0; i < H->num_hor_vb; i++ ) { const int vb = H->vb_pos_y[i] >> cs; if( y == vb || y == vb - 1 ) atVb = 1; }
          if( !atVb && ax >= 0 && ax < cw && ay >= 0 && ay < chh && bx >= 0 && bx < cw && by >= 0 && by < chh
              && vvo_lf_may_cross( pic, ( y / cctu ) * ctusX + x / cctu, ( ay / cctu ) * ctusX + ax / cctu ) && vvo_lf_may_cross( pic, ( y / cctu ) * ctusX + x / cctu, ( by / cctu ) * ctusX + bx / cctu ) )
          {
            const int a = src->p[c][(size_t) ay * src->stride[c] + ax], b = src->p[c][(size_t) by * src->stride[c] + bx];
            const int e = vvo_sgn( v - a ) + vvo_sgn( v - b );      /* -2 valley .. +2 peak */
            static const int cls[5] = { 0, 1, -1, 2, 3 };          /* offset[0,1,3,4] */
            if( e ) out = vvo_clip_pel( v + s->offset[c][cls[e + 2]], bd );
          }
        }
      }
      dst->p[c][(size_t) y * dst->stride[c] + x] = (pel) out;
    }
  }
}

/* ================================================================================================================= */
/* ALF                                                                                                               */
/* ================================================================================================================= */
/* What the ALF of the current CTU may read.  Picture borders are replicated (prepareCTU :453); at a CTU edge behind which lies a slice or tile the
 * filter must not look into, the reference filters a padded copy of the CTU instead (isClipOrCrossedByVirtualBoundaries :118-291, filterCTU
 * :764-840: copy + extendBorderPel), which is the same as clamping the coordinates to the CTU on the clipped sides.  rasterSliceAlfPad: the CTU
 * diagonally above-left (below-right) belongs to another slice while the ones above and left (below and right) do not - that corner is filled
 * from the first (last) column of the CTU, row by row (AreaBuf::padBorderPel, Buffer.h:608). */
static struct { int on; int x0[2], y0[2], x1[2], y1[2]; int l, r, t, b, tl, br; } g_alf;     /* [0] luma, [1] chroma coordinates of the CTU */
static void alf_set_ctu( const vvr_picture* pic, int ctuX, int ctuY )
{
  const vvr_pic_header* H = &pic->hdr;
  const int ctu = 1 << H->log2_ctu, ctusX = ( H->width + ctu - 1 ) / ctu, ctusY = ( H->height + ctu - 1 ) / ctu, a = ctuY * ctusX + ctuX;
  memset( &g_alf, 0, sizeof( g_alf ) );
  for( int k = 0; k < 2; k++ ) { const int S = ctu >> k; g_alf.x0[k] = ctuX * S; g_alf.y0[k] = ctuY * S; g_alf.x1[k] = g_alf.x0[k] + S - 1; g_alf.y1[k] = g_alf.y0[k] + S - 1; }
  const int hasL = ctuX > 0, hasR = ctuX + 1 < ctusX, hasT = ctuY > 0, hasB = ctuY + 1 < ctusY;
  g_alf.l = hasL && !vvo_lf_may_cross( pic, a, a - 1 );     g_alf.r = hasR && !vvo_lf_may_cross( pic, a, a + 1 );
  g_alf.t = hasT && !vvo_lf_may_cross( pic, a, a - ctusX ); g_alf.b = hasB && !vvo_lf_may_cross( pic, a, a + ctusX );
  if( ( H->tool_flags & VVR_TOOL_NO_LF_ACROSS_SLICES ) && pic->ctu_slice )
  {
    g_alf.tl = !g_alf.t && !g_alf.l && hasL && hasT && pic->ctu_slice[a - ctusX - 1] != pic->ctu_slice[a];
    g_alf.br = !g_alf.b && !g_alf.r && hasR && hasB && pic->ctu_slice[a + ctusX + 1] != pic->ctu_slice[a];
  }
  g_alf.on = g_alf.l | g_alf.r | g_alf.t | g_alf.b | g_alf.tl | g_alf.br;
}
/* picture-header virtual boundaries: those that touch or cross the CTU cut it into parts, each filtered with a replicated border of its own (a
 * boundary on the CTU's edge is a clipped edge, :146-172; interior ones: the loops of filterCTU :764-850).  Narrows the CTU's clip to the part
 * that holds the luma position (lx, ly); the corner padding of raster-scan slices only belongs to the part at the CTU's origin / end (:792,798). */
static void alf_set_part( const vvr_picture* pic, int lx, int ly )
{
  const vvr_pic_header* H = &pic->hdr;
  const int S = 1 << H->log2_ctu, cx0 = lx & ~( S - 1 ), cy0 = ly & ~( S - 1 );
  for( int i = 0; i < H->num_ver_vb; i++ )
  {
    const int v = H->vb_pos_x[i];
    if( v < cx0 || v > cx0 + S ) continue;
    if( v <= lx ) { g_alf.l = 1; g_alf.tl = 0; for( int k = 0; k < 2; k++ ) g_alf.x0[k] = vvo_max( g_alf.x0[k], v >> k ); }
    else          { g_alf.r = 1; g_alf.br = 0; for( int k = 0; k < 2; k++ ) g_alf.x1[k] = vvo_min( g_alf.x1[k], ( v >> k ) - 1 ); }
  }
  for( int i = 0; i < H->num_hor_vb; i++ )
  {
    const int v = H->vb_pos_y[i];
    if( v < cy0 || v > cy0 + S ) continue;
    if( v <= ly ) { g_alf.t = 1; g_alf.tl = 0; for( int k = 0; k < 2; k++ ) g_alf.y0[k] = vvo_max( g_alf.y0[k], v >> k ); }
    else          { g_alf.b = 1; g_alf.br = 0; for( int k = 0; k < 2; k++ ) g_alf.y1[k] = vvo_min( g_alf.y1[k], ( v >> k ) - 1 ); }
  }
  g_alf.on = g_alf.l | g_alf.r | g_alf.t | g_alf.b | g_alf.tl | g_alf.br;
}
static inline int alf_at( const vvo_planes* s, int c, int x, int y )
{
  if( g_alf.on )
  {
    const int k = c ? 1 : 0;
    if( g_alf.l && x < g_alf.x0[k] ) x = g_alf.x0[k];
    if( g_alf.r && x > g_alf.x1[k] ) x = g_alf.x1[k];
    if( g_alf.t && y < g_alf.y0[k] ) y = g_alf.y0[k];
    if( g_alf.b && y > g_alf.y1[k] ) y = g_alf.y1[k];
    if( g_alf.tl && x < g_alf.x0[k] && y < g_alf.y0[k] ) x = g_alf.x0[k];
    if( g_alf.br && x > g_alf.x1[k] && y > g_alf.y1[k] ) x = g_alf.x1[k];
  }
  x = vvo_clip3( 0, s->w[c] - 1, x ); y = vvo_clip3( 0, s->h[c] - 1, y );
  return s->p[c][(size_t) y * s->stride[c] + x];
}
static inline int clip_alf( int clip, int ref, int v0, int v1 ) { return vvo_clip3( -clip, clip, v0 - ref ) + vvo_clip3( -clip, clip, v1 - ref ); }   /* AdaptiveLoopFilter.h:93 */

/* class + transpose of the 4x4 block at (bx,by): deriveClassificationBlk (:969) */
static void alf_classify( const vvo_planes* s, int bx, int by, int bd, int ctu, int* classIdx, int* transposeIdx )
{
  static const int th[16] = { 0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4 };
  const int vbPos = ctu - 4, yInCtu = by & ( ctu - 1 );
  int sumV = 0, sumH = 0, sumD0 = 0, sumD1 = 0;
  for( int i = 0; i < 8; i += 2 )            /* rows  by-2+i, cells of 2 lines */
  {
    const int r = by - 2 + i;
    /* rows excluded at the virtual boundary (:1088-1109) */
    if( yInCtu == vbPos - 4 && i == 6 ) continue;
    if( yInCtu == vbPos     && i == 0 ) continue;
    int rm1 = r - 1, rp2 = r + 2;
    {
      /* row substitution at the virtual boundary while computing the Laplacians (:1006-1016); 'blk.y - 2 + i' there is CTU-relative */
      const int rel = yInCtu - 2 + i;
      if( rel > 0 && ( rel % ctu ) == vbPos - 2 ) rp2 = r + 1;
      else if( rel > 0 && ( rel % ctu ) == vbPos ) rm1 = r;
    }
    for( int j = 0; j < 8; j += 2 )
    {
      const int cX = bx - 2 + j;
      const int y0 = alf_at( s, 0, cX, r ) << 1, yup1 = alf_at( s, 0, cX + 1, r + 1 ) << 1;
      sumV  += vvo_abs( y0 - alf_at( s, 0, cX, rm1 ) - alf_at( s, 0, cX, r + 1 ) )         + vvo_abs( yup1 - alf_at( s, 0, cX + 1, r ) - alf_at( s, 0, cX + 1, rp2 ) );
      sumH  += vvo_abs( y0 - alf_at( s, 0, cX + 1, r ) - alf_at( s, 0, cX - 1, r ) )       + vvo_abs( yup1 - alf_at( s, 0, cX + 2, r + 1 ) - alf_at( s, 0, cX, r + 1 ) );
      sumD0 += vvo_abs( y0 - alf_at( s, 0, cX - 1, rm1 ) - alf_at( s, 0, cX + 1, r + 1 ) ) + vvo_abs( yup1 - alf_at( s, 0, cX, r ) - alf_at( s, 0, cX + 2, rp2 ) );
      sumD1 += vvo_abs( y0 - alf_at( s, 0, cX - 1, r + 1 ) - alf_at( s, 0, cX + 1, rm1 ) ) + vvo_abs( yup1 - alf_at( s, 0, cX, rp2 ) - alf_at( s, 0, cX + 2, r ) );
    }
  }
  const int shift = bd + 4;
  const int act = vvo_clip3( 0, 15, ( ( sumV + sumH ) * ( ( yInCtu == vbPos - 4 || yInCtu == vbPos ) ? 96 : 64 ) ) >> shift );
  int cls = th[act];
  int hv1, hv0, d1, d0, dirHV, dirD, hvd1, hvd0, mainDir, secDir;
  if( sumV > sumH ) { hv1 = sumV; hv0 = sumH; dirHV = 1; } else { hv1 = sumH; hv0 = sumV; dirHV = 3; }
  if( sumD0 > sumD1 ) { d1 = sumD0; d0 = sumD1; dirD = 0; } else { d1 = sumD1; d0 = sumD0; dirD = 2; }
  if( (uint32_t) d1 * (uint32_t) hv0 > (uint32_t) hv1 * (uint32_t) d0 ) { hvd1 = d1; hvd0 = d0; mainDir = dirD; secDir = dirHV; }
  else { hvd1 = hv1; hvd0 = hv0; mainDir = dirHV; secDir = dirD; }
  int strength = 0;
  if( hvd1 > 2 * hvd0 ) strength = 1;
  if( hvd1 * 2 > 9 * hvd0 ) strength = 2;
  if( strength ) cls += ( ( ( mainDir & 1 ) << 1 ) + strength ) * 5;
  static const int tt[8] = { 0, 1, 0, 2, 2, 3, 1, 3 };
  *classIdx = cls; *transposeIdx = tt[mainDir * 2 + ( secDir >> 1 )];
}

/* coefficient permutation for the four geometric transforms (AdaptiveLoopFilter.cpp:931-961) */
static const int alf_perm[4][13] = {
  { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12 },
  { 9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6, 12 },
  { 0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11, 12 },
  { 9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6, 12 } };

/* one sample of the diamond filter: filterBlk (:1176); taps[] / clips[] already permuted; n7 = 7x7 luma shape */
static int alf_sample( const vvo_planes* s, int c, int x, int y, const int16_t* cf, const int16_t* cl, int n7, int ctuH, int bd )
{
  const int vbPos = ctuH - ( n7 ? 4 : 2 ), yVb = y & ( ctuH - 1 );
  int r1 = y + 1, r2 = y - 1, r3 = y + 2, r4 = y - 2, r5 = y + 3, r6 = y - 3;    /* pImg1..6 */
  if( yVb < vbPos && yVb >= vbPos - ( n7 ? 4 : 2 ) )
  {
    r1 = ( yVb == vbPos - 1 ) ? y : r1;  r3 = ( yVb >= vbPos - 2 ) ? r1 : r3;  r5 = ( yVb >= vbPos - 3 ) ? r3 : r5;
    r2 = ( yVb == vbPos - 1 ) ? y : r2;  r4 = ( yVb >= vbPos - 2 ) ? r2 : r4;  r6 = ( yVb >= vbPos - 3 ) ? r4 : r6;
  }
  else if( yVb >= vbPos && yVb <= vbPos + ( n7 ? 3 : 1 ) )
  {
    r2 = ( yVb == vbPos ) ? y : r2;  r4 = ( yVb <= vbPos + 1 ) ? r2 : r4;  r6 = ( yVb <= vbPos + 2 ) ? r4 : r6;
    r1 = ( yVb == vbPos ) ? y : r1;  r3 = ( yVb <= vbPos + 1 ) ? r1 : r3;  r5 = ( yVb <= vbPos + 2 ) ? r3 : r5;
  }
  const int near = ( yVb == vbPos - 1 ) || ( yVb == vbPos );
  const int cur = alf_at( s, c, x, y );
  int sum = 0;
#define P( xx, rr ) alf_at( s, c, x + ( xx ), rr )
  if( n7 )
  {
    sum += cf[0]  * clip_alf( cl[0],  cur, P( 0, r5 ),  P( 0, r6 ) );
    sum += cf[1]  * clip_alf( cl[1],  cur, P( 1, r3 ),  P( -1, r4 ) );
    sum += cf[2]  * clip_alf( cl[2],  cur, P( 0, r3 ),  P( 0, r4 ) );
    sum += cf[3]  * clip_alf( cl[3],  cur, P( -1, r3 ), P( 1, r4 ) );
    sum += cf[4]  * clip_alf( cl[4],  cur, P( 2, r1 ),  P( -2, r2 ) );
    sum += cf[5]  * clip_alf( cl[5],  cur, P( 1, r1 ),  P( -1, r2 ) );
    sum += cf[6]  * clip_alf( cl[6],  cur, P( 0, r1 ),  P( 0, r2 ) );
    sum += cf[7]  * clip_alf( cl[7],  cur, P( -1, r1 ), P( 1, r2 ) );
    sum += cf[8]  * clip_alf( cl[8],  cur, P( -2, r1 ), P( 2, r2 ) );
    sum += cf[9]  * clip_alf( cl[9],  cur, P( 3, y ),   P( -3, y ) );
    sum += cf[10] * clip_alf( cl[10], cur, P( 2, y ),   P( -2, y ) );
    sum += cf[11] * clip_alf( cl[11], cur, P( 1, y ),   P( -1, y ) );
  }
  else
  {
    sum += cf[0] * clip_alf( cl[0], cur, P( 0, r3 ),  P( 0, r4 ) );
    sum += cf[1] * clip_alf( cl[1], cur, P( 1, r1 ),  P( -1, r2 ) );
    sum += cf[2] * clip_alf( cl[2], cur, P( 0, r1 ),  P( 0, r2 ) );
    sum += cf[3] * clip_alf( cl[3], cur, P( -1, r1 ), P( 1, r2 ) );
    sum += cf[4] * clip_alf( cl[4], cur, P( 2, y ),   P( -2, y ) );
    sum += cf[5] * clip_alf( cl[5], cur, P( 1, y ),   P( -1, y ) );
  }
#undef P
  sum = near ? ( sum + 512 ) >> 10 : ( sum + 64 ) >> 7;
  return vvo_clip_pel( sum + cur, bd );
}

static int ccalf_sample( const vvo_planes* s, int cx, int cy, const int16_t* cf, int ctu, int bd )   /* filterBlkCcAlf (:1348), 4:2:0 */
{
  const int vbPos = ctu - 4;
  const int lx = cx << 1, ly = cy << 1, pos = ly & ( ctu - 1 );
  int o1 = 1, o2 = -1, o3 = 2;
  if( pos == vbPos - 2 || pos == vbPos + 1 ) o3 = o1;
  else if( pos == vbPos - 1 || pos == vbPos ) { o1 = 0; o2 = 0; o3 = 0; }
  const int cur = alf_at( s, 0, lx, ly );
  int sum = 0;
  sum += cf[0] * ( alf_at( s, 0, lx,     ly + o2 ) - cur );
  sum += cf[1] * ( alf_at( s, 0, lx - 1, ly      ) - cur );
  sum += cf[2] * ( alf_at( s, 0, lx + 1, ly      ) - cur );
  sum += cf[3] * ( alf_at( s, 0, lx - 1, ly + o1 ) - cur );
  sum += cf[4] * ( alf_at( s, 0, lx,     ly + o1 ) - cur );
  sum += cf[5] * ( alf_at( s, 0, lx + 1, ly + o1 ) - cur );
  sum += cf[6] * ( alf_at( s, 0, lx,     ly + o3 ) - cur );
  sum = ( sum + 64 ) >> 7;
  const int off = 1 << bd >> 1;
  return vvo_clip_pel( sum + off, bd ) - off;
}

void vvo_alf( const vvr_picture* pic, const vvo_planes* src, vvo_planes* dst )   /* filterCTU (:664); slice / tile clipping and picture-header virtual boundaries through alf_at */
{
  const vvr_pic_header* H = &pic->hdr;
  const int ctu = 1 << H->log2_ctu, ctusX = ( H->width + ctu - 1 ) / ctu, bd = H->bit_depth;
  static const int clipDef[3] = { 256, 512, 1024 };
  for( int c = 0; c < src->ncomp; c++ )
  {
    const int cs = c ? 1 : 0, cctu = ctu >> cs;
    for( int y = 0; y < src->h[c]; y += 4 ) for( int x = 0; x < src->w[c]; x += 4 )
    {
      const vvr_alf_ctu* f = &pic->alf[( y / cctu ) * ctusX + ( x / cctu )];
      const vvr_alf_params* A = vvo_alf_set_at( pic, x << cs, y << cs );      /* the filters of the APSs the CTU's slice refers to (AdaptiveLoopFilter.cpp:515,558,603) */
      alf_set_ctu( pic, x / cctu, y / cctu );
      if( H->num_ver_vb | H->num_hor_vb ) alf_set_part( pic, x << cs, y << cs );
      int16_t cf[13], cl[13];
      int on = f->enable[c];
      if( on && c == 0 )
      {
        int cls, tr; alf_classify( src, x, y, bd, ctu, &cls, &tr );
        for( int k = 0; k < 12; k++ )
        {
          const int sk = alf_perm[tr][k];
          if( f->luma_filter_idx < 16 ) { cf[k] = (int16_t) vvc_alf_fixed_coeff[vvc_alf_class_to_filter[f->luma_filter_idx][cls]][sk]; cl[k] = (int16_t) clipDef[bd - 8]; }
          else { cf[k] = A->luma_coeff[f->luma_filter_idx - 16][cls][sk]; cl[k] = A->luma_clip[f->luma_filter_idx - 16][cls][sk]; }
        }
      }
      else if( on ) for( int k = 0; k < 6; k++ ) { cf[k] = A->chroma_coeff[f->alt[c - 1]][k]; cl[k] = A->chroma_clip[f->alt[c - 1]][k]; }
      for( int yy = y; yy < vvo_min( y + 4, src->h[c] ); yy++ ) for( int xx = x; xx < vvo_min( x + 4, src->w[c] ); xx++ )
      {
        int v = on ? alf_sample( src, c, xx, yy, cf, cl, c == 0, cctu, bd ) : src->p[c][(size_t) yy * src->stride[c] + xx];
        if( c && ( H->tool_flags & VVR_TOOL_CCALF ) && f->cc_idc[c - 1] )
          v = vvo_clip_pel( v + ccalf_sample( src, xx, yy, A->ccalf_coeff[c - 1][f->cc_idc[c - 1] - 1], ctu, bd ), bd );
        dst->p[c][(size_t) yy * dst->stride[c] + xx] = (pel) v;
      }
    }
  }
}
