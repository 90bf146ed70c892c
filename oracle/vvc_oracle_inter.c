/* oracle/vvc_oracle_inter.c — CPU restatement (TEST INFRASTRUCTURE): motion compensation.
 *
 * Follows  CommonLib/InterPrediction.cpp:1372-1459 (motionCompensation dispatch), :623-684 (xPredInterUni),
 *          :686-749 (xPredInterBi), :751-890 (xPredInterBlk), :1349-1370 (xWeightedAverage),
 *          CommonLib/InterpolationFilter.cpp:424-553 (filterCopy), :556-651 (filter<N>), :1057-1215 (filterHor/filterVer),
 *          CommonLib/Buffer.cpp:441-480 (addAvg), CommonLib/Mv.cpp:64-82 (clipMvInPic). */
#include "vvc_oracle_common.h"
#include "../tables/vvc_tables.inc"

#define IF_INTERNAL_PREC 14
#define IF_FILTER_PREC    6
#define IF_INTERNAL_OFFS ( 1 << ( IF_INTERNAL_PREC - 1 ) )

/* one component of one uni-directional prediction block: xPredInterBlk (InterPrediction.cpp:751).
 * bi = 1: output stays at the 14-bit intermediate precision; bi = 0: rounded and clipped samples. */
static void pred_block( const vvo_planes* ref, int comp, int bx, int by, int w, int h, int mvx, int mvy, int bi, int altHpel, int bd, pel* dst, int dstStride )
{
  const int sh = 4 + ( comp ? 1 : 0 );
  const int xFrac = mvx & ( ( 1 << sh ) - 1 ), yFrac = mvy & ( ( 1 << sh ) - 1 );
  const int x0 = bx + ( mvx >> sh ), y0 = by + ( mvy >> sh );
  const int ntaps = comp ? 4 : 8, half = ntaps / 2 - 1;
  const int headroom = IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2;
  const int16_t *ch, *cv;
  if( comp ) { ch = vvc_chroma_filter[xFrac]; cv = vvc_chroma_filter[yFrac]; }   /* frac << (1 - csx), csx = 1 for 4:2:0 */
  else
  {
    const int use4x4 = ( w == 4 && h == 4 );                                     /* InterpolationFilter.cpp:1078-1085, 669-676 */
    ch = ( xFrac == 8 && altHpel ) ? vvc_luma_alt_hpel : use4x4 ? vvc_luma_filter_4x4[xFrac] : vvc_luma_filter[xFrac];
    cv = ( yFrac == 8 && altHpel ) ? vvc_luma_alt_hpel : use4x4 ? vvc_luma_filter_4x4[yFrac] : vvc_luma_filter[yFrac];
  }
  if( xFrac == 0 && yFrac == 0 )
  {
    for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
    {
      const int s = vvo_ref_at( ref, comp, x0 + x, y0 + y );
      dst[y * dstStride + x] = bi ? (pel) ( (pel) ( s * ( 1 << headroom ) ) - (pel) IF_INTERNAL_OFFS ) : (pel) s;
    }
    return;
  }
  if( yFrac == 0 || xFrac == 0 )
  {   /* single 1-D pass, isFirst = true, isLast = !bi */
    const int16_t* c = yFrac == 0 ? ch : cv;
    const int dx = yFrac == 0 ? 1 : 0, dy = yFrac == 0 ? 0 : 1;
    int shift, offset;
    if( !bi ) { shift = IF_FILTER_PREC; offset = 1 << ( shift - 1 ); }
    else      { shift = IF_FILTER_PREC - headroom; offset = -IF_INTERNAL_OFFS * ( 1 << shift ); }
    for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
    {
      int sum = 0;
      for( int t = 0; t < ntaps; t++ ) sum += vvo_ref_at( ref, comp, x0 + x + ( t - half ) * dx, y0 + y + ( t - half ) * dy ) * c[t];
      pel val = (pel) ( ( sum + offset ) >> shift );
      dst[y * dstStride + x] = bi ? val : (pel) vvo_clip_pel( val, bd );
    }
    return;
  }
  {   /* separable 2-D: horizontal first into 16-bit temporaries (isFirst, !isLast), then vertical (!isFirst, isLast = !bi) */
    const int th = h + ntaps - 1;
    pel* tmp = (pel*) malloc( sizeof( pel ) * (size_t) w * th );
    const int shift1 = IF_FILTER_PREC - headroom, offset1 = -IF_INTERNAL_OFFS * ( 1 << shift1 );
    for( int y = 0; y < th; y++ ) for( int x = 0; x < w; x++ )
    {
      int sum = 0;
      for( int t = 0; t < ntaps; t++ ) sum += vvo_ref_at( ref, comp, x0 + x + t - half, y0 + y - half ) * ch[t];
      tmp[y * w + x] = (pel) ( ( sum + offset1 ) >> shift1 );
    }
    int shift2, offset2;
    if( !bi ) { shift2 = IF_FILTER_PREC + headroom; offset2 = ( 1 << ( shift2 - 1 ) ) + ( IF_INTERNAL_OFFS << IF_FILTER_PREC ); }
    else      { shift2 = IF_FILTER_PREC; offset2 = 0; }
    for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
    {
      int sum = 0;
      for( int t = 0; t < ntaps; t++ ) sum += tmp[( y + t ) * w + x] * cv[t];
      pel val = (pel) ( ( sum + offset2 ) >> shift2 );
      dst[y * dstStride + x] = bi ? val : (pel) vvo_clip_pel( val, bd );
    }
    free( tmp );
  }
}

static void clip_mv( int mv[2], int x, int y, int W, int H, int ctu )   /* clipMvInPic (Mv.cpp:64) */
{
  const int horMax = ( W + 8 - x - 1 ) * 16, horMin = ( -ctu - 8 - x + 1 ) * 16;
  const int verMax = ( H + 8 - y - 1 ) * 16, verMin = ( -ctu - 8 - y + 1 ) * 16;
  mv[0] = vvo_min( horMax, vvo_max( horMin, mv[0] ) );
  mv[1] = vvo_min( verMax, vvo_max( verMin, mv[1] ) );
}

/* ---------------------------------------------------------------------------------------------------------------------
 * BDOF on one <= 16x16 luma sub-block: InterPrediction::xSubPuBio (InterPrediction.cpp:551) -> xPredInterUni( bi, bioApplied )
 * -> xPredInterBlk border fill (:863-890, PaddBIOCore :269) -> xWeightedAverage( bioApplied ) (:1349) -> applyBiOptFlow (:1290):
 * gradFilterCore<true> (:213), BiOptFlowCore (:162), calcBIOSums (:134), rightShiftMSB (:92), addBIOAvg4 (:108).
 * src[l]: 14-bit predictions of both lists in a (w+8)-stride buffer, block at (row 2, col 1), one-sample border from the
 * nearest integer reference samples around it. */
#define BIO_STRIDE_MAX ( 16 + 8 )
static int bdof_right_shift_msb( int numer, int denom )
{
  int msb = 0;
  for( msb = 0; msb < 32; msb++ ) if( denom < ( 1 << msb ) ) break;
  return numer >> ( msb - 1 );
}

static void bdof_border( const vvo_planes* ref, int bx, int by, int w, int h, int mvx, int mvy, int bd, pel* blk /* row 0 of the (w+8)-stride buffer */ )
{
  const int S = w + 8;
  const int shift = IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2;
  const int x0 = bx + ( mvx >> 4 ), y0 = by + ( mvy >> 4 );
  const int xOff = ( mvx & 15 ) < 8 ? 1 : 0, yOff = ( mvy & 15 ) < 8 ? 1 : 0;
  for( int r = 0; r < h; r++ )
  {
    pel* d = blk + ( 2 + r ) * S;
    d[0]     = (pel) ( vvo_ref_at( ref, 0, x0 - xOff,         y0 + 1 - yOff + r ) * ( 1 << shift ) - (pel) IF_INTERNAL_OFFS );
    d[w + 1] = (pel) ( vvo_ref_at( ref, 0, x0 - xOff + w + 1, y0 + 1 - yOff + r ) * ( 1 << shift ) - (pel) IF_INTERNAL_OFFS );
  }
  for( int i = 0; i < w + 2; i++ )
  {
    blk[1 * S + i]         = (pel) ( vvo_ref_at( ref, 0, x0 - xOff + i, y0 - yOff )         * ( 1 << shift ) - (pel) IF_INTERNAL_OFFS );
    blk[( h + 2 ) * S + i] = (pel) ( vvo_ref_at( ref, 0, x0 - xOff + i, y0 + h + 1 - yOff ) * ( 1 << shift ) - (pel) IF_INTERNAL_OFFS );
  }
}

static void bdof_apply( pel* blk0, pel* blk1, int w, int h, int bd, pel* dst, int dstStride )
{
  const int S = w + 8;                                     /* stridePredMC = widthG = width + BIO_ALIGN_SIZE */
  pel gx[2][( 16 + 2 ) * BIO_STRIDE_MAX], gy[2][( 16 + 2 ) * BIO_STRIDE_MAX];
  pel* P[2] = { blk0 + S, blk1 + S };                      /* padded (w+2) x (h+2) region, origin = top border row */
  for( int l = 0; l < 2; l++ )
  {
    pel* s = P[l];
    for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
    {
      gy[l][( 1 + y ) * S + 1 + x] = (pel) ( ( s[( 2 + y ) * S + 1 + x] >> 6 ) - ( s[y * S + 1 + x] >> 6 ) );
      gx[l][( 1 + y ) * S + 1 + x] = (pel) ( ( s[( 1 + y ) * S + 2 + x] >> 6 ) - ( s[( 1 + y ) * S + x] >> 6 ) );
    }
    /* padding: gradients AND the prediction border are replaced by replicas of the interior (:236-264) */
    for( int y = 1; y <= h; y++ )
    {
      gx[l][y * S] = gx[l][y * S + 1]; gx[l][y * S + w + 1] = gx[l][y * S + w];
      gy[l][y * S] = gy[l][y * S + 1]; gy[l][y * S + w + 1] = gy[l][y * S + w];
      s[y * S] = s[y * S + 1];         s[y * S + w + 1] = s[y * S + w];
    }
    memcpy( &gx[l][0], &gx[l][S], sizeof( pel ) * ( w + 2 ) ); memcpy( &gx[l][( h + 1 ) * S], &gx[l][h * S], sizeof( pel ) * ( w + 2 ) );
    memcpy( &gy[l][0], &gy[l][S], sizeof( pel ) * ( w + 2 ) ); memcpy( &gy[l][( h + 1 ) * S], &gy[l][h * S], sizeof( pel ) * ( w + 2 ) );
    memcpy( &s[0], &s[S], sizeof( pel ) * ( w + 2 ) );         memcpy( &s[( h + 1 ) * S], &s[h * S], sizeof( pel ) * ( w + 2 ) );
  }
  const int shiftNum = IF_INTERNAL_PREC + 1 - bd, offset = ( 1 << ( shiftNum - 1 ) ) + 2 * IF_INTERNAL_OFFS, limit = 15;
  for( int yu = 0; yu < h >> 2; yu++ ) for( int xu = 0; xu < w >> 2; xu++ )
  {
    int sumAbsGX = 0, sumAbsGY = 0, sumDIX = 0, sumDIY = 0, sumSignGY_GX = 0;
    for( int y = 0; y < 6; y++ ) for( int x = 0; x < 6; x++ )
    {
      const int o = ( ( yu << 2 ) + y ) * S + ( xu << 2 ) + x;
      const int tGX = ( gx[0][o] + gx[1][o] ) >> 1, tGY = ( gy[0][o] + gy[1][o] ) >> 1;
      const int tDI = ( P[1][o] >> 4 ) - ( P[0][o] >> 4 );
      sumAbsGX += vvo_abs( tGX ); sumAbsGY += vvo_abs( tGY );
      sumDIX += tGX < 0 ? -tDI : tGX == 0 ? 0 : tDI;
      sumDIY += tGY < 0 ? -tDI : tGY == 0 ? 0 : tDI;
      sumSignGY_GX += tGY < 0 ? -tGX : tGY == 0 ? 0 : tGX;
    }
    int tmpx = sumAbsGX == 0 ? 0 : bdof_right_shift_msb( sumDIX * 4, sumAbsGX );
    tmpx = vvo_clip3( -limit, limit, tmpx );
    const int mains = sumSignGY_GX >> 12, secs = sumSignGY_GX & ( ( 1 << 12 ) - 1 );
    int tmpData = tmpx * mains;
    tmpData = ( ( tmpData * ( 1 << 12 ) ) + tmpx * secs ) >> 1;
    int tmpy = sumAbsGY == 0 ? 0 : bdof_right_shift_msb( ( sumDIY * 4 ) - tmpData, sumAbsGY );
    tmpy = vvo_clip3( -limit, limit, tmpy );
    for( int y = 0; y < 4; y++ ) for( int x = 0; x < 4; x++ )
    {
      const int o = ( 1 + ( yu << 2 ) + y ) * S + 1 + ( xu << 2 ) + x;
      const int b = tmpx * ( gx[0][o] - gx[1][o] ) + tmpy * ( gy[0][o] - gy[1][o] );
      dst[( ( yu << 2 ) + y ) * dstStride + ( xu << 2 ) + x] = (pel) vvo_clip_pel( (int16_t) ( ( P[0][o] + P[1][o] + b + offset ) >> shiftNum ), bd );
    }
  }
}

/* luma of one bi-predicted sub-block with BDOF; mv[l] already clipped against the CU (xPredInterUni :655-658) */
static void bdof_luma_subblock( const vvo_planes* ref0, const vvo_planes* ref1, int bx, int by, int w, int h, const int mv0[2], const int mv1[2],
                                int altHpel, int bd, pel* dst, int dstStride )
{
  pel blk[2][( 16 + 4 ) * BIO_STRIDE_MAX];
  const int S = w + 8;
  memset( blk, 0, sizeof( blk ) );
  pred_block( ref0, 0, bx, by, w, h, mv0[0], mv0[1], 1, altHpel, bd, blk[0] + 2 * S + 1, S );
  pred_block( ref1, 0, bx, by, w, h, mv1[0], mv1[1], 1, altHpel, bd, blk[1] + 2 * S + 1, S );
  bdof_border( ref0, bx, by, w, h, mv0[0], mv0[1], bd, blk[0] );
  bdof_border( ref1, bx, by, w, h, mv1[0], mv1[1], bd, blk[1] );
  bdof_apply( blk[0], blk[1], w, h, bd, dst, dstStride );
}

int vvo_inter_cu( const vvr_picture* pic, const vvr_cu* cu, const vvo_planes* refs, int num_slots, vvo_planes* reco )
{
  const vvr_pic_header* H = &pic->hdr;
  const int bd = H->bit_depth, ctu = 1 << H->log2_ctu;
  const int ncomp = H->chroma_format ? 3 : 1;
  if( cu->mc_mode != VVR_MC_UNI && cu->mc_mode != VVR_MC_BI && cu->mc_mode != VVR_MC_BDOF ) { vvo_set_error( "inter mode not restated yet" ); return -1; }
  const int altHpel = cu->imv == 3;
  const int biPred = cu->ref_idx[0] >= 0 && cu->ref_idx[1] >= 0;
  for( int c = 0; c < ncomp; c++ )
  {
    const int cs = c ? 1 : 0;
    const int bx = cu->x >> cs, by = cu->y >> cs, w = cu->w >> cs, h = cu->h >> cs;
    pel* dst = reco->p[c] + (size_t) by * reco->stride[c] + bx;
    if( cu->mc_mode == VVR_MC_UNI )
    {
      /* one list, or bi with identical motion: xPredInterUni( cu, L0, predBuf, bi=false ) (InterPrediction.cpp:1451-1454) */
      const int l = ( biPred || cu->ref_idx[0] >= 0 ) ? 0 : 1;
      int mv[2] = { cu->mv[l][0][0], cu->mv[l][0][1] };
      clip_mv( mv, cu->x, cu->y, H->width, H->height, ctu );
      const int slot = H->ref_slot[l][cu->ref_idx[l]];
      if( slot < 0 || slot >= num_slots || !refs[slot].p[0] ) { vvo_set_error( "missing reference slot" ); return -1; }
      pred_block( &refs[slot], c, bx, by, w, h, mv[0], mv[1], 0, altHpel, bd, dst, reco->stride[c] );
    }
    else
    {
      pel* t0 = (pel*) malloc( sizeof( pel ) * (size_t) w * h * 2 ); pel* t1 = t0 + (size_t) w * h;
      for( int l = 0; l < 2; l++ )
      {
        int mv[2] = { cu->mv[l][0][0], cu->mv[l][0][1] };
        clip_mv( mv, cu->x, cu->y, H->width, H->height, ctu );
        const int slot = H->ref_slot[l][cu->ref_idx[l]];
        if( slot < 0 || slot >= num_slots || !refs[slot].p[0] ) { free( t0 ); vvo_set_error( "missing reference slot" ); return -1; }
        pred_block( &refs[slot], c, bx, by, w, h, mv[0], mv[1], 1, altHpel, bd, l ? t1 : t0, w );
      }
      if( c == 0 && cu->mc_mode == VVR_MC_BDOF )
      {   /* xSubPuBio (:551): sub-blocks of at most 16x16 (MAX_BDOF_APPLICATION_REGION), each with its own border fetch */
        int mv0[2] = { cu->mv[0][0][0], cu->mv[0][0][1] }, mv1[2] = { cu->mv[1][0][0], cu->mv[1][0][1] };
        clip_mv( mv0, cu->x, cu->y, H->width, H->height, ctu ); clip_mv( mv1, cu->x, cu->y, H->width, H->height, ctu );
        const int sw = vvo_min( 16, w ), shh = vvo_min( 16, h );
        for( int y = 0; y < h; y += shh ) for( int x = 0; x < w; x += sw )
          bdof_luma_subblock( &refs[H->ref_slot[0][cu->ref_idx[0]]], &refs[H->ref_slot[1][cu->ref_idx[1]]], bx + x, by + y, sw, shh, mv0, mv1, altHpel, bd,
                              dst + (size_t) y * reco->stride[c] + x, reco->stride[c] );
      }
      else if( cu->bcw_idx != 2 )
      {   /* addWeightedAvg (Buffer.cpp:372): BCW */
        const int w1 = vvc_bcw_weights[cu->bcw_idx], w0 = 8 - w1;
        const int shift = ( IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2 ) + 3, offset = ( 1 << ( shift - 1 ) ) + ( IF_INTERNAL_OFFS << 3 );
        for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
          dst[y * reco->stride[c] + x] = (pel) vvo_clip_pel( ( t0[y * w + x] * w0 + t1[y * w + x] * w1 + offset ) >> shift, bd );
      }
      else
      {   /* addAvg (Buffer.cpp:441) */
        const int shift = ( IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2 ) + 1, offset = ( 1 << ( shift - 1 ) ) + 2 * IF_INTERNAL_OFFS;
        for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
          dst[y * reco->stride[c] + x] = (pel) vvo_clip_pel( ( t0[y * w + x] + t1[y * w + x] + offset ) >> shift, bd );
      }
      free( t0 );
    }
  }
  return 0;
}
