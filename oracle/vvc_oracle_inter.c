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

int vvo_inter_cu( const vvr_picture* pic, const vvr_cu* cu, const vvo_planes* refs, int num_slots, vvo_planes* reco )
{
  const vvr_pic_header* H = &pic->hdr;
  const int bd = H->bit_depth, ctu = 1 << H->log2_ctu;
  const int ncomp = H->chroma_format ? 3 : 1;
  if( cu->mc_mode != VVR_MC_UNI && cu->mc_mode != VVR_MC_BI ) { vvo_set_error( "inter mode not restated yet" ); return -1; }
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
      if( cu->bcw_idx != 2 )
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
