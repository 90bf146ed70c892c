/* oracle/vvc_oracle_inter.c — CPU restatement (TEST INFRASTRUCTURE): motion compensation.
 *
 * Follows  CommonLib/InterPrediction.cpp:1372-1459 (motionCompensation dispatch), :623-684 (xPredInterUni),
 *          :686-749 (xPredInterBi), :751-890 (xPredInterBlk), :1349-1370 (xWeightedAverage),
 *          CommonLib/InterpolationFilter.cpp:424-553 (filterCopy), :556-651 (filter<N>), :1057-1215 (filterHor/filterVer),
 *          CommonLib/Buffer.cpp:441-480 (addAvg), CommonLib/Mv.cpp:64-82 (clipMvInPic),
 *          CommonLib/InterPrediction.cpp:2081-2217 (xPredInterBlkRPR: predictions from scaled reference pictures). */
#include "vvc_oracle_common.h"
#include "../tables/vvc_tables.inc"

#define IF_INTERNAL_PREC 14
#define IF_FILTER_PREC    6
#define IF_INTERNAL_OFFS ( 1 << ( IF_INTERNAL_PREC - 1 ) )

/* one component of one uni-directional prediction block: xPredInterBlk (InterPrediction.cpp:751).
 * bi = 1: output stays at the 14-bit intermediate precision; bi = 0: rounded and clipped samples. */
/* sample source of a prediction block: the (border-extended) reference picture, or DMVR's padded local copy of it */
typedef struct {
  const vvo_planes* ref; int comp;
  const pel* pad; int padStride, padW, padH, padX0, padY0;   /* pad != NULL: pad[(y - padY0) * padStride + (x - padX0)], x/y in block-relative coordinates */
} vvo_src;
/* Reference wrap-around (pps_ref_wraparound_enabled_flag, vvr_pic_header.wrap_offset): the reference keeps a second copy of every reference picture
 * whose left / right margins continue the picture one period further on (Picture::extendPicBorderWrap, Picture.cpp:410-518: margin sample -k-1 =
 * sample offset-k-1 as long as k < offset, else the edge sample), and every prediction reads either that copy or the ordinary, edge-replicated one:
 * g_wrapFetch says which, set by the callers from what wrapClipMv (Mv.cpp:112) returned for the block. */
static int g_wrapOff = 0, g_wrapFetch = 0;
/* Sub-pictures: a CU in a sub-picture that is treated as a picture (sps_subpic_treated_as_pic_flag) is predicted from a copy of that sub-picture of
 * the reference picture with its own replicated border (Picture::getSubPicBuf, DecLibRecon::createSubPicRefBufs, DecLibRecon.cpp:388-421) = reads
 * clamped to the rectangle, and its MVs are clipped against the rectangle (clipMvInSubpic, Mv.cpp:84).  g_mcRect: luma rectangle of the current CU. */
static int g_mcRect[4] = { 0, 0, 0, 0 }, g_mcRectOn = 0;
static inline int ref_at( const vvo_planes* r, int c, int x, int y )
{
  if( g_mcRectOn )
  {
    const int cs = c ? 1 : 0;
    x = vvo_clip3( g_mcRect[0] >> cs, g_mcRect[2] >> cs, x ); y = vvo_clip3( g_mcRect[1] >> cs, g_mcRect[3] >> cs, y );
  }
  if( g_wrapFetch )
  {
    const int w = r->w[c], off = g_wrapOff >> ( c ? 1 : 0 );
    if( x < 0 ) x = -x <= off ? x + off : 0;
    else if( x >= w ) x = x - w < off ? x - off : w - 1;
  }
  return vvo_ref_at( r, c, x, y );
}
static inline int src_at( const vvo_src* s, int x, int y )
{
  if( s->pad ) return s->pad[( y - s->padY0 ) * s->padStride + ( x - s->padX0 )];
  return ref_at( s->ref, s->comp, x, y );
}
static void pred_block_src( const vvo_src* src, int comp, int bx, int by, int w, int h, int mvx, int mvy, int bi, int altHpel, int bd, pel* dst, int dstStride );
static void pred_block( const vvo_planes* ref, int comp, int bx, int by, int w, int h, int mvx, int mvy, int bi, int altHpel, int bd, pel* dst, int dstStride )
{
  vvo_src s; memset( &s, 0, sizeof( s ) ); s.ref = ref; s.comp = comp;
  pred_block_src( &s, comp, bx, by, w, h, mvx, mvy, bi, altHpel, bd, dst, dstStride );
}
/* with a pad source (altSrc, InterPrediction.cpp:789-797) x0/y0 below are relative to the pad's own origin: the caller passes bx = by = 0
 * and an mv whose integer part is zero */
static void pred_block_src( const vvo_src* ref, int comp, int bx, int by, int w, int h, int mvx, int mvy, int bi, int altHpel, int bd, pel* dst, int dstStride )
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
      const int s = src_at( ref, x0 + x, y0 + y );
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
      for( int t = 0; t < ntaps; t++ ) sum += src_at( ref, x0 + x + ( t - half ) * dx, y0 + y + ( t - half ) * dy ) * c[t];
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
      for( int t = 0; t < ntaps; t++ ) sum += src_at( ref, x0 + x + t - half, y0 + y - half ) * ch[t];
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

/* ---------------------------------------------------------------------------------------------------------------------
 * Reference picture resampling: InterPrediction::xPredInterBlkRPR (InterPrediction.cpp:2081-2217).  The reference filters column by column into a
 * buffer and then row by row out of it; every output sample is  sum_t fV[yFrac(row)][t] * tmp( yInt(row) - (N/2 - 1) + t, col )  with
 * tmp( y, col ) = ( sum_s fH[xFrac(col)][s] * ref( xInt(col) - (N/2 - 1) + s, y ) + offset ) >> shift  (16-bit), and the rows / columns it reads
 * outside the picture are the border extension (the rows below picture + margin are copies of the last filtered row, :2188-2196, which is a row of
 * the margin itself) = clamped reads.  frac == 0 with the regular filters takes filterCopy in the reference (InterpolationFilter.cpp:1059-1065,
 * 1147-1150); the filters' phase 0 is { 0, 0, 0, 64, 0 .. } and gives the same numbers, so one code path serves.
 * filterIndex: 0 regular CU, 2 affine sub-block (6-tap filters and their own low-pass sets).  No 4x4 special case: the reference calls
 * filterHor / filterVer with width 1 / height 1 here. */
static const vvr_rpr_params* g_rpr = 0;
static const vvr_rpr_ref* rpr_of( int l, int ri ) { return g_rpr && l >= 0 && ri >= 0 && g_rpr->ref[l][ri].scaled ? &g_rpr->ref[l][ri] : 0; }
static void pred_block_rpr( const vvo_planes* ref, const vvr_rpr_ref* rr, int comp, int bx, int by, int w, int h, int mvx, int mvy, int bi, int altHpel, int filterIndex, int bd, pel* dst, int dstStride )
{
  const int cs = comp ? 1 : 0, shiftHor = 4 + cs, shiftVer = 4 + cs;
  const int thr1 = ( 1 << 14 ) * 5 / 4, thr2 = ( 1 << 14 ) * 7 / 4;
  const int rx = rr->ratio[0], ry = rr->ratio[1];
  int xFilter = filterIndex, yFilter = filterIndex;
  if( rx > thr2 ) xFilter = 4; else if( rx > thr1 ) xFilter = 3;
  if( ry > thr2 ) yFilter = 4; else if( ry > thr1 ) yFilter = 3;
  if( !comp && filterIndex == 2 ) { if( rx > thr1 ) xFilter += 2; if( ry > thr1 ) yFilter += 2; }
  const int posShift = 14 - 4;
  const int stepX = ( rx + 8 ) >> 4, stepY = ( ry + 8 ) >> 4;
  const int offX = 1 << ( posShift - shiftHor - 1 ), offY = 1 << ( posShift - shiftVer - 1 );
  const int64_t posX = ( ( bx << cs ) - g_rpr->win_left ) >> cs, posY = ( ( by << cs ) - g_rpr->win_top ) >> cs;
  const int addX = comp ? ( 1 - rr->hor_collocated_chroma ) * 8 * ( rx - ( 1 << 14 ) ) : 0;
  const int addY = comp ? ( 1 - rr->ver_collocated_chroma ) * 8 * ( ry - ( 1 << 14 ) ) : 0;
  int64_t x0 = ( posX * ( 1 << ( 4 + cs ) ) + mvx ) * (int64_t) rx + addX;
  x0 = ( x0 >= 0 ? 1 : -1 ) * ( ( ( x0 >= 0 ? x0 : -x0 ) + ( (int64_t) 1 << ( 7 + cs ) ) ) >> ( 8 + cs ) ) + ( (int64_t) rr->win_left * ( 1 << ( posShift - cs ) ) );
  int64_t y0 = ( posY * ( 1 << ( 4 + cs ) ) + mvy ) * (int64_t) ry + addY;
  y0 = ( y0 >= 0 ? 1 : -1 ) * ( ( ( y0 >= 0 ? y0 : -y0 ) + ( (int64_t) 1 << ( 7 + cs ) ) ) >> ( 8 + cs ) ) + ( (int64_t) rr->win_top * ( 1 << ( posShift - cs ) ) );
  const int ntaps = comp ? 4 : 8, half = ntaps / 2 - 1;
  const int rw = rr->width >> cs, rh = rr->height >> cs;
  const int headroom = IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2;
  const int shift1 = IF_FILTER_PREC - headroom, offset1 = -IF_INTERNAL_OFFS * ( 1 << shift1 );
  int shift2, offset2;
  if( !bi ) { shift2 = IF_FILTER_PREC + headroom; offset2 = ( 1 << ( shift2 - 1 ) ) + ( IF_INTERNAL_OFFS << IF_FILTER_PREC ); }
  else      { shift2 = IF_FILTER_PREC; offset2 = 0; }
  for( int row = 0; row < h; row++ )
  {
    const int32_t py = (int32_t) y0 + row * stepY;
    const int yInt = vvo_clip3( -4, rh + 4, ( py + offY ) >> posShift );
    const int yFrac = ( ( py + offY ) >> ( posShift - shiftVer ) ) & ( ( 1 << shiftVer ) - 1 );
    const int16_t* cv;
    if( comp ) cv = yFilter == 3 ? vvc_chroma_filter_rpr1[yFrac] : yFilter == 4 ? vvc_chroma_filter_rpr2[yFrac] : vvc_chroma_filter[yFrac];
    else if( yFilter == 0 ) cv = ( yFrac == 8 && altHpel && ry == ( 1 << 14 ) ) ? vvc_luma_alt_hpel : vvc_luma_filter[yFrac];
    else cv = yFilter == 2 ? vvc_luma_filter_4x4[yFrac] : yFilter == 3 ? vvc_luma_filter_rpr1[yFrac] : yFilter == 4 ? vvc_luma_filter_rpr2[yFrac]
            : yFilter == 5 ? vvc_affine_luma_filter_rpr1[yFrac] : vvc_affine_luma_filter_rpr2[yFrac];
    for( int col = 0; col < w; col++ )
    {
      const int32_t px = (int32_t) x0 + col * stepX;
      const int xInt = vvo_clip3( -4, rw + 4, ( px + offX ) >> posShift );
      const int xFrac = ( ( px + offX ) >> ( posShift - shiftHor ) ) & ( ( 1 << shiftHor ) - 1 );
      const int16_t* ch;
      if( comp ) ch = xFilter == 3 ? vvc_chroma_filter_rpr1[xFrac] : xFilter == 4 ? vvc_chroma_filter_rpr2[xFrac] : vvc_chroma_filter[xFrac];
      else if( xFilter == 0 ) ch = ( xFrac == 8 && altHpel && rx == ( 1 << 14 ) ) ? vvc_luma_alt_hpel : vvc_luma_filter[xFrac];
      else ch = xFilter == 2 ? vvc_luma_filter_4x4[xFrac] : xFilter == 3 ? vvc_luma_filter_rpr1[xFrac] : xFilter == 4 ? vvc_luma_filter_rpr2[xFrac]
              : xFilter == 5 ? vvc_affine_luma_filter_rpr1[xFrac] : vvc_affine_luma_filter_rpr2[xFrac];
      int sum2 = 0;
      for( int t = 0; t < ntaps; t++ )
      {
        int sum = 0;
        for( int u = 0; u < ntaps; u++ ) sum += vvo_ref_at( ref, comp, xInt - half + u, yInt - half + t ) * ch[u];
        sum2 += (pel) ( ( sum + offset1 ) >> shift1 ) * cv[t];
      }
      const pel val = (pel) ( ( sum2 + offset2 ) >> shift2 );
      dst[row * dstStride + col] = bi ? val : (pel) vvo_clip_pel( val, bd );
    }
  }
}

/* a block of a regular (non-affine) CU from reference picture (l, ri): scaled -> the RPR path, MV as it is (xPredInterUni, InterPrediction.cpp:650,673) */
static void pred_any( const vvo_planes* ref, const vvr_rpr_ref* rr, int comp, int bx, int by, int w, int h, int mvx, int mvy, int bi, int altHpel, int bd, pel* dst, int dstStride )
{
  if( rr ) pred_block_rpr( ref, rr, comp, bx, by, w, h, mvx, mvy, bi, altHpel, 0, bd, dst, dstStride );
  else pred_block( ref, comp, bx, by, w, h, mvx, mvy, bi, altHpel, bd, dst, dstStride );
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Explicit weighted prediction: WeightPrediction::getWpScaling (WeightPrediction.cpp:66-157), addWeightUni (:238-338) and
 * addWeightBi (:164-236).  It replaces the final rounding / averaging of plain, affine, SbTMVP and CIIP inter predictions of a
 * picture with VVR_TOOL_WP (InterPrediction::xPredInterBi, InterPrediction.cpp:707,735-742) unless the CU uses BCW weights or GPM;
 * the inputs are the 14-bit intermediate predictions.
 * ------------------------------------------------------------------------------------------------------------------- */
/* the table and the switch are the slice's (vvr_slice_header.wp_set, VVR_TOOL_WP among its flags): set per CU by vvo_inter_cu */
static const vvr_wp_params* g_wp = 0;
static int g_wpOn = 0;
static int wp_on( const vvr_picture* pic, int bcw_idx ) { (void) pic; return g_wpOn && g_wp && bcw_idx == 2; }
static int wp_uni( const vvr_picture* pic, int l, int ri, int c, int p )
{
  const vvr_wp_entry* e = &g_wp->e[l][ri][c];
  const int bd = pic->hdr.bit_depth, den = g_wp->log2_denom[c ? 1 : 0];
  const int shiftNum = IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2, shift = den + shiftNum;
  const int offset = e->offset * ( 1 << ( bd - 8 ) );
  if( e->weight != ( 1 << den ) ) return vvo_clip_pel( ( ( e->weight * ( p + IF_INTERNAL_OFFS ) + ( 1 << ( shift - 1 ) ) ) >> shift ) + offset, bd );
  return vvo_clip_pel( ( ( p + IF_INTERNAL_OFFS + ( 1 << ( shiftNum - 1 ) ) ) >> shiftNum ) + offset, bd );
}
static int wp_bi( const vvr_picture* pic, int r0, int r1, int c, int p0, int p1 )
{
  const vvr_wp_entry* e0 = &g_wp->e[0][r0][c]; const vvr_wp_entry* e1 = &g_wp->e[1][r1][c];
  const int bd = pic->hdr.bit_depth, den = g_wp->log2_denom[c ? 1 : 0];
  const int shiftNum = IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2, shift = den + 1 + shiftNum;
  const int offset = ( e0->offset + e1->offset ) * ( 1 << ( bd - 8 ) );
  return vvo_clip_pel( ( e0->weight * ( p0 + IF_INTERNAL_OFFS ) + e1->weight * ( p1 + IF_INTERNAL_OFFS ) + ( ( 1 << shift ) >> 1 ) + offset * ( 1 << ( shift - 1 ) ) ) >> shift, bd );
}

/* wrapClipMv (Mv.cpp:112): an MV that points further out than the wrap copy's margins is moved by one period and clamped; returns whether the wrap
 * copy is the one to read (it is not after a move).  bw: width of the block the MV belongs to. */
static int wrap_clip_mv( int mv[2], int x, int y, int bw, int W, int H, int ctu )
{
  int wrapRef = 1;
  const int horMax = ( W + ctu - bw + 8 - x - 1 ) * 16, horMin = ( -ctu - 8 - x + 1 ) * 16;
  const int verMax = ( H + 8 - y - 1 ) * 16, verMin = ( -ctu - 8 - y + 1 ) * 16;
  int mx = mv[0];
  if( mx > horMax ) { mx -= g_wrapOff * 16; mx = vvo_min( horMax, vvo_max( horMin, mx ) ); wrapRef = 0; }
  if( mx < horMin ) { mx += g_wrapOff * 16; mx = vvo_min( horMax, vvo_max( horMin, mx ) ); wrapRef = 0; }
  mv[0] = mx; mv[1] = vvo_min( verMax, vvo_max( verMin, mv[1] ) );
  return wrapRef;
}
/* clipMvInPic (Mv.cpp:64) for the block at luma (x, y), bw wide; with wrap-around it is wrapClipMv (:66-70).  Sets g_wrapFetch for the prediction that
 * follows: the regular paths call wrapClipMv a second time on the result (InterPrediction.cpp:656,1752,1814), which then lies inside the range, so they
 * always read the wrap copy. */
static void clip_mv_w( int mv[2], int x, int y, int bw, int W, int H, int ctu )
{
  if( g_wrapOff ) { wrap_clip_mv( mv, x, y, bw, W, H, ctu ); g_wrapFetch = 1; return; }
  int horMax = ( W + 8 - x - 1 ) * 16, horMin = ( -ctu - 8 - x + 1 ) * 16;
  int verMax = ( H + 8 - y - 1 ) * 16, verMin = ( -ctu - 8 - y + 1 ) * 16;
  if( g_mcRectOn )
  {   /* clipMvInSubpic (Mv.cpp:84-107) */
    horMax = ( g_mcRect[2] + 1 + 8 - x - 1 ) * 16; horMin = ( -ctu - 8 - ( x - g_mcRect[0] ) + 1 ) * 16;
    verMax = ( g_mcRect[3] + 1 + 8 - y - 1 ) * 16; verMin = ( -ctu - 8 - ( y - g_mcRect[1] ) + 1 ) * 16;
  }
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

static void bdof_border_src( const vvo_src* ref, int bx, int by, int w, int h, int mvx, int mvy, int bd, pel* blk );
static void bdof_border( const vvo_planes* ref, int bx, int by, int w, int h, int mvx, int mvy, int bd, pel* blk /* row 0 of the (w+8)-stride buffer */ )
{
  vvo_src s; memset( &s, 0, sizeof( s ) ); s.ref = ref; s.comp = 0;
  bdof_border_src( &s, bx, by, w, h, mvx, mvy, bd, blk );
}
static void bdof_border_src( const vvo_src* ref, int bx, int by, int w, int h, int mvx, int mvy, int bd, pel* blk )
{
  const int S = w + 8;
  const int shift = IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2;
  const int x0 = bx + ( mvx >> 4 ), y0 = by + ( mvy >> 4 );
  const int xOff = ( mvx & 15 ) < 8 ? 1 : 0, yOff = ( mvy & 15 ) < 8 ? 1 : 0;
  for( int r = 0; r < h; r++ )
  {
    pel* d = blk + ( 2 + r ) * S;
    d[0]     = (pel) ( src_at( ref, x0 - xOff,         y0 + 1 - yOff + r ) * ( 1 << shift ) - (pel) IF_INTERNAL_OFFS );
    d[w + 1] = (pel) ( src_at( ref, x0 - xOff + w + 1, y0 + 1 - yOff + r ) * ( 1 << shift ) - (pel) IF_INTERNAL_OFFS );
  }
  for( int i = 0; i < w + 2; i++ )
  {
    blk[1 * S + i]         = (pel) ( src_at( ref, x0 - xOff + i, y0 - yOff )         * ( 1 << shift ) - (pel) IF_INTERNAL_OFFS );
    blk[( h + 2 ) * S + i] = (pel) ( src_at( ref, x0 - xOff + i, y0 + h + 1 - yOff ) * ( 1 << shift ) - (pel) IF_INTERNAL_OFFS );
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

/* ---------------------------------------------------------------------------------------------------------------------
 * DMVR: InterPrediction::xProcessDMVR (InterPrediction.cpp:1847), xinitMC (:1804), xBIPMVRefine (:1702),
 * xDMVRSubPixelErrorSurface (:1785), xSubPelErrorSrfc (:1647), div_for_maxq7 (:1612), xPrefetchPad (:1525),
 * prefetchPadCore / paddingCore (:283-315), xFinalPaddedMCForDMVR (:1731); SADs: RdCost::xGetSAD8/16 and X5 (RdCost.cpp:107-220);
 * bilinear filter: InterpolationFilter::filter<2> (InterpolationFilter.cpp:589-600), filterCopy biMCForDMVR (:445-477). */
static int32_t g_dmvr_out[2 * 65536]; static uint32_t g_dmvr_count;

/* bilinear prediction at IF_INTERNAL_PREC_BILINEAR = 10 bit of a w x h block at integer position (x0, y0) + frac */
static void bilinear_block( const vvo_planes* ref, int x0, int y0, int xFrac, int yFrac, int w, int h, int bd, pel* dst, int dstStride )
{
  const int shiftF = 4 - ( 10 - bd ), offF = shiftF > 0 ? 1 << ( shiftF - 1 ) : 0;
  if( !xFrac && !yFrac ) { for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ ) dst[y * dstStride + x] = (pel) ( ref_at( ref, 0, x0 + x, y0 + y ) * ( 1 << ( 10 - bd ) ) ); return; }
  if( !yFrac || !xFrac )
  {
    const int f = yFrac ? yFrac : xFrac, dx = yFrac ? 0 : 1, dy = yFrac ? 1 : 0;
    for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
      dst[y * dstStride + x] = (pel) ( ( ref_at( ref, 0, x0 + x, y0 + y ) * ( 16 - f ) + ref_at( ref, 0, x0 + x + dx, y0 + y + dy ) * f + offF ) >> shiftF );
    return;
  }
  pel tmp[( 128 + 4 + 1 ) * ( 128 + 4 )];
  for( int y = 0; y < h + 1; y++ ) for( int x = 0; x < w; x++ )
    tmp[y * w + x] = (pel) ( ( ref_at( ref, 0, x0 + x, y0 + y ) * ( 16 - xFrac ) + ref_at( ref, 0, x0 + x + 1, y0 + y ) * xFrac + offF ) >> shiftF );
  for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
    dst[y * dstStride + x] = (pel) ( ( tmp[y * w + x] * ( 16 - yFrac ) + tmp[( y + 1 ) * w + x] * yFrac + 8 ) >> 4 );
}

static uint64_t dmvr_sad( const pel* a, const pel* b, int stride, int w, int h )     /* subShift = 1: even rows, << 1 */
{
  uint64_t s = 0;
  for( int y = 0; y < h; y += 2 ) for( int x = 0; x < w; x++ ) s += (uint64_t) vvo_abs( a[y * stride + x] - b[y * stride + x] );
  return s << 1;
}

static int32_t div_for_maxq7( int64_t N, int64_t D )
{
  int32_t sign = 0, q = 0;
  if( N < 0 ) { sign = 1; N = -N; }
  D = D << 3;
  if( N >= D ) { N -= D; q++; }
  q = q << 1;
  D = D >> 1;
  if( N >= D ) { N -= D; q++; }
  q = q << 1;
  if( N >= ( D >> 1 ) ) q++;
  return sign ? -q : q;
}

static void dmvr_subpel( const uint64_t sad[5], int32_t delta[2] )
{
  int64_t num, den;
  num = (int64_t) ( sad[1] - sad[3] ) * ( (int64_t) 1 << 4 );
  den = (int64_t) ( sad[1] + sad[3] - ( sad[0] << 1 ) );
  if( den != 0 )
  {
    if( sad[1] != sad[0] && sad[3] != sad[0] ) delta[0] = div_for_maxq7( num, den );
    else delta[0] = sad[1] == sad[0] ? -8 : 8;
  }
  num = (int64_t) ( ( sad[2] - sad[4] ) << 4 );
  den = (int64_t) ( sad[2] + sad[4] - ( sad[0] << 1 ) );
  if( den != 0 )
  {
    if( sad[2] != sad[0] && sad[4] != sad[0] ) delta[1] = div_for_maxq7( num, den );
    else delta[1] = sad[2] == sad[0] ? -8 : 8;
  }
}

/* xPrefetchPad + prefetchPadCore: copy of the (w + ntaps - 1)^2 window at the (clipped) start MV, replicated `pad` samples outwards.
 * The copy is addressed in coordinates relative to the sub-block origin displaced by the integer start MV. */
static void dmvr_prefetch( const vvo_planes* ref, int comp, int sx, int sy /* luma pos of the sub-block */, int w, int h /* component size */,
                           const int mergeMv[2], int W, int H, int ctu, pel* pad, int* padStride, int* originX, int* originY )
{
  const int cs = comp ? 1 : 0, sh = 4 + cs, ntaps = comp ? 4 : 8, half = ntaps / 2 - 1, padSize = comp ? 1 : 2;
  int mv[2] = { mergeMv[0] - ( half << sh ), mergeMv[1] - ( half << sh ) };
  if( g_wrapOff ) g_wrapFetch = wrap_clip_mv( mv, sx, sy, w << cs, W, H, ctu );      /* one call here (:1551): the ordinary copy after a move by one period */
  else clip_mv_w( mv, sx, sy, w << cs, W, H, ctu );
  const int px = ( sx >> cs ) + ( mv[0] >> sh ), py = ( sy >> cs ) + ( mv[1] >> sh );      /* top-left of the copied window in the reference */
  const int cw = w + ntaps - 1, chh = h + ntaps - 1;
  const int stride = w + 4 + ntaps;                                                        /* width + 2 * DMVR_NUM_ITERATION + filtersize */
  for( int y = -padSize; y < chh + padSize; y++ ) for( int x = -padSize; x < cw + padSize; x++ )
    pad[( y + 2 ) * stride + ( x + 2 )] = (pel) ref_at( ref, comp, px + vvo_clip3( 0, cw - 1, x ), py + vvo_clip3( 0, chh - 1, y ) );
  *padStride = stride; *originX = 2 + half; *originY = 2 + half;    /* pad coordinates of the block's integer-sample origin for a zero integer delta */
}

static int dmvr_cu( const vvr_picture* pic, const vvr_cu* cu, const vvo_planes* refs, vvo_planes* reco, int bio )
{
  const vvr_pic_header* H = &pic->hdr;
  const int bd = H->bit_depth, ctu = 1 << H->log2_ctu, W = H->width, Hh = H->height;
  const int ncomp = H->chroma_format ? 3 : 1;
  const int altHpel = cu->imv == 3;
  const vvo_planes* ref[2] = { &refs[H->ref_slot[0][cu->ref_idx[0]]], &refs[H->ref_slot[1][cu->ref_idx[1]]] };
  const int mergeMv[2][2] = { { cu->mv[0][0][0], cu->mv[0][0][1] }, { cu->mv[1][0][0], cu->mv[1][0][1] } };
  /* xinitMC: bilinear prediction of the whole CU extended by 2 samples, start MVs clipped against the CU */
  const int ew = cu->w + 4, eh = cu->h + 4;
  pel* bil[2]; bil[0] = (pel*) malloc( sizeof( pel ) * (size_t) ew * eh * 2 ); bil[1] = bil[0] + (size_t) ew * eh;
  for( int l = 0; l < 2; l++ )
  {
    int mv[2] = { mergeMv[l][0], mergeMv[l][1] };
    clip_mv_w( mv, cu->x, cu->y, cu->w, W, Hh, ctu );
    mv[0] -= 2 << 4; mv[1] -= 2 << 4;
    bilinear_block( ref[l], cu->x + ( mv[0] >> 4 ), cu->y + ( mv[1] >> 4 ), mv[0] & 15, mv[1] & 15, ew, eh, bd, bil[l], ew );
  }
  const int dx = vvo_min( cu->w, 16 ), dy = vvo_min( cu->h, 16 );
  int num = 0;
  for( int ys = 0; ys < cu->h; ys += dy ) for( int xs = 0; xs < cu->w; xs += dx, num++ )
  {
    const int sx = cu->x + xs, sy = cu->y + ys;
    const pel* c0 = bil[0] + ( 2 + ys ) * ew + 2 + xs; const pel* c1 = bil[1] + ( 2 + ys ) * ew + 2 + xs;
    const int bst = ew;      /* (xinitMC runs ONCE per CU, InterPrediction.cpp:1859 - also with wrap-around: the start MVs are clipped against the CU, not the sub-block.  Until round 4
                               * this was done per sub-block under wrap-around, which gives the same samples as long as no clamp is involved - every generated picture - and other
                               * ones for vectors beyond a wrap period: found with parsed streams, tools/fuzz_dropin_on_the_oracle.py) */
    uint64_t minCost = dmvr_sad( c0, c1, bst, dx, dy );
    minCost >>= 1; minCost -= minCost >> 2;
    int mv[2][2] = { { mergeMv[0][0], mergeMv[0][1] }, { mergeMv[1][0], mergeMv[1][1] } };
    int16_t total[2] = { 0, 0 };
    int refined = 0;
    if( !( minCost < (uint64_t) ( dx * dy ) ) )
    {
      uint64_t sads[25]; int16_t d[2] = { 0, 0 };
      sads[12] = minCost;
      for( int ver = -2; ver <= 2; ver++ ) for( int hor = -2; hor <= 2; hor++ )
      {
        if( !( ver == 0 && hor == 0 ) ) sads[( ver + 2 ) * 5 + hor + 2] = dmvr_sad( c0 + ver * bst + hor, c1 - ver * bst - hor, bst, dx, dy ) >> 1;
        const uint64_t cost = sads[( ver + 2 ) * 5 + hor + 2];
        if( cost < minCost ) { minCost = cost; d[0] = (int16_t) hor; d[1] = (int16_t) ver; }
      }
      total[0] = (int16_t) ( d[0] * 16 ); total[1] = (int16_t) ( d[1] * 16 );
      if( vvo_abs( total[0] ) != 32 && vvo_abs( total[1] ) != 32 )
      {
        const int ci = ( d[1] + 2 ) * 5 + d[0] + 2;
        const uint64_t sb[5] = { sads[ci], sads[ci - 1], sads[ci - 5], sads[ci + 1], sads[ci + 5] };
        int32_t t[2] = { 0, 0 };
        dmvr_subpel( sb, t );
        total[0] = (int16_t) ( total[0] + t[0] ); total[1] = (int16_t) ( total[1] + t[1] );
      }
      for( int k = 0; k < 2; k++ )
      {
        mv[0][k] = vvo_clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, mergeMv[0][k] + total[k] );
        mv[1][k] = vvo_clip3( -( 1 << 17 ), ( 1 << 17 ) - 1, mergeMv[1][k] - total[k] );
      }
      refined = 1;
    }
    if( g_dmvr_count <= cu->dmvr_off + num ) g_dmvr_count = cu->dmvr_off + num + 1;
    if( cu->dmvr_off + num < 65536 ) { g_dmvr_out[2 * ( cu->dmvr_off + num )] = total[0]; g_dmvr_out[2 * ( cu->dmvr_off + num ) + 1] = total[1]; }
    const int bioSub = minCost < (uint64_t) ( 2 * dx * dy ) ? 0 : bio;
    /* xFinalPaddedMCForDMVR + xWeightedAverage on the sub-block */
    pel blk[2][( 16 + 4 ) * BIO_STRIDE_MAX];
    pel pr[2][3][16 * 16];
    memset( blk, 0, sizeof( blk ) );
    for( int l = 0; l < 2; l++ )
    {
      int cmv[2] = { mv[l][0], mv[l][1] };
      clip_mv_w( cmv, sx, sy, dx, W, Hh, ctu );                                 /* clipped against the SUB-block (:1752) */
      for( int c = 0; c < ncomp; c++ )
      {
        const int cs = c ? 1 : 0, sh = 4 + cs, w = dx >> cs, h = dy >> cs;
        const int dIntX = ( mv[l][0] >> sh ) - ( mergeMv[l][0] >> sh ), dIntY = ( mv[l][1] >> sh ) - ( mergeMv[l][1] >> sh );
        pel* dst = ( c == 0 && bioSub ) ? blk[l] + 2 * ( dx + 8 ) + 1 : pr[l][c];
        const int dstStride = ( c == 0 && bioSub ) ? dx + 8 : w;
        if( refined && ( dIntX || dIntY ) )
        {
          pel pad[( 16 + 4 + 8 ) * ( 16 + 4 + 8 )]; int pst, ox, oy;
          dmvr_prefetch( ref[l], c, sx, sy, w, h, mergeMv[l], W, Hh, ctu, pad, &pst, &ox, &oy );
          vvo_src s; memset( &s, 0, sizeof( s ) ); s.pad = pad; s.padStride = pst; s.padX0 = -( ox + dIntX ); s.padY0 = -( oy + dIntY );
          const int fmx = cmv[0] & ( ( 1 << sh ) - 1 ), fmy = cmv[1] & ( ( 1 << sh ) - 1 );
          pred_block_src( &s, c, 0, 0, w, h, fmx, fmy, 1, altHpel, bd, dst, dstStride );
          if( c == 0 && bioSub ) bdof_border_src( &s, 0, 0, w, h, fmx, fmy, bd, blk[l] );
        }
        else
        {
          if( g_wrapOff ) g_wrapFetch = 1;                                        /* (the prefetch of another component may have chosen the ordinary copy) */
          pred_block( ref[l], c, sx >> cs, sy >> cs, w, h, cmv[0], cmv[1], 1, altHpel, bd, dst, dstStride );
          if( c == 0 && bioSub ) bdof_border( ref[l], sx, sy, w, h, cmv[0], cmv[1], bd, blk[l] );
        }
      }
    }
    for( int c = 0; c < ncomp; c++ )
    {
      const int cs = c ? 1 : 0, w = dx >> cs, h = dy >> cs;
      pel* dst = reco->p[c] + (size_t) ( sy >> cs ) * reco->stride[c] + ( sx >> cs );
      if( c == 0 && bioSub ) { bdof_apply( blk[0], blk[1], w, h, bd, dst, reco->stride[c] ); continue; }
      const int shift = ( IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2 ) + 1, offset = ( 1 << ( shift - 1 ) ) + 2 * IF_INTERNAL_OFFS;
      for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
        dst[y * reco->stride[c] + x] = (pel) vvo_clip_pel( ( pr[0][c][y * w + x] + pr[1][c][y * w + x] + offset ) >> shift, bd );
    }
  }
  free( bil[0] );
  return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Affine motion compensation + PROF: InterPrediction::xPredAffineBlk (InterPrediction.cpp:934-1288),
 * isSubblockVectorSpreadOverLimit (:892), applyPROFCore (:61), gradFilterCore<false> (:213), roundAffineMv (Mv.cpp:57).
 * Sub-block MVs are read from the motion field (the parser side stores them, PU::setAllAffineMv UnitTools.cpp:2689). */
static void round_affine_mv( int* mx, int* my, int sh ) { const int o = 1 << ( sh - 1 ); *mx = ( *mx + o - ( *mx >= 0 ) ) >> sh; *my = ( *my + o - ( *my >= 0 ) ) >> sh; }

static int affine_spread_over_limit( int a, int b, int c, int d, int predType )
{
  const int s4 = 4 << 11, filterTap = 6;
  if( predType == 3 )
  {
    int rw = vvo_max( vvo_max( 0, 4 * a + s4 ), vvo_max( 4 * c, 4 * a + 4 * c + s4 ) ) - vvo_min( vvo_min( 0, 4 * a + s4 ), vvo_min( 4 * c, 4 * a + 4 * c + s4 ) );
    int rh = vvo_max( vvo_max( 0, 4 * b ), vvo_max( 4 * d + s4, 4 * b + 4 * d + s4 ) ) - vvo_min( vvo_min( 0, 4 * b ), vvo_min( 4 * d + s4, 4 * b + 4 * d + s4 ) );
    rw = ( rw >> 11 ) + filterTap + 3; rh = ( rh >> 11 ) + filterTap + 3;
    return rw * rh > ( filterTap + 9 ) * ( filterTap + 9 );
  }
  int rw = vvo_max( 0, 4 * a + s4 ) - vvo_min( 0, 4 * a + s4 ), rh = vvo_max( 0, 4 * b ) - vvo_min( 0, 4 * b );
  rw = ( rw >> 11 ) + filterTap + 3; rh = ( rh >> 11 ) + filterTap + 3;
  if( rw * rh > ( filterTap + 9 ) * ( filterTap + 5 ) ) return 1;
  rw = vvo_max( 0, 4 * c ) - vvo_min( 0, 4 * c ); rh = vvo_max( 0, 4 * d + s4 ) - vvo_min( 0, 4 * d + s4 );
  rw = ( rw >> 11 ) + filterTap + 3; rh = ( rh >> 11 ) + filterTap + 3;
  return rw * rh > ( filterTap + 5 ) * ( filterTap + 9 );
}

/* one list of an affine CU into dst[c] (component-sized buffers, stride = component width); bi = 1 keeps 14-bit samples */
static void affine_list( const vvr_picture* pic, const vvr_cu* cu, int l, const vvo_planes* ref, int bi, pel* const dst[3] )
{
  const vvr_pic_header* H = &pic->hdr;
  const int bd = H->bit_depth, ctu = 1 << H->log2_ctu, w4 = ( H->width + 3 ) >> 2;
  const int ncomp = H->chroma_format ? 3 : 1;
  const int shift = 7;                                               /* MAX_CU_DEPTH */
  const int lw = vvo_log2( cu->w ), lh = vvo_log2( cu->h );
  const int dHX = ( cu->mv[l][1][0] - cu->mv[l][0][0] ) * ( 1 << ( shift - lw ) ), dHY = ( cu->mv[l][1][1] - cu->mv[l][0][1] ) * ( 1 << ( shift - lw ) );
  int dVX, dVY;
  if( cu->flags & VVR_CU_AFFINE_6P ) { dVX = ( cu->mv[l][2][0] - cu->mv[l][0][0] ) * ( 1 << ( shift - lh ) ); dVY = ( cu->mv[l][2][1] - cu->mv[l][0][1] ) * ( 1 << ( shift - lh ) ); }
  else { dVX = -dHY; dVY = dHX; }
  const int over = affine_spread_over_limit( dHX, dHY, dVX, dVY, cu->inter_dir );
  const int sixP = ( cu->flags & VVR_CU_AFFINE_6P ) != 0;
  const int eqRT = cu->mv[l][0][0] == cu->mv[l][1][0] && cu->mv[l][0][1] == cu->mv[l][1][1];
  const int eqLB = cu->mv[l][0][0] == cu->mv[l][2][0] && cu->mv[l][0][1] == cu->mv[l][2][1];
  int prof = ( H->tool_flags & VVR_TOOL_PROF ) != 0;
  prof &= !( ( sixP && eqRT && eqLB ) || ( !sixP && eqRT ) );
  prof &= !over;
  const vvr_rpr_ref* rr = rpr_of( l, cu->ref_idx[l] );
  prof &= !rr;                                                       /* enablePROF &= !refPicScaled (:1029) */
  int dMvH[16], dMvV[16];
  if( prof )
  {
    const int qHX = dHX * 4, qHY = dHY * 4, qVX = dVX * 4, qVY = dVY * 4;
    dMvH[0] = ( ( dHX + dVX ) * 2 ) - ( ( qHX + qVX ) * 2 );
    dMvV[0] = ( ( dHY + dVY ) * 2 ) - ( ( qHY + qVY ) * 2 );
    for( int x = 1; x < 4; x++ ) { dMvH[x] = dMvH[x - 1] + qHX; dMvV[x] = dMvV[x - 1] + qHY; }
    for( int y = 1; y < 4; y++ ) for( int x = 0; x < 4; x++ ) { dMvH[y * 4 + x] = dMvH[( y - 1 ) * 4 + x] + qVX; dMvV[y * 4 + x] = dMvV[( y - 1 ) * 4 + x] + qVY; }
    for( int i = 0; i < 16; i++ ) { round_affine_mv( &dMvH[i], &dMvV[i], 8 ); dMvH[i] = vvo_clip3( -31, 31, dMvH[i] ); dMvV[i] = vvo_clip3( -31, 31, dMvV[i] ); }
  }
  int horMax = ( H->width + 8 - cu->x - 1 ) * 16, horMin = ( -ctu - 8 - cu->x + 1 ) * 16;
  int verMax = ( H->height + 8 - cu->y - 1 ) * 16, verMin = ( -ctu - 8 - cu->y + 1 ) * 16;
  if( g_mcRectOn )
  {   /* clipSubPic: clipMvInSubpic against the CU (:1188-1193) */
    horMax = ( g_mcRect[2] + 1 + 8 - cu->x - 1 ) * 16; horMin = ( -ctu - 8 - ( cu->x - g_mcRect[0] ) + 1 ) * 16;
    verMax = ( g_mcRect[3] + 1 + 8 - cu->y - 1 ) * 16; verMin = ( -ctu - 8 - ( cu->y - g_mcRect[1] ) + 1 ) * 16;
  }
  const int headroom = IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2;
  for( int c = 0; c < ncomp; c++ )
  {
    const int cs = c ? 1 : 0, sh = 4 + cs;
    const int cw = cu->w >> cs, chh = cu->h >> cs;
    for( int y = 0; y < chh; y += 4 ) for( int x = 0; x < cw; x += 4 )
    {
      int mx, my;
      if( c == 0 )
      {
        const vvr_motion* m = &pic->motion[(size_t) ( ( cu->y + y ) >> 2 ) * w4 + ( ( cu->x + x ) >> 2 )];
        mx = m->mv[l][0]; my = m->mv[l][1];
      }
      else
      {   /* 4:2:0: sum of the luma sub-block MVs at (0,0) and (1,1) of the 2x2 group, halved with rounding (:1156-1176) */
        const int lx = cu->x + 2 * x, ly = cu->y + 2 * y;
        const vvr_motion* m0 = &pic->motion[(size_t) ( ly >> 2 ) * w4 + ( lx >> 2 )];
        const vvr_motion* m1 = &pic->motion[(size_t) ( ( ly >> 2 ) + 1 ) * w4 + ( lx >> 2 ) + 1];
        mx = m0->mv[l][0] + m1->mv[l][0]; my = m0->mv[l][1] + m1->mv[l][1];
        round_affine_mv( &mx, &my, 1 );
      }
      if( g_wrapOff )
      {   /* one wrapClipMv per sub-block, against the sub-block (luma position, luma size; :1177-1186) */
        int t[2] = { mx, my };
        g_wrapFetch = wrap_clip_mv( t, cu->x + ( x << cs ), cu->y + ( y << cs ), 4 << cs, H->width, H->height, ctu );
        mx = t[0]; my = t[1];
      }
      else { mx = vvo_min( horMax, vvo_max( horMin, mx ) ); my = vvo_min( verMax, vvo_max( verMin, my ) ); }
      const int bx = ( cu->x >> cs ) + x, by = ( cu->y >> cs ) + y;
      pel* d = dst[c] + y * cw + x;
      if( c == 0 && prof )
      {
        pel ext[6 * 6], gX[16], gY[16];
        pred_block( ref, 0, bx, by, 4, 4, mx, my, 1, 0, bd, ext + 6 + 1, 6 );
        const int xFrac = mx & 15, yFrac = my & 15, xOff = xFrac >> 3, yOff = yFrac >> 3;
        const int x0 = bx + ( mx >> 4 ), y0 = by + ( my >> 4 );
        for( int j = 0; j < 6; j++ ) for( int i = 0; i < 6; i++ )
        {
          if( i >= 1 && i <= 4 && j >= 1 && j <= 4 ) continue;
          if( ( i == 0 || i == 5 ) && ( j == 0 || j == 5 ) && 0 ) continue;
          ext[j * 6 + i] = (pel) ( ref_at( ref, 0, x0 + i - 1 + xOff, y0 + j - 1 + yOff ) * ( 1 << headroom ) - (pel) IF_INTERNAL_OFFS );
        }
        for( int yy = 0; yy < 4; yy++ ) for( int xx = 0; xx < 4; xx++ )
        {
          const pel* sp = ext + ( 1 + yy ) * 6 + 1 + xx;
          gY[yy * 4 + xx] = (pel) ( ( sp[6] >> 6 ) - ( sp[-6] >> 6 ) );
          gX[yy * 4 + xx] = (pel) ( ( sp[1] >> 6 ) - ( sp[-1] >> 6 ) );
        }
        const int dILimit = 1 << vvo_max( bd + 1, 13 );
        const int offset = ( 1 << ( headroom - 1 ) ) + IF_INTERNAL_OFFS;
        for( int yy = 0; yy < 4; yy++ ) for( int xx = 0; xx < 4; xx++ )
        {
          int dI = dMvH[yy * 4 + xx] * gX[yy * 4 + xx] + dMvV[yy * 4 + xx] * gY[yy * 4 + xx];
          dI = vvo_clip3( -dILimit, dILimit - 1, dI );
          pel v = (pel) ( ext[( 1 + yy ) * 6 + 1 + xx] + dI );
          if( !bi ) { v = (pel) ( ( v + offset ) >> headroom ); v = (pel) vvo_clip_pel( v, bd ); }
          d[yy * cw + xx] = v;
        }
      }
      else if( rr ) pred_block_rpr( ref, rr, c, bx, by, 4, 4, mx, my, bi, 0, 2, bd, d, cw );      /* :1200-1204, filterIndex 2 */
      else pred_block( ref, c, bx, by, 4, 4, mx, my, bi, 0, bd, d, cw );
    }
  }
}

static int affine_cu( const vvr_picture* pic, const vvr_cu* cu, const vvo_planes* refs, vvo_planes* reco )
{
  const vvr_pic_header* H = &pic->hdr;
  const int bd = H->bit_depth, ncomp = H->chroma_format ? 3 : 1;
  int biPred = cu->ref_idx[0] >= 0 && cu->ref_idx[1] >= 0;
  /* xCheckIdenticalMotion (InterPrediction.cpp:404-436): same reference picture and same control-point MVs -> list 0 only */
  if( biPred && H->ref_poc[0][cu->ref_idx[0]] == H->ref_poc[1][cu->ref_idx[1]]
      && cu->mv[0][0][0] == cu->mv[1][0][0] && cu->mv[0][0][1] == cu->mv[1][0][1] && cu->mv[0][1][0] == cu->mv[1][1][0] && cu->mv[0][1][1] == cu->mv[1][1][1]
      && ( !( cu->flags & VVR_CU_AFFINE_6P ) || ( cu->mv[0][2][0] == cu->mv[1][2][0] && cu->mv[0][2][1] == cu->mv[1][2][1] ) )
      && !( H->tool_flags & VVR_TOOL_WP ) /* :408 */ ) biPred = 0;
  const int wp = wp_on( pic, cu->bcw_idx );
  const size_t n = (size_t) cu->w * cu->h;
  pel* buf = (pel*) malloc( sizeof( pel ) * n * 3 );
  pel* p0[3] = { buf, buf + n, buf + n + n / 4 };
  pel* p1[3] = { buf + n + n / 2, buf + 2 * n + n / 2, buf + 2 * n + n / 2 + n / 4 };
  if( !pic->motion ) { free( buf ); vvo_set_error( "affine CU without a motion field" ); return -1; }
  if( biPred )
  {
    affine_list( pic, cu, 0, &refs[H->ref_slot[0][cu->ref_idx[0]]], 1, p0 );
    affine_list( pic, cu, 1, &refs[H->ref_slot[1][cu->ref_idx[1]]], 1, p1 );
  }
  else
  {
    const int l = cu->ref_idx[0] >= 0 ? 0 : 1;
    affine_list( pic, cu, l, &refs[H->ref_slot[l][cu->ref_idx[l]]], wp, p0 );
  }
  for( int c = 0; c < ncomp; c++ )
  {
    const int cs = c ? 1 : 0, w = cu->w >> cs, h = cu->h >> cs;
    pel* dst = reco->p[c] + (size_t) ( cu->y >> cs ) * reco->stride[c] + ( cu->x >> cs );
    for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
    {
      int v;
      if( wp ) { const int l = cu->ref_idx[0] >= 0 ? 0 : 1; v = biPred ? wp_bi( pic, cu->ref_idx[0], cu->ref_idx[1], c, p0[c][y * w + x], p1[c][y * w + x] ) : wp_uni( pic, l, cu->ref_idx[l], c, p0[c][y * w + x] ); }
      else if( !biPred ) v = p0[c][y * w + x];
      else if( cu->bcw_idx != 2 )
      {
        const int w1 = vvc_bcw_weights[cu->bcw_idx], w0 = 8 - w1;
        const int shift = ( IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2 ) + 3, offset = ( 1 << ( shift - 1 ) ) + ( IF_INTERNAL_OFFS << 3 );
        v = vvo_clip_pel( ( p0[c][y * w + x] * w0 + p1[c][y * w + x] * w1 + offset ) >> shift, bd );
      }
      else
      {
        const int shift = ( IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2 ) + 1, offset = ( 1 << ( shift - 1 ) ) + 2 * IF_INTERNAL_OFFS;
        v = vvo_clip_pel( ( p0[c][y * w + x] + p1[c][y * w + x] + offset ) >> shift, bd );
      }
      dst[y * reco->stride[c] + x] = (pel) v;
    }
  }
  free( buf );
  return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Geometric partitioning: InterPrediction::motionCompensationGeo (InterPrediction.cpp:1461) = two uni-directional predictions
 * kept at 14 bit (xPredInterBi :711 with cu.geoFlag()), blended by InterpolationFilter::xWeightedGeoBlk
 * (InterpolationFilter.cpp:1217) with the weight masks g_globalGeoWeights (Rom.cpp:519-586). */
static int geo_weight( const vvr_cu* cu, int cs, int x, int y )      /* weight of partition 0 at component sample (x, y) of the CU */
{
  const int MS = 112;                                                 /* GEO_WEIGHT_MASK_SIZE */
  const int angle = vvc_geo_params[cu->geo_split_dir][0];
  const int wIdx = vvo_log2( cu->w ) - 3, hIdx = vvo_log2( cu->h ) - 3;
  const int ox = vvc_geo_weight_offset[cu->geo_split_dir][hIdx][wIdx][0], oy = vvc_geo_weight_offset[cu->geo_split_dir][hIdx][wIdx][1];
  const int8_t* W = vvc_geo_weights[vvc_geo_angle2mask[angle]];
  const int lx = x << cs, ly = y << cs;
  if( vvc_geo_angle2mirror[angle] == 2 ) return W[( MS - 1 - oy - ly ) * MS + ox + lx];
  if( vvc_geo_angle2mirror[angle] == 1 ) return W[( oy + ly ) * MS + ( MS - 1 - ox ) - lx];
  return W[( oy + ly ) * MS + ox + lx];
}

static int geo_cu( const vvr_picture* pic, const vvr_cu* cu, const vvo_planes* refs, int num_slots, vvo_planes* reco )
{
  const vvr_pic_header* H = &pic->hdr;
  const int bd = H->bit_depth, ctu = 1 << H->log2_ctu, ncomp = H->chroma_format ? 3 : 1;
  const size_t n = (size_t) cu->w * cu->h;
  pel* buf = (pel*) malloc( sizeof( pel ) * n * 2 );
  for( int c = 0; c < ncomp; c++ )
  {
    const int cs = c ? 1 : 0, w = cu->w >> cs, h = cu->h >> cs;
    for( int k = 0; k < 2; k++ )
    {
      const int l = ( cu->geo_dir_ref[k] >> 4 ) - 1, ri = cu->geo_dir_ref[k] & 15;
      const int slot = H->ref_slot[l][ri];
      if( l < 0 || l > 1 || slot < 0 || slot >= num_slots || !refs[slot].p[0] ) { free( buf ); vvo_set_error( "GPM: bad reference" ); return -1; }
      int mv[2] = { cu->geo_mv[k][0], cu->geo_mv[k][1] };
      const vvr_rpr_ref* rr = rpr_of( l, ri );
      if( !rr ) clip_mv_w( mv, cu->x, cu->y, cu->w, H->width, H->height, ctu );
      pred_any( &refs[slot], rr, c, cu->x >> cs, cu->y >> cs, w, h, mv[0], mv[1], 1, 0, bd, buf + k * n, w );
    }
    const int shift = ( IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2 ) + 3, offset = ( 1 << ( shift - 1 ) ) + ( IF_INTERNAL_OFFS << 3 );
    pel* dst = reco->p[c] + (size_t) ( cu->y >> cs ) * reco->stride[c] + ( cu->x >> cs );
    for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
    {
      const int wt = geo_weight( cu, cs, x, y );
      dst[y * reco->stride[c] + x] = (pel) vvo_clip_pel( ( wt * buf[y * w + x] + ( 8 - wt ) * buf[n + y * w + x] + offset ) >> shift, bd );
    }
  }
  free( buf );
  return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Sub-block temporal MV prediction: InterPrediction::xSubPuMC (InterPrediction.cpp:438-549): 8x8 sub-blocks, each with the motion
 * stored in the motion field, predicted by the regular uni/bi path (no BDOF, no DMVR: m_subPuMC).  The reference joins sub-blocks
 * with equal motion along the CU's longer side (:466-512), cuts a joined run of more than 16 samples that is not a multiple of 16
 * into its multiple-of-16 part and the rest (:514-538), and predicts each piece as one block: motionCompensation makes the piece
 * m_currCuArea (:1375), so the MV clip refers to the PIECE.  Without wrap-around that changes nothing in the samples (the clamp
 * only acts outside the padded picture); wrapClipMv (Mv.cpp:117) however moves an MV by a period depending on the piece's
 * position and WIDTH, so sbtmvp_cu below forms the same pieces and hands each sub-block the area of its piece. */
static void plain_block( const vvr_picture* pic, const vvo_planes* refs, const vvr_cu* cu /* m_currCuArea: what the MVs are clipped against */, int x, int y, int w, int h, const int mv[2][2], const int ref_idx[2], int bcw_idx, int altHpel, vvo_planes* reco )
{
  const vvr_pic_header* H = &pic->hdr;
  const int bd = H->bit_depth, ctu = 1 << H->log2_ctu, ncomp = H->chroma_format ? 3 : 1;
  int bi = ref_idx[0] >= 0 && ref_idx[1] >= 0;
  if( bi && H->ref_poc[0][ref_idx[0]] == H->ref_poc[1][ref_idx[1]] && mv[0][0] == mv[1][0] && mv[0][1] == mv[1][1] && !( H->tool_flags & VVR_TOOL_WP ) ) bi = 0;     /* xCheckIdenticalMotion */
  const int wp = wp_on( pic, bcw_idx );
  for( int c = 0; c < ncomp; c++ )
  {
    const int cs = c ? 1 : 0, bx = x >> cs, by = y >> cs, bw = w >> cs, bh = h >> cs;
    pel* dst = reco->p[c] + (size_t) by * reco->stride[c] + bx;
    if( !bi )
    {
      const int l = ref_idx[0] >= 0 ? 0 : 1;
      int m[2] = { mv[l][0], mv[l][1] };
      const vvr_rpr_ref* rr = rpr_of( l, ref_idx[l] );
      if( rr ) ; else
      if( g_wrapOff ) clip_mv_w( m, cu->x, cu->y, cu->w, H->width, H->height, ctu ); else clip_mv_w( m, x, y, w, H->width, H->height, ctu );
      if( wp )
      {
        pel t[16 * 16];
        pred_any( &refs[H->ref_slot[l][ref_idx[l]]], rr, c, bx, by, bw, bh, m[0], m[1], 1, altHpel, bd, t, bw );
        for( int yy = 0; yy < bh; yy++ ) for( int xx = 0; xx < bw; xx++ ) dst[yy * reco->stride[c] + xx] = (pel) wp_uni( pic, l, ref_idx[l], c, t[yy * bw + xx] );
      }
      else pred_any( &refs[H->ref_slot[l][ref_idx[l]]], rr, c, bx, by, bw, bh, m[0], m[1], 0, altHpel, bd, dst, reco->stride[c] );
      continue;
    }
    pel t[2][16 * 16];
    for( int l = 0; l < 2; l++ )
    {
      int m[2] = { mv[l][0], mv[l][1] };
      const vvr_rpr_ref* rr = rpr_of( l, ref_idx[l] );
      if( rr ) ; else
      if( g_wrapOff ) clip_mv_w( m, cu->x, cu->y, cu->w, H->width, H->height, ctu ); else clip_mv_w( m, x, y, w, H->width, H->height, ctu );
      pred_any( &refs[H->ref_slot[l][ref_idx[l]]], rr, c, bx, by, bw, bh, m[0], m[1], 1, altHpel, bd, t[l], bw );
    }
    const int hr = IF_INTERNAL_PREC - bd > 2 ? IF_INTERNAL_PREC - bd : 2;
    for( int yy = 0; yy < bh; yy++ ) for( int xx = 0; xx < bw; xx++ )
    {
      int v;
      if( wp ) { dst[yy * reco->stride[c] + xx] = (pel) wp_bi( pic, ref_idx[0], ref_idx[1], c, t[0][yy * bw + xx], t[1][yy * bw + xx] ); continue; }
      if( bcw_idx != 2 ) { const int w1 = vvc_bcw_weights[bcw_idx], w0 = 8 - w1; v = ( t[0][yy * bw + xx] * w0 + t[1][yy * bw + xx] * w1 + ( 1 << ( hr + 2 ) ) + ( IF_INTERNAL_OFFS << 3 ) ) >> ( hr + 3 ); }
      else v = ( t[0][yy * bw + xx] + t[1][yy * bw + xx] + ( 1 << hr ) + 2 * IF_INTERNAL_OFFS ) >> ( hr + 1 );
      dst[yy * reco->stride[c] + xx] = (pel) vvo_clip_pel( v, bd );
    }
  }
}

static int sbtmvp_cu( const vvr_picture* pic, const vvr_cu* cu, const vvo_planes* refs, vvo_planes* reco )
{
  const vvr_pic_header* H = &pic->hdr;
  const int w4 = ( H->width + 3 ) >> 2;
  if( !pic->motion ) { vvo_set_error( "SbTMVP CU without a motion field" ); return -1; }
  /* the pieces xSubPuMC predicts (see above): runs of equal motion along the longer side, cut at the largest multiple of 16 */
  const int verMC = cu->h > cu->w;
  const int nFst = ( verMC ? cu->w : cu->h ) >> 3, nSec = ( verMC ? cu->h : cu->w ) >> 3;
  const int scaled = rpr_of( 0, 0 ) || ( H->num_ref[1] > 0 && rpr_of( 1, 0 ) );      /* :478 RPR_FIX: no joining when the first reference picture of a list is scaled */
  for( int f = 0; f < nFst; f++ )
    for( int s0 = 0; s0 < nSec; )
    {
#define SB_AT( f_, s_ ) ( &pic->motion[(size_t) ( ( cu->y + 8 * ( verMC ? (s_) : (f_) ) ) >> 2 ) * w4 + ( ( cu->x + 8 * ( verMC ? (f_) : (s_) ) ) >> 2 )] )
      const vvr_motion* m = SB_AT( f, s0 );
      const int ri[2] = { m->ref_idx[0], m->ref_idx[1] };
      if( ( ri[0] < 0 && ri[1] < 0 ) || ri[0] >= H->num_ref[0] || ri[1] >= H->num_ref[1] ) { vvo_set_error( "SbTMVP: bad sub-block motion" ); return -1; }
      const int mv[2][2] = { { m->mv[0][0], m->mv[0][1] }, { m->mv[1][0], m->mv[1][1] } };
      int s1 = s0 + 1;
      for( ; s1 < nSec && !scaled; s1++ )      /* MotionInfo::operator== (MotionInfo.h:127): the MV of an unused list does not count */
      {
        const vvr_motion* n = SB_AT( f, s1 );
        if( n->ref_idx[0] != ri[0] || n->ref_idx[1] != ri[1] ) break;
        if( ri[0] >= 0 && ( n->mv[0][0] != mv[0][0] || n->mv[0][1] != mv[0][1] ) ) break;
        if( ri[1] >= 0 && ( n->mv[1][0] != mv[1][0] || n->mv[1][1] != mv[1][1] ) ) break;
      }
#undef SB_AT
      const int len = 8 * ( s1 - s0 );
      const int cut = ( len > 16 && ( len & 15 ) ) ? ( len & ~15 ) : len;          /* first piece; the rest (8 samples) is the second */
      for( int s = s0; s < s1; s++ )
      {
        const int inFirst = 8 * ( s - s0 ) < cut;
        const int p0 = 8 * s0 + ( inFirst ? 0 : cut ), pl = inFirst ? cut : len - cut;
        vvr_cu piece = *cu;
        if( verMC ) { piece.x = (uint16_t) ( cu->x + 8 * f ); piece.w = 8; piece.y = (uint16_t) ( cu->y + p0 ); piece.h = (uint8_t) pl; }
        else        { piece.y = (uint16_t) ( cu->y + 8 * f ); piece.h = 8; piece.x = (uint16_t) ( cu->x + p0 ); piece.w = (uint8_t) pl; }
        plain_block( pic, refs, &piece, cu->x + 8 * ( verMC ? f : s ), cu->y + 8 * ( verMC ? s : f ), 8, 8, mv, ri, cu->bcw_idx, cu->imv == 3, reco );
      }
      s0 = s1;
    }
  return 0;
}

void vvo_dmvr_reset( void ) { g_dmvr_count = 0; memset( g_dmvr_out, 0, sizeof( g_dmvr_out ) ); }
uint32_t vvo_get_dmvr( int32_t* dst, uint32_t max_entries )
{
  const uint32_t n = g_dmvr_count < max_entries ? g_dmvr_count : max_entries;
  if( dst ) memcpy( dst, g_dmvr_out, sizeof( int32_t ) * 2 * n );
  return g_dmvr_count;
}

int vvo_inter_cu( const vvr_picture* pic, const vvr_cu* cu, const vvo_planes* refs, int num_slots, vvo_planes* reco )
{
  const vvr_pic_header* H = &pic->hdr;
  const int bd = H->bit_depth, ctu = 1 << H->log2_ctu;
  const int ncomp = H->chroma_format ? 3 : 1;
  g_wrapOff = H->wrap_offset; g_wrapFetch = 0;
  g_mcRectOn = 0;
  g_rpr = pic->rpr;
  g_wp = vvo_wp_set_at( pic, cu->x, cu->y ); g_wpOn = ( vvo_flags_at( pic, cu->x, cu->y ) & VVR_TOOL_WP ) != 0;
  if( pic->subpics && pic->num_subpics > 1 )
    for( uint32_t k = 0; k < pic->num_subpics; k++ )
    {
      const vvr_subpic* sp = &pic->subpics[k];
      if( cu->x >= sp->x0 && cu->x <= sp->x1 && cu->y >= sp->y0 && cu->y <= sp->y1 && sp->treated_as_pic ) { g_mcRect[0] = sp->x0; g_mcRect[1] = sp->y0; g_mcRect[2] = sp->x1; g_mcRect[3] = sp->y1; g_mcRectOn = 1; }
    }
  if( cu->mc_mode == VVR_MC_AFFINE ) return affine_cu( pic, cu, refs, reco );
  if( cu->mc_mode == VVR_MC_SBTMVP ) return sbtmvp_cu( pic, cu, refs, reco );
  if( cu->mc_mode == VVR_MC_GEO ) return geo_cu( pic, cu, refs, num_slots, reco );
  if( ( cu->mc_mode == VVR_MC_DMVR || cu->mc_mode == VVR_MC_DMVR_BDOF || cu->mc_mode == VVR_MC_BDOF ) && ( rpr_of( 0, cu->ref_idx[0] ) || rpr_of( 1, cu->ref_idx[1] ) ) )
  { vvo_set_error( "BDOF / DMVR CU with a scaled reference picture (InterPrediction.cpp:1431-1435)" ); return -1; }
  if( g_rpr && ( g_wrapOff || g_mcRectOn ) ) { vvo_set_error( "scaled reference pictures with wrap-around / sub-pictures treated as pictures" ); return -1; }
  if( cu->mc_mode == VVR_MC_DMVR || cu->mc_mode == VVR_MC_DMVR_BDOF ) return dmvr_cu( pic, cu, refs, reco, cu->mc_mode == VVR_MC_DMVR_BDOF );
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
      const vvr_rpr_ref* rr = rpr_of( l, cu->ref_idx[l] );
      if( !rr ) clip_mv_w( mv, cu->x, cu->y, cu->w, H->width, H->height, ctu );
      const int slot = H->ref_slot[l][cu->ref_idx[l]];
      if( slot < 0 || slot >= num_slots || !refs[slot].p[0] ) { vvo_set_error( "missing reference slot" ); return -1; }
      if( wp_on( pic, cu->bcw_idx ) )
      {   /* xPredInterUni at 14 bit, then addWeightUni */
        pel* t = (pel*) malloc( sizeof( pel ) * (size_t) w * h );
        pred_any( &refs[slot], rr, c, bx, by, w, h, mv[0], mv[1], 1, altHpel, bd, t, w );
        for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ ) dst[y * reco->stride[c] + x] = (pel) wp_uni( pic, l, cu->ref_idx[l], c, t[y * w + x] );
        free( t );
      }
      else pred_any( &refs[slot], rr, c, bx, by, w, h, mv[0], mv[1], 0, altHpel, bd, dst, reco->stride[c] );
    }
    else
    {
      pel* t0 = (pel*) malloc( sizeof( pel ) * (size_t) w * h * 2 ); pel* t1 = t0 + (size_t) w * h;
      for( int l = 0; l < 2; l++ )
      {
        int mv[2] = { cu->mv[l][0][0], cu->mv[l][0][1] };
        const vvr_rpr_ref* rr = rpr_of( l, cu->ref_idx[l] );
        if( !rr ) clip_mv_w( mv, cu->x, cu->y, cu->w, H->width, H->height, ctu );
        const int slot = H->ref_slot[l][cu->ref_idx[l]];
        if( slot < 0 || slot >= num_slots || !refs[slot].p[0] ) { free( t0 ); vvo_set_error( "missing reference slot" ); return -1; }
        pred_any( &refs[slot], rr, c, bx, by, w, h, mv[0], mv[1], 1, altHpel, bd, l ? t1 : t0, w );
      }
      if( c == 0 && cu->mc_mode == VVR_MC_BDOF )
      {   /* xSubPuBio (:551): sub-blocks of at most 16x16 (MAX_BDOF_APPLICATION_REGION), each with its own border fetch */
        int mv0[2] = { cu->mv[0][0][0], cu->mv[0][0][1] }, mv1[2] = { cu->mv[1][0][0], cu->mv[1][0][1] };
        clip_mv_w( mv0, cu->x, cu->y, cu->w, H->width, H->height, ctu ); clip_mv_w( mv1, cu->x, cu->y, cu->w, H->width, H->height, ctu );
        const int sw = vvo_min( 16, w ), shh = vvo_min( 16, h );
        for( int y = 0; y < h; y += shh ) for( int x = 0; x < w; x += sw )
          bdof_luma_subblock( &refs[H->ref_slot[0][cu->ref_idx[0]]], &refs[H->ref_slot[1][cu->ref_idx[1]]], bx + x, by + y, sw, shh, mv0, mv1, altHpel, bd,
                              dst + (size_t) y * reco->stride[c] + x, reco->stride[c] );
      }
      else if( wp_on( pic, cu->bcw_idx ) )
      {
        for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
          dst[y * reco->stride[c] + x] = (pel) wp_bi( pic, cu->ref_idx[0], cu->ref_idx[1], c, t0[y * w + x], t1[y * w + x] );
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
