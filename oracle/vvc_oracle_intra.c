/* oracle/vvc_oracle_intra.c — CPU restatement (TEST INFRASTRUCTURE): intra prediction + reconstruction of one transform block.
 *
 * Follows  DecoderLib/DecCu.cpp:271-401 (predAndReco, intra branch),
 *          CommonLib/IntraPrediction.cpp:947-964 (initIntraPatternChType), :1072-1249 (xFillReferenceSamples),
 *          :1251-1288 (xFilterReferenceSamples), :1301-1330 (useFilteredIntraRefSamples), :1343-1400 (is{Above,Left}Available),
 *          :412-441 (xGetPredValDc), :154-210 (xPredIntraPlanarCore), :212-233 (IntraPredSampleFilterCore, PDPC planar/DC),
 *          :443-458 (getWideAngle), :474-517 (predIntraAng), :592-848 (xPredIntraAng), :850-885 (xPredIntraBDPCM),
 *          CommonLib/CodingStructure.cpp:464-500 (getCURestricted),  CommonLib/Buffer.cpp:482 (reconstruct).
 *
 * Not restated in this version (the description is rejected): ISP, MIP, CCLM/MDLM. */
#include "vvc_oracle_common.h"
#include "../tables/vvc_tables.inc"

#define MAXREF ( 2 * 64 + 8 )

static const uint8_t intraFilterThr[2][8] = { { 24, 24, 24, 14, 2, 0, 0, 0 }, { 40, 40, 40, 28, 4, 0, 0, 0 } };   /* m_aucIntraFilter (:72) */
static const int angTable[32]    = { 0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 23, 26, 29, 32, 35, 39, 45, 51, 57, 64, 73, 86, 102, 128, 171, 256, 341, 512, 1024 };
static const int invAngTable[32] = { 0, 16384, 8192, 5461, 4096, 2731, 2048, 1638, 1365, 1170, 1024, 910, 819, 712, 630, 565, 512, 468, 420, 364, 321, 287, 256, 224, 191, 161, 128, 96, 64, 48, 32, 16 };
static const int8_t gauss[32][4] = {   /* g_intraGaussFilter (:96) */
  { 16, 32, 16, 0 }, { 16, 32, 16, 0 }, { 15, 31, 17, 1 }, { 15, 31, 17, 1 }, { 14, 30, 18, 2 }, { 14, 30, 18, 2 }, { 13, 29, 19, 3 }, { 13, 29, 19, 3 },
  { 12, 28, 20, 4 }, { 12, 28, 20, 4 }, { 11, 27, 21, 5 }, { 11, 27, 21, 5 }, { 10, 26, 22, 6 }, { 10, 26, 22, 6 }, { 9, 25, 23, 7 }, { 9, 25, 23, 7 },
  { 8, 24, 24, 8 }, { 8, 24, 24, 8 }, { 7, 23, 25, 9 }, { 7, 23, 25, 9 }, { 6, 22, 26, 10 }, { 6, 22, 26, 10 }, { 5, 21, 27, 11 }, { 5, 21, 27, 11 },
  { 4, 20, 28, 12 }, { 4, 20, 28, 12 }, { 3, 19, 29, 13 }, { 3, 19, 29, 13 }, { 2, 18, 30, 14 }, { 2, 18, 30, 14 }, { 1, 17, 31, 15 }, { 1, 17, 31, 15 } };

static int wide_angle( int w, int h, int mode )   /* IntraPrediction::getWideAngle (:443) */
{
  static const int modeShift[] = { 0, 6, 10, 12, 14, 15 };
  if( mode > 1 && mode <= 66 )
  {
    const int d = vvo_abs( vvo_log2( w ) - vvo_log2( h ) );
    if( w > h && mode < 2 + modeShift[d] ) mode += 65;
    else if( h > w && mode > 66 - modeShift[d] ) mode -= 65;
  }
  return mode;
}

/* reference availability of the 4x4 luma-grid unit that contains channel position (x,y): the unit must lie inside the picture and
 * its transform block must precede the current one in decoding order (getCURestricted + the TU index test of is*Available). */
static const vvr_picture* g_pic;      /* the picture being reconstructed (slice / tile maps) */
static int g_curCtu;                  /* CTU of the block whose neighbourhood is looked at */
static int unit_avail( const vvr_pic_header* H, const int32_t* order, int ch, int x, int y, int32_t cur )
{
  const int cs = ch ? 1 : 0;
  const int lx = x << cs, ly = y << cs;
  if( x < 0 || y < 0 || lx >= H->width || ly >= H->height ) return 0;
  const int w4 = ( H->width + 3 ) >> 2, h4 = ( H->height + 3 ) >> 2;
  if( g_pic && !vvo_same_slice_tile( g_pic, vvo_ctu_of( H, lx, ly ), g_curCtu ) ) return 0;      /* getCURestricted: same slice, same tile */
  return order[(size_t) ch * w4 * h4 + ( ly >> 2 ) * w4 + ( lx >> 2 )] < cur;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Cross-component linear model (CCLM / MDLM_L / MDLM_T), 4:2:0, sps_cclm_collocated_chroma_flag = 0:
 * IntraPrediction::xGetLumaRecPixels (IntraPrediction.cpp:1403-1690), xGetLMParameters (:1694-1905), predIntraChromaLM (:519),
 * AreaBuf::linearTransform (Buffer.cpp:516).  top / left = the chroma block's unfiltered reference lines (xFillReferenceSamples).
 * Only the template positions that xGetLMParameters selects are down-sampled (the reference down-samples whole lines, the
 * unselected values never reach the output). */
#define LM_CHROMA_IDX 67
#define MDLM_L_IDX    68
#define MDLM_T_IDX    69
static void cclm_predict( const vvr_picture* pic, const vvr_cu* cu, const vvr_tu* tu, uint32_t tu_idx, int comp, const vvo_planes* reco, const int32_t* order,
                          const pel* top, const pel* left, pel* pred )
{
  const vvr_pic_header* H = &pic->hdr;
  const int bd = H->bit_depth, ctu = 1 << H->log2_ctu;
  /* ISP: the chroma block of the whole CU sits in the last TU */
  const int lx0 = cu->isp_mode ? cu->x : tu->x, ly0 = cu->isp_mode ? cu->y : tu->y;
  const int cw = ( cu->isp_mode ? cu->w : tu->w ) >> 1, chh = ( cu->isp_mode ? cu->h : tu->h ) >> 1, x0 = lx0 >> 1, y0 = ly0 >> 1;
  const pel* Y = reco->p[0]; const int ys = reco->stride[0];
#define LU( xx, yy ) ( (int) Y[(size_t) ( ly0 + ( yy ) ) * ys + lx0 + ( xx )] )
  const int mode = cu->intra_dir[1];
  /* cu.above / cu.left: the neighbouring CU exists in this slice and tile (a block inside its CU has the CU itself above / left) */
  const int curCtu = vvo_ctu_of( H, cu->x, cu->y );
  const int aboveCu = ly0 > cu->y || ( cu->y > 0 && vvo_same_slice_tile( pic, vvo_ctu_of( H, cu->x, cu->y - 1 ), curCtu ) );
  const int leftCu = lx0 > cu->x || ( cu->x > 0 && vvo_same_slice_tile( pic, vvo_ctu_of( H, cu->x - 1, cu->y ), curCtu ) );
  const int unit = 2;                                                                       /* 4 luma samples in chroma units */
  const int tuWU = cw / unit, tuHU = chh / unit;
  /* ---- xGetLumaRecPixels: which borders take part in the edge handling of the down-sampling filters */
  const int totalAboveUnits = mode == MDLM_T_IDX ? ( 2 * cw + unit - 1 ) / unit : tuWU;
  const int totalLeftUnits  = mode == MDLM_L_IDX ? ( 2 * chh + unit - 1 ) / unit : tuHU;
  const int bLeft  = ( leftCu ? totalLeftUnits : 0 ) >= tuHU;
  const int bAbove = ( aboveCu ? totalAboveUnits : 0 ) >= tuWU;
  const int firstRowOfCtu = ( ly0 & ( ctu - 1 ) ) == 0;
  const int colloc = ( H->tool_flags & VVR_TOOL_CCLM_COLLOC ) != 0;
  /* ---- xGetLMParameters: template sizes */
  int aboveAvailable = 0, leftAvailable = 0, actualTop = 0, actualLeft = 0;
  {
    const int totA = ( 2 * cw + unit - 1 ) / unit, totL = ( 2 * chh + unit - 1 ) / unit;
    int aboveRightUnits = totA - tuWU, leftBelowUnits = totL - tuHU;
    if( mode == MDLM_T_IDX )
    {
      int avai = 0;
      if( aboveCu )
      {
        avai = tuWU;
        aboveRightUnits = aboveRightUnits > ( chh / unit ) ? chh / unit : aboveRightUnits;
        int n = 0;
        for( int k = 0; k < aboveRightUnits; k++ ) { if( !unit_avail( H, order, 1, x0 + cw + k * unit, y0 - 1, (int32_t) tu_idx ) ) break; n++; }
        avai += n;
      }
      aboveAvailable = avai >= tuWU; actualTop = unit * avai;
    }
    else if( mode == MDLM_L_IDX )
    {
      int avai = 0;
      if( leftCu )
      {
        avai = tuHU;
        leftBelowUnits = leftBelowUnits > ( cw / unit ) ? cw / unit : leftBelowUnits;
        int n = 0;
        for( int k = 0; k < leftBelowUnits; k++ ) { if( !unit_avail( H, order, 1, x0 - 1, y0 + chh + k * unit, (int32_t) tu_idx ) ) break; n++; }
        avai += n;
      }
      leftAvailable = avai >= tuHU; actualLeft = unit * avai;
    }
    else { aboveAvailable = aboveCu; leftAvailable = leftCu; actualTop = cw; actualLeft = chh; }
  }
  const int aboveIs4 = leftAvailable ? 0 : 1, leftIs4 = aboveAvailable ? 0 : 1;
  const int startPos[2] = { actualTop >> ( 2 + aboveIs4 ), actualLeft >> ( 2 + leftIs4 ) };
  const int pickStep[2] = { vvo_max( 1, actualTop >> ( 1 + aboveIs4 ) ), vvo_max( 1, actualLeft >> ( 1 + leftIs4 ) ) };
  int selL[4] = { 0, 0, 0, 0 }, selC[4] = { 0, 0, 0, 0 };
  int cntT = 0, cntL = 0;
  if( aboveAvailable )
  {
    cntT = vvo_min( actualTop, ( 1 + aboveIs4 ) << 1 );
    for( int k = 0, pos = startPos[0]; k < cntT; pos += pickStep[0], k++ )
    {
      const int i = pos; int v;
      if( firstRowOfCtu )
      {   /* one luma line above: [1 2 1] */
        const int l = ( i == 0 && !bLeft ) ? LU( 2 * i, -1 ) : LU( 2 * i - 1, -1 );
        v = ( LU( 2 * i, -1 ) * 2 + l + LU( 2 * i + 1, -1 ) + 2 ) >> 2;
      }
      else if( colloc )
      {   /* sps_chroma_vertical_collocated_flag: 5-tap cross centred on the luma sample that sits at the chroma position (:1516-1534) */
        const int xl = ( i == 0 && !bLeft ) ? 2 * i : 2 * i - 1;
        v = ( LU( 2 * i, -3 ) + LU( 2 * i, -2 ) * 4 + LU( xl, -2 ) + LU( 2 * i + 1, -2 ) + LU( 2 * i, -1 ) + 4 ) >> 3;
      }
      else
      {   /* two luma lines above: 6-tap */
        const int xl = ( i == 0 && !bLeft ) ? 2 * i : 2 * i - 1;
        v = ( LU( 2 * i, -2 ) * 2 + LU( xl, -2 ) + LU( 2 * i + 1, -2 ) + LU( 2 * i, -1 ) * 2 + LU( xl, -1 ) + LU( 2 * i + 1, -1 ) + 4 ) >> 3;
      }
      selL[k] = (pel) v; selC[k] = top[1 + pos];
    }
  }
  if( leftAvailable )
  {
    cntL = vvo_min( actualLeft, ( 1 + leftIs4 ) << 1 );
    for( int k = 0, pos = startPos[1]; k < cntL; pos += pickStep[1], k++ )
    {
      const int j = pos;
      int v;
      if( colloc ) { const int yu = ( j == 0 && !bAbove ) ? 2 * j : 2 * j - 1; v = ( LU( -2, yu ) + LU( -2, 2 * j ) * 4 + LU( -3, 2 * j ) + LU( -1, 2 * j ) + LU( -2, 2 * j + 1 ) + 4 ) >> 3; }     /* (:1556-1571) */
      else v = ( LU( -2, 2 * j ) * 2 + LU( -3, 2 * j ) + LU( -1, 2 * j ) + LU( -2, 2 * j + 1 ) * 2 + LU( -3, 2 * j + 1 ) + LU( -1, 2 * j + 1 ) + 4 ) >> 3;
      selL[k + cntT] = (pel) v; selC[k + cntT] = left[1 + pos];
    }
  }
  const int cnt = cntL + cntT;
  if( cnt == 2 )
  {
    selL[3] = selL[0]; selC[3] = selC[0]; selL[2] = selL[1]; selC[2] = selC[1];
    selL[0] = selL[1]; selC[0] = selC[1]; selL[1] = selL[3]; selC[1] = selC[3];
  }
  int minGrp[2] = { 0, 2 }, maxGrp[2] = { 1, 3 };
  int *tmin = minGrp, *tmax = maxGrp;
  if( selL[tmin[0]] > selL[tmin[1]] ) { const int t = tmin[0]; tmin[0] = tmin[1]; tmin[1] = t; }
  if( selL[tmax[0]] > selL[tmax[1]] ) { const int t = tmax[0]; tmax[0] = tmax[1]; tmax[1] = t; }
  if( selL[tmin[0]] > selL[tmax[1]] ) { int* t = tmin; tmin = tmax; tmax = t; }
  if( selL[tmin[1]] > selL[tmax[0]] ) { const int t = tmin[1]; tmin[1] = tmax[0]; tmax[0] = t; }
  const int minL = ( selL[tmin[0]] + selL[tmin[1]] + 1 ) >> 1, minC = ( selC[tmin[0]] + selC[tmin[1]] + 1 ) >> 1;
  const int maxL = ( selL[tmax[0]] + selL[tmax[1]] + 1 ) >> 1, maxC = ( selC[tmax[0]] + selC[tmax[1]] + 1 ) >> 1;
  int a, b, shift;
  if( leftAvailable || aboveAvailable )
  {
    const int diff = maxL - minL;
    if( diff > 0 )
    {
      static const uint8_t DivSigTable[16] = { 0, 7, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1, 1, 1, 0 };
      const int diffC = maxC - minC;
      int x = vvo_floor_log2( diff );
      const int normDiff = ( diff << 4 >> x ) & 15;
      const int v = DivSigTable[normDiff] | 8;
      x += normDiff != 0;
      const int y = diffC == 0 ? 0 : vvo_floor_log2( vvo_abs( diffC ) ) + 1;
      const int add = 1 << y >> 1;
      a = ( diffC * v + add ) >> y;
      shift = 3 + x - y;
      if( shift < 1 ) { shift = 1; a = a == 0 ? 0 : a < 0 ? -15 : 15; }
      b = minC - ( ( a * minL ) >> shift );
    }
    else { a = 0; b = minC; shift = 0; }
  }
  else { a = 0; b = 1 << ( bd - 1 ); shift = 0; }
  /* ---- down-sampled luma of the block + linear model */
  for( int y = 0; y < chh; y++ ) for( int x = 0; x < cw; x++ )
  {
    const int xl = ( x == 0 && !bLeft ) ? 0 : 2 * x - 1;
    const int yu = ( y == 0 && !bAbove ) ? 0 : 2 * y - 1;
    const int t = colloc ? (pel) ( ( LU( 2 * x, yu ) + LU( 2 * x, 2 * y ) * 4 + LU( xl, 2 * y ) + LU( 2 * x + 1, 2 * y ) + LU( 2 * x, 2 * y + 1 ) + 4 ) >> 3 )       /* (:1588-1625) */
                       : (pel) ( ( LU( 2 * x, 2 * y ) * 2 + LU( 2 * x + 1, 2 * y ) + LU( xl, 2 * y ) + LU( 2 * x, 2 * y + 1 ) * 2 + LU( 2 * x + 1, 2 * y + 1 ) + LU( xl, 2 * y + 1 ) + 4 ) >> 3 );
    pred[y * cw + x] = (pel) vvo_clip_pel( ( ( a * t ) >> shift ) + b, bd );
  }
  (void) comp;
#undef LU
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Matrix-based intra prediction (luma): IntraPrediction::initIntraMip / predIntraMip (IntraPrediction.cpp:1906,1919),
 * PredictorMIP::deriveBoundaryData (MatrixIntraPrediction.cpp:68), boundaryDownsampling1D (:170), computeReducedPred (:279),
 * predictionUpsampling / predictionUpsampling1D (:196-262), matrices MipData.h:55,347,495. */
static void mip_upsample_1d( pel* dst, const pel* src, const pel* bndry, int srcSizeUpsmpDim, int srcSizeOrthDim, int srcStep, int srcStride,
                             int dstStep, int dstStride, int bndryStep, int upsmpFactor )
{
  const int l2 = vvo_log2( upsmpFactor ), rnd = 1 << ( l2 - 1 );
  const pel* srcLine = src; pel* dstLine = dst; const pel* bndryLine = bndry + bndryStep - 1;
  for( int k = 0; k < srcSizeOrthDim; k++ )
  {
    const pel* before = bndryLine; const pel* behind = srcLine; pel* cur = dstLine;
    for( int j = 0; j < srcSizeUpsmpDim; j++ )
    {
      const pel vBehind = *behind, vBefore = *before, diff = (pel) ( vBehind - vBefore );
      pel scaled = (pel) ( vBefore * ( 1 << l2 ) + rnd );
      for( int i = 0; i < upsmpFactor; i++ ) { scaled = (pel) ( scaled + diff ); *cur = (pel) ( scaled >> l2 ); cur += dstStep; }
      before = behind; behind += srcStep;
    }
    srcLine += srcStride; dstLine += dstStride; bndryLine += bndryStep;
  }
}

static void mip_predict( int w, int h, int modeIdx, int transpose, int bd, const pel* top /* [0] = corner */, const pel* left, pel* pred /* w x h, stride w */ )
{
  const int sizeId = ( w == 4 && h == 4 ) ? 0 : ( w == 4 || h == 4 || ( w == 8 && h == 8 ) ) ? 1 : 2;
  const int bdry = sizeId == 0 ? 2 : 4, red = sizeId < 2 ? 4 : 8;
  const int upH = w / red, upV = h / red;
  pel refT[64], refL[64], rb[8], rbT[8];
  for( int x = 0; x < w; x++ ) refT[x] = top[1 + x];
  for( int y = 0; y < h; y++ ) refL[y] = left[1 + y];
  /* Haar down-sampling of the boundaries */
  for( int side = 0; side < 2; side++ )
  {
    const pel* src = side ? refL : refT; const int len = side ? h : w; pel* dst = rb + side * bdry;
    if( bdry < len )
    {
      const int f = len / bdry, l2 = vvo_log2( f ), rnd = 1 << ( l2 - 1 );
      for( int i = 0, si = 0; i < bdry; i++ ) { int sum = 0; for( int k = 0; k < f; k++ ) sum += src[si++]; dst[i] = (pel) ( ( sum + rnd ) >> l2 ); }
    }
    else memcpy( dst, src, sizeof( pel ) * bdry );
  }
  for( int i = 0; i < bdry; i++ ) { rbT[bdry + i] = rb[i]; rbT[i] = rb[bdry + i]; }
  const int inSize = 2 * bdry;
  const int inOff = rb[0], inOffT = rbT[0];
  const int hasFirstCol = sizeId < 2;
  rb[0]  = (pel) ( hasFirstCol ? ( ( 1 << ( bd - 1 ) ) - inOff  ) : 0 );
  rbT[0] = (pel) ( hasFirstCol ? ( ( 1 << ( bd - 1 ) ) - inOffT ) : 0 );
  for( int i = 1; i < inSize; i++ ) { rb[i] = (pel) ( rb[i] - inOff ); rbT[i] = (pel) ( rbT[i] - inOffT ); }
  /* matrix-vector product on the reduced boundary */
  const uint8_t* matrix = sizeId == 0 ? &vvc_mip_matrix_4x4[modeIdx][0][0] : sizeId == 1 ? &vvc_mip_matrix_8x8[modeIdx][0][0] : &vvc_mip_matrix_16x16[modeIdx][0][0];
  const pel* in = transpose ? rbT : rb;
  const int inputOffset = transpose ? inOffT : inOff;
  int sum = 0;
  for( int i = 0; i < inSize; i++ ) sum += in[i];
  const int offset = ( 1 << 5 ) - 32 * sum;                 /* MIP_SHIFT_MATRIX 6, MIP_OFFSET_MATRIX 32 */
  const int redSize = sizeId == 2;
  pel redPred[64], tmpT[64];
  pel* res = transpose ? tmpT : redPred;
  const uint8_t* wt = matrix;
  for( int p = 0; p < red * red; p++ )
  {
    int acc = redSize ? 0 : in[0] * wt[0];
    for( int i = 1; i < inSize; i++ ) acc += in[i] * wt[i - redSize];
    res[p] = (pel) vvo_clip_pel( ( ( acc + offset ) >> 6 ) + inputOffset, bd );
    wt += inSize - redSize;
  }
  if( transpose ) for( int y = 0; y < red; y++ ) for( int x = 0; x < red; x++ ) redPred[y * red + x] = tmpT[x * red + y];
  /* up-sampling: horizontally into every upV-th row, then vertically */
  if( upH == 1 && upV == 1 ) { memcpy( pred, redPred, sizeof( pel ) * w * h ); return; }
  const pel* verSrc = redPred; int verSrcStep = w;
  if( upH > 1 )
  {
    pel* horDst = pred + ( upV - 1 ) * w;
    verSrc = horDst; verSrcStep *= upV;
    mip_upsample_1d( horDst, redPred, refL, red, red, 1, red, 1, verSrcStep, upV, upH );
  }
  if( upV > 1 ) mip_upsample_1d( pred, verSrc, refT, red, w, verSrcStep, 1, w, 1, 1, upV );
}

/* ciip_w_intra != 0: the block already holds the inter prediction; the intra prediction is blended into it with weight
 * ciip_w_intra / 4 before the residual is added (IntraPrediction::predBlendIntraCiip, IntraPrediction.cpp:887-946) */
/* IntraPrediction::xFillReferenceSamples (:1072): reference line of a w x h block at (x0, y0) of channel ch; top[0] = left[0] = corner.
 * Returns ( left available ) | ( above available ) << 1. */
static int fill_reference( const vvr_pic_header* H, const int32_t* order, const pel* plane, int stride, int ch, int x0, int y0, int w, int h,
                           int topLen, int leftLen, int mrl, int32_t cur, pel* top, pel* left )
{
  const int bd = H->bit_depth, cs = ch;
  const int unit = 4 >> cs;
  const int totalAbove = ( topLen + unit - 1 ) / unit, totalLeft = ( leftLen + unit - 1 ) / unit;
  const int numAbove = w / unit, numLeft = h / unit;
  int nTL = unit_avail( H, order, ch, x0 - 1, y0 - 1, cur );
  int nA = 0, nL = 0;
  if( unit_avail( H, order, ch, x0, y0 - 1, cur ) )
  {
    nA = numAbove;
    for( int k = 0; k < totalAbove - numAbove; k++ ) { if( !unit_avail( H, order, ch, x0 + w + k * unit, y0 - 1, cur ) ) break; nA++; }
  }
  if( unit_avail( H, order, ch, x0 - 1, y0, cur ) )
  {
    nL = numLeft;
    for( int k = 0; k < totalLeft - numLeft; k++ ) { if( !unit_avail( H, order, ch, x0 - 1, y0 + h + k * unit, cur ) ) break; nL++; }
  }
  const int total = totalAbove + totalLeft + 1, nAll = nTL + nA + nL;
  const int dcv = 1 << ( bd - 1 );
#define R( xx, yy ) plane[(size_t) ( yy ) * stride + ( xx )]
  if( nAll == 0 )
  {
    for( int j = 0; j <= topLen + mrl; j++ ) top[j] = (pel) dcv;
    for( int i = 0; i <= leftLen + mrl; i++ ) left[i] = (pel) dcv;
  }
  else if( nAll == total )
  {
    for( int j = 0; j <= topLen + mrl; j++ ) top[j] = R( x0 - ( 1 + mrl ) + j, y0 - ( 1 + mrl ) );
    left[0] = top[0];
    for( int i = 1; i <= leftLen + mrl; i++ ) left[i] = R( x0 - ( 1 + mrl ), y0 - mrl + ( i - 1 ) );
  }
  else if( nL > 0 )
  {
    /* left & below-left (downwards), padded */
    int sz = vvo_min( nL * unit, leftLen );
    for( int i = 0; i < sz; i++ ) left[1 + mrl + i] = R( x0 - ( 1 + mrl ), y0 + i );
    for( int i = sz; i < leftLen; i++ ) left[1 + mrl + i] = left[1 + mrl + sz - 1];
    /* top-left sample(s) */
    if( nTL )
    {
      for( int j = 0; j <= mrl; j++ ) top[j] = R( x0 - ( 1 + mrl ) + j, y0 - ( 1 + mrl ) );
      for( int i = 1; i <= mrl; i++ ) left[i] = R( x0 - ( 1 + mrl ), y0 - ( 1 + mrl ) + i );
    }
    else
    {
      const pel t = R( x0 - ( 1 + mrl ), y0 );
      top[0] = t;
      for( int i = 1; i <= mrl; i++ ) { top[i] = t; left[i] = t; }
    }
    left[0] = top[0];
    /* above & above-right */
    if( nA )
    {
      sz = vvo_min( nA * unit, topLen );
      for( int j = 0; j < sz; j++ ) top[1 + mrl + j] = R( x0 + j, y0 - ( 1 + mrl ) );
      for( int j = sz; j < topLen; j++ ) top[1 + mrl + j] = top[1 + mrl + sz - 1];
    }
    else
      for( int j = 0; j < topLen; j++ ) top[1 + mrl + j] = top[mrl];
  }
  else
  {
    /* left missing, top present */
    const int sz = vvo_min( nA * unit, topLen );
    for( int j = 0; j < sz; j++ ) top[1 + mrl + j] = R( x0 + j, y0 - ( 1 + mrl ) );
    for( int j = sz; j < topLen; j++ ) top[1 + mrl + j] = top[1 + mrl + sz - 1];
    const pel t = R( x0, y0 - ( 1 + mrl ) );
    top[0] = t; left[0] = t;
    for( int i = 1; i <= mrl; i++ ) { top[i] = t; left[i] = t; }
    for( int i = 0; i < leftLen; i++ ) left[1 + mrl + i] = t;
  }
#undef R
  return ( nL > 0 ) | ( ( nA > 0 ) << 1 );
}

int vvo_intra_tu( const vvr_picture* pic, const vvr_cu* cu, const vvr_tu* tu, uint32_t tu_idx, int comp, vvo_planes* reco,
                  const int32_t* order, const int16_t* resi, int has_resi, int ciip_w_intra )
{
  const vvr_pic_header* H = &pic->hdr;
  g_pic = pic; g_curCtu = vvo_ctu_of( H, cu->x, cu->y );
  const int bd = H->bit_depth, cs = comp ? 1 : 0, ch = comp ? 1 : 0;
  const int x0 = ( comp && cu->isp_mode ) ? cu->x >> 1 : tu->x >> cs, y0 = ( comp && cu->isp_mode ) ? cu->y >> 1 : tu->y >> cs;
  const int tbw = ( comp && cu->isp_mode ) ? cu->w >> 1 : tu->w >> cs;      /* ISP: the chroma block of the CU sits in the last TU */
  int w = tbw, h = ( comp && cu->isp_mode ) ? cu->h >> 1 : tu->h >> cs;
  pel* plane = reco->p[comp]; const int stride = reco->stride[comp];
  /* intra sub-partitions (luma): 1 = horizontal split, 2 = vertical split (HOR/VER_INTRA_SUBPARTITIONS) */
  const int isp = !comp && cu->isp_mode, ispVer = cu->isp_mode == 2;
  if( isp && ( cu->multi_ref_idx || cu->bdpcm[0] || ( cu->flags & VVR_CU_MIP ) ) ) { vvo_set_error( "ISP combined with MRL / BDPCM / MIP" ); return -1; }
  if( isp && ispVer && w < 4 )
  {
    /* sub-partitions narrower than 4 are predicted in groups of width 4 (CU::isPredRegDiffFromTB / isFirstTBInPredReg / adjustPredArea,
     * DecCu.cpp:333-371): the first block of a group predicts the group, the others only add their residual */
    if( ( x0 - cu->x ) & 3 )
    {
      if( has_resi ) for( int y = 0; y < h; y++ ) for( int x = 0; x < tbw; x++ )
        plane[(size_t) ( y0 + y ) * stride + x0 + x] = (pel) vvo_clip_pel( plane[(size_t) ( y0 + y ) * stride + x0 + x] + resi[y * tbw + x], bd );
      return 0;
    }
    w = 4;
  }
  if( comp && cu->intra_dir[1] > MDLM_T_IDX ) { vvo_set_error( "bad chroma intra mode" ); return -1; }
  if( !isp && ( w < 4 || h < ( comp ? 2 : 4 ) ) ) { vvo_set_error( "intra blocks narrower than 4 are not restated" ); return -1; }     /* (chroma Nx2 blocks of Nx4 luma CUs exist; 2xN ones do not) */
  const int mrl = comp ? 0 : cu->multi_ref_idx;
  const int bdpcm = comp ? cu->bdpcm[1] : cu->bdpcm[0];
  const int dirMode = cu->intra_dir[ch];
  /* setReferenceArrayLengths (:460); ISP: CU size + sub-partition size along the split, twice the CU size across (:1000-1001) */
  const int topLen = isp ? ( ispVer ? cu->w + w : 2 * cu->w ) : 2 * w, leftLen = isp ? ( ispVer ? 2 * cu->h : cu->h + h ) : 2 * h;
  const int waW = isp ? cu->w : w, waH = isp ? cu->h : h;      /* the wide-angle mapping of ISP blocks uses the CU size (:502,604) */
  pel top[MAXREF + 8], left[MAXREF + 8], ftop[MAXREF + 8], fleft[MAXREF + 8];

  /* ---- reference samples */
  if( !isp ) fill_reference( H, order, plane, stride, ch, x0, y0, w, h, topLen, leftLen, mrl, (int32_t) tu_idx, top, left );
  else
  {
    /* initIntraPatternChTypeISP (:966): the whole CU's reference line is fetched once (first sub-partition); later sub-partitions take
     * the row above / column left of them from the reconstruction of the previous one and the other line from the CU's line */
    pel ctop[MAXREF + 8], cleft[MAXREF + 8];
    const int avail = fill_reference( H, order, plane, stride, 0, cu->x, cu->y, cu->w, cu->h, 2 * cu->w, 2 * cu->h, 0, (int32_t) cu->first_tu, ctop, cleft );
    const int dx = x0 - cu->x, dy = y0 - cu->y;
#define R( xx, yy ) plane[(size_t) ( yy ) * stride + ( xx )]
    if( !dx && !dy ) { for( int j = 0; j <= topLen; j++ ) top[j] = ctop[j]; for( int i = 0; i <= leftLen; i++ ) left[i] = cleft[i]; }
    else if( !ispVer )
    {
      for( int j = 0; j < w; j++ ) top[1 + j] = R( x0 + j, y0 - 1 );
      for( int j = w; j < topLen; j++ ) top[1 + j] = R( x0 + w - 1, y0 - 1 );
      for( int i = 0; i <= leftLen; i++ ) left[i] = cleft[dy + i];
      if( !( avail & 1 ) ) for( int i = 0; i <= leftLen; i++ ) left[i] = R( x0, y0 - 1 );
      top[0] = left[0];
    }
    else
    {
      for( int i = 0; i < h; i++ ) left[1 + i] = R( x0 - 1, y0 + i );
      for( int i = h; i < leftLen; i++ ) left[1 + i] = R( x0 - 1, y0 + h - 1 );
      for( int j = 0; j <= topLen; j++ ) top[j] = ctop[dx + j];
      if( !( avail & 2 ) ) for( int j = 0; j <= topLen; j++ ) top[j] = R( x0 - 1, y0 );
      left[0] = top[0];
    }
#undef R
  }

  /* ---- reference smoothing decision (DecCu.cpp:337 + useFilteredIntraRefSamples :1301) */
  int useFilt = 0;
  if( !comp && !mrl && !bdpcm && dirMode != 1 && !( cu->flags & VVR_CU_MIP ) && !isp )
  {
    if( dirMode == 0 ) useFilt = w * h > 32;
    else
    {
      const int pm = wide_angle( w, h, dirMode );
      const int diff = vvo_min( vvo_abs( pm - 18 ), vvo_abs( pm - 50 ) );
      const int l2 = ( vvo_log2( w ) + vvo_log2( h ) ) >> 1;
      const int am = pm >= 34 ? pm - 50 : -( pm - 18 );
      useFilt = diff > intraFilterThr[0][l2] && ( ( angTable[vvo_abs( am )] & 0x1F ) == 0 );
    }
  }
  const pel *T = top, *L = left;
  if( useFilt )
  {   /* xFilterReferenceSamples (:1251), multiRefIdx is 0 here */
    ftop[0] = fleft[0] = (pel) ( ( left[1] + 2 * top[0] + top[1] + 2 ) >> 2 );
    for( int j = 1; j < topLen; j++ ) ftop[j] = (pel) ( ( top[j + 1] + 2 * top[j] + top[j - 1] + 2 ) >> 2 );
    ftop[topLen] = top[topLen];
    for( int i = 1; i < leftLen; i++ ) fleft[i] = (pel) ( ( left[i + 1] + 2 * left[i] + left[i - 1] + 2 ) >> 2 );
    fleft[leftLen] = left[leftLen];
    T = ftop; L = fleft;
  }

  /* ---- prediction (predIntraAng :474) */
  pel pred[64 * 64];
  int doPDPC = ( w >= 4 && h >= 4 ) && mrl == 0;
  if( !comp && ( cu->flags & VVR_CU_MIP ) )
  {
    mip_predict( w, h, dirMode, ( cu->flags & VVR_CU_MIP_TRANSP ) != 0, bd, top, left, pred );
    doPDPC = 0;
  }
  else if( comp && dirMode >= LM_CHROMA_IDX )
  {
    cclm_predict( pic, cu, tu, tu_idx, comp, reco, order, top, left, pred );
    doPDPC = 0;
  }
  else if( bdpcm )
  {   /* xPredIntraBDPCM (:850) */
    for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ ) pred[y * w + x] = bdpcm == 1 ? L[y + 1] : T[x + 1];
  }
  else if( dirMode == 0 )
  {   /* xPredIntraPlanarCore (:154) */
    const int l2w = vvo_log2( w ), l2h = vvo_log2( h );
    const int bl = L[h + 1], tr = T[w + 1];
    for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
    {
      const int hor = ( L[y + 1] << l2w ) + ( x + 1 ) * ( tr - L[y + 1] );
      const int ver = ( T[x + 1] << l2h ) + ( y + 1 ) * ( bl - T[x + 1] );
      pred[y * w + x] = (pel) ( ( ( hor << l2h ) + ( ver << l2w ) + ( 1 << ( l2w + l2h ) ) ) >> ( 1 + l2w + l2h ) );
    }
  }
  else if( dirMode == 1 )
  {   /* xGetPredValDc (:412) */
    int sum = 0;
    const int denom = w == h ? w << 1 : vvo_max( w, h );
    if( w >= h ) for( int i = 0; i < w; i++ ) sum += T[mrl + 1 + i];
    if( w <= h ) for( int i = 0; i < h; i++ ) sum += L[mrl + 1 + i];
    const pel dc = (pel) ( ( sum + ( denom >> 1 ) ) >> vvo_log2( denom ) );
    for( int i = 0; i < w * h; i++ ) pred[i] = dc;
  }
  else
  {   /* xPredIntraAng (:592) */
    const int predMode = wide_angle( waW, waH, dirMode );
    const int isVer = predMode >= 34;
    const int am = isVer ? predMode - 50 : -( predMode - 18 );
    const int absAm = vvo_abs( am ), sign = am < 0 ? -1 : 1;
    const int invAngle = invAngTable[absAm], absAng = angTable[absAm], angle = sign * absAng;
    pel refAboveBuf[2 * 64 + 3 + 33 * 3 + 64], refLeftBuf[2 * 64 + 3 + 33 * 3 + 64];
    pel *refMain, *refSide;
    int bw = w, bh = h;
    if( angle < 0 )
    {
      pel* ra = refAboveBuf + h; pel* rl = refLeftBuf + w;
      for( int x = 0; x <= w + 1 + mrl; x++ ) ra[x] = T[x];
      for( int y = 0; y <= h + 1 + mrl; y++ ) rl[y] = L[y];
      refMain = isVer ? ra : rl; refSide = isVer ? rl : ra;
      const int sizeSide = isVer ? h : w;
      for( int k = -sizeSide; k <= -1; k++ ) refMain[k] = refSide[vvo_min( ( -k * invAngle + 256 ) >> 9, sizeSide )];
    }
    else
    {
      for( int x = 0; x <= topLen + mrl; x++ ) refAboveBuf[x] = T[x];
      for( int y = 0; y <= leftLen + mrl; y++ ) refLeftBuf[y] = L[y];
      refMain = isVer ? refAboveBuf : refLeftBuf; refSide = isVer ? refLeftBuf : refAboveBuf;
      const int l2r = vvo_log2( w ) - vvo_log2( h );
      const int s = vvo_max( 0, isVer ? l2r : -l2r );
      const int maxIndex = ( mrl << s ) + 2;
      const int refLength = isVer ? topLen : leftLen;
      const pel val = refMain[refLength + mrl];
      for( int z = 1; z <= maxIndex; z++ ) refMain[refLength + mrl + z] = val;
    }
    if( !isVer ) { bw = h; bh = w; }                           /* predict the transposed block */
    pel tmp[64 * 64];
    pel* dst = isVer ? pred : tmp;
    refMain += mrl; refSide += mrl;
    if( angle == 0 )
    {
      if( doPDPC )
      {
        const int scale = ( vvo_log2( bw ) - 2 + vvo_log2( bh ) - 2 + 2 ) >> 2;
        static const int levT[4] = { 3, 6, 12, 24 };
        const int lev = vvo_min( levT[scale], bw );
        const int topLeft = T[0];
        for( int y = 0; y < bh; y++ )
        {
          const int lf = refSide[y + 1];
          for( int x = 0; x < lev; x++ ) { const int wL = 32 >> vvo_min( 31, ( x << 1 ) >> scale ); dst[y * bw + x] = (pel) vvo_clip_pel( ( wL * ( lf - topLeft ) + ( refMain[x + 1] << 6 ) + 32 ) >> 6, bd ); }
          for( int x = lev; x < bw; x++ ) dst[y * bw + x] = refMain[x + 1];
        }
      }
      else
        for( int y = 0; y < bh; y++ ) for( int x = 0; x < bw; x++ ) dst[y * bw + x] = refMain[x + 1];
    }
    else
    {
      if( absAng & 0x1F )
      {
        if( !comp )
        {
          const int diff = vvo_min( vvo_abs( predMode - 18 ), vvo_abs( predMode - 50 ) );
          const int l2 = ( vvo_log2( bw ) + vvo_log2( bh ) ) >> 1;
          const int filterFlag = diff > intraFilterThr[0][l2];
          const int useCubic = isp ? 1 : ( !filterFlag || mrl > 0 );
          int deltaPos = angle * ( 1 + mrl );
          for( int y = 0; y < bh; y++, deltaPos += angle )
          {
            const int di = deltaPos >> 5, df = deltaPos & 31;
            for( int x = 0; x < bw; x++ )
            {
              const int i = di + 1 + x;
              int v;
              if( useCubic ) { const int16_t* f = vvc_chroma_filter[df]; v = ( f[0] * refMain[i - 1] + f[1] * refMain[i] + f[2] * refMain[i + 1] + f[3] * refMain[i + 2] + 32 ) >> 6; v = vvo_clip_pel( (pel) v, bd ); }
              else           { const int8_t* f = gauss[df];              v = ( f[0] * refMain[i - 1] + f[1] * refMain[i] + f[2] * refMain[i + 1] + f[3] * refMain[i + 2] + 32 ) >> 6; }
              dst[y * bw + x] = (pel) v;
            }
          }
        }
        else
        {   /* IntraPredAngleChroma (:334): 2-tap */
          int deltaPos = angle * ( 1 + mrl );
          for( int y = 0; y < bh; y++, deltaPos += angle )
          {
            const int di = deltaPos >> 5, df = deltaPos & 31;
            for( int x = 0; x < bw; x++ ) dst[y * bw + x] = (pel) ( ( ( 32 - df ) * refMain[x + di + 1] + df * refMain[x + di + 2] + 16 ) >> 5 );
          }
        }
      }
      else
      {
        int deltaPos = angle * ( 1 + mrl );
        for( int y = 0; y < bh; y++, deltaPos += angle ) for( int x = 0; x < bw; x++ ) dst[y * bw + x] = refMain[( deltaPos >> 5 ) + 1 + x];
      }
      /* angular PDPC (:810-841) */
      {
        int angularScale = 0;
        if( angle < 0 ) doPDPC = 0;
        else
        {
          const int sideSize = predMode >= 34 ? h : w;
          angularScale = vvo_min( 2, vvo_log2( sideSize ) - ( vvo_floor_log2( 3 * invAngle - 2 ) - 8 ) );
          doPDPC = doPDPC && angularScale >= 0;
        }
        if( doPDPC )
          for( int y = 0; y < bh; y++ )
          {
            int invAngleSum = 256;
            for( int x = 0; x < vvo_min( 3 << angularScale, bw ); x++ )
            {
              invAngleSum += invAngle;
              const int wL = 32 >> ( 2 * x >> angularScale );
              const int lf = refSide[y + ( invAngleSum >> 9 ) + 1];
              dst[y * bw + x] = (pel) ( dst[y * bw + x] + ( ( wL * ( lf - dst[y * bw + x] ) + 32 ) >> 6 ) );
            }
          }
      }
    }
    if( !isVer ) for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ ) pred[y * w + x] = tmp[x * bw + y];
    doPDPC = 0;   /* planar/DC PDPC below must not run for angular modes */
  }
  if( !bdpcm && doPDPC && ( dirMode == 0 || dirMode == 1 ) )
  {   /* IntraPredSampleFilterCore (:212) — applied on the (possibly filtered) reference the prediction used */
    const int scale = ( vvo_log2( w ) - 2 + vvo_log2( h ) - 2 + 2 ) >> 2;
    for( int y = 0; y < h; y++ )
    {
      const int wT = 32 >> vvo_min( 31, ( y << 1 ) >> scale );
      const int lf = L[y + 1];
      for( int x = 0; x < w; x++ )
      {
        const int wL = 32 >> vvo_min( 31, ( x << 1 ) >> scale );
        const int tp = T[x + 1], val = pred[y * w + x];
        pred[y * w + x] = (pel) ( val + ( ( wL * ( lf - val ) + wT * ( tp - val ) + 32 ) >> 6 ) );
      }
    }
  }
  /* ---- reconstruction (DecCu.cpp:396-403) */
  for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
  {
    int pv = pred[y * w + x];
    if( ciip_w_intra ) pv = ( ( 4 - ciip_w_intra ) * plane[(size_t) ( y0 + y ) * stride + x0 + x] + ciip_w_intra * pv + 2 ) >> 2;
    const int v = ( has_resi && x < tbw ) ? vvo_clip_pel( pv + resi[y * tbw + x], bd ) : pv;
    plane[(size_t) ( y0 + y ) * stride + x0 + x] = (pel) v;
  }
  return 0;
}
