/* oracle/vvc_oracle_trafo.c — CPU restatement (TEST INFRASTRUCTURE): dequantisation, LFNST, inverse transforms.
 *
 * Follows  CommonLib/Quant.cpp:295-382 (Quant::dequant), :122-195 (DeQuantImpl),
 *          CommonLib/TrQuant.cpp:79-107 (invLfnstNxNCore), :201-288 (xInvLfnst), :290-318 (invTransformNxN),
 *          :410-486 (xIT), :489-507 (xITransformSkip), CommonLib/TrQuant_EMT.cpp:103-123,389-404 (matrix passes),
 *          :366-375 (cpyResiClip). */
#include "vvc_oracle_common.h"
#include "../tables/vvc_tables.inc"

static const int16_t* tr_matrix( int type, int n )   /* type: 0 DCT2, 1 DCT8, 2 DST7 (TrQuant.cpp:69-74) */
{
  if( type == 0 ) switch( n ) { case 2: return vvc_dct2_2; case 4: return vvc_dct2_4; case 8: return vvc_dct2_8; case 16: return vvc_dct2_16; case 32: return vvc_dct2_32; case 64: return vvc_dct2_64; }
  if( type == 1 ) switch( n ) { case 4: return vvc_dct8_4; case 8: return vvc_dct8_8; case 16: return vvc_dct8_16; case 32: return vvc_dct8_32; }
  if( type == 2 ) switch( n ) { case 4: return vvc_dst7_4; case 8: return vvc_dst7_8; case 16: return vvc_dst7_16; case 32: return vvc_dst7_32; }
  return 0;
}

/* Explicit scaling lists: the matrix entry the reference's expanded tables hold at (x, y) of a (1 << lw) x (1 << lh) block
 * (Quant::setScalingListDec / xSetScalingListDec / xSetRecScalingListDec / processScalingListDec, Quant.cpp:386-570; list ids
 * g_scalingListId, Rom.cpp:504; list type = ( intra ? 0 : 3 ) + component, Quant.h getScalingListType). */
static const vvr_scaling_list* g_scaling = 0;
void vvo_set_scaling_list( const vvr_scaling_list* sl ) { g_scaling = sl; }
static int scaling_entry( const vvr_scaling_list* sl, int listType, int lw, int lh, int x, int y )
{
  static const uint8_t ids[7][6] = { { 0, 0, 0, 0, 0, 0 }, { 0, 0, 0, 0, 0, 1 }, { 2, 3, 4, 5, 6, 7 }, { 8, 9, 10, 11, 12, 13 }, { 14, 15, 16, 17, 18, 19 },
                                     { 20, 21, 22, 23, 24, 25 }, { 26, 21, 22, 27, 24, 25 } };
  const int large = lw > lh ? lw : lh, id = ids[large][listType];
  if( x >= 32 || y >= 32 ) return 0;                                  /* zero-out region: never written */
  if( lw == lh )
  {
    const int sl2 = lw < 3 ? lw : 3, rl2 = lw - sl2;
    if( rl2 > 0 && x == 0 && y == 0 ) return sl->dc[id];
    return sl->coef[id][( ( y >> rl2 ) << sl2 ) + ( x >> rl2 )];
  }
  const int sl2 = large >= 3 ? 3 : 2;
  if( large > 3 && x == 0 && y == 0 ) return sl->dc[id];
  if( lh > lw ) { const int rWH = lh - lw, rH = lh - sl2; return sl->coef[id][( ( y >> rH ) << sl2 ) + ( ( x << rWH ) >> rH )]; }
  { const int rWH = lw - lh, rW = lw - sl2; return sl->coef[id][( ( ( y << rWH ) >> rW ) << sl2 ) + ( x >> rW )]; }
}

/* one 1-D inverse pass: TrQuant_EMT.cpp:103-123 + fastInvCore_ :389-404.
 * dst[i*N + j] = sum_{k < N - skipRows} src[k*lines + i] * M[k*N + j]   for i < lines - skipLines, 0 elsewhere;
 * with clip: dst = clip16( (dst + rnd) >> shift ) on the computed lines. */
static void inv_pass( const int32_t* src, int32_t* dst, const int16_t* M, int N, int lines, int skipLines, int skipRows, int clip, int shift )
{
  const int reduced = lines - skipLines, cutoff = N - skipRows;
  memset( dst, 0, sizeof( int32_t ) * (size_t) lines * N );
  for( int k = 0; k < cutoff; k++ )
    for( int i = 0; i < reduced; i++ )
    {
      const int32_t s = src[k * lines + i];
      for( int j = 0; j < N; j++ ) dst[i * N + j] += s * M[k * N + j];
    }
  if( clip )
  {
    const int rnd = 1 << ( shift - 1 );
    for( int i = 0; i < reduced; i++ ) for( int j = 0; j < N; j++ )
      dst[i * N + j] = vvo_clip3( -32768, 32767, ( dst[i * N + j] + rnd ) >> shift );
  }
}

/* PU::getWideAngIntraMode (UnitTools.cpp:617) */
static int wide_angle_mode( int w, int h, int mode )
{
  static const int modeShift[] = { 0, 6, 10, 12, 14, 15 };
  if( mode < 2 ) return mode;
  const int d = vvo_abs( vvo_log2( w ) - vvo_log2( h ) );
  if( w > h && mode < 2 + modeShift[d] ) mode += 65;            /* VDIA_IDX - 1 */
  else if( h > w && mode > 66 - modeShift[d] ) mode -= 67;      /* VDIA_IDX + 1 */
  return mode;
}

int vvo_residual_block( const vvr_pic_header* hdr, const vvr_cu* cu, const vvr_tu* tu, int comp, const int16_t* coefStream, int16_t* resi, int rstride )
{
  const int bd = hdr->bit_depth;
  const int csh = comp ? 1 : 0;
  int bw = tu->w >> csh, bh = tu->h >> csh;
  if( comp && cu->isp_mode ) { bw = cu->w >> 1; bh = cu->h >> 1; }
  const int lw = vvo_log2( bw ), lh = vvo_log2( bh );
  const int isTS = tu->mts_idx[comp] == VVR_MTS_SKIP;
  const int bdpcm = comp ? cu->bdpcm[1] : cu->bdpcm[0];
  int maxX = tu->max_scan_x[comp], maxY = tu->max_scan_y[comp];
  const int16_t* lev = coefStream + tu->coef_off[comp];
  int32_t* dq  = (int32_t*) calloc( (size_t) bw * bh, sizeof( int32_t ) );   /* m_dqnt, zeroed (TrQuant.cpp:296-297) */
  int32_t* tmp = (int32_t*) calloc( (size_t) bw * bh, sizeof( int32_t ) );
  int32_t* blk = (int32_t*) calloc( (size_t) bw * bh, sizeof( int32_t ) );
  int levStride = maxX + 1;

  /* ---- Quant::dequant (Quant.cpp:295) */
  if( bdpcm )
  {   /* invResDPCM (Quant.cpp:239): accumulate the full block of levels first */
    levStride = bw; maxX = bw - 1; maxY = bh - 1;
    for( int y = 0; y < bh; y++ ) for( int x = 0; x < bw; x++ )
    {
      int v = lev[y * bw + x];
      if( bdpcm == 1 && x > 0 ) v = vvo_clip3( -32768, 32767, dq[y * bw + x - 1] + v );
      if( bdpcm == 2 && y > 0 ) v = vvo_clip3( -32768, 32767, dq[( y - 1 ) * bw + x] + v );
      dq[y * bw + x] = v;
    }
  }
  {
    const int depQuant = ( hdr->tool_flags & VVR_TOOL_DEP_QUANT ) && !isTS;
    int qp = tu->qp[comp];
    if( isTS ) qp = vvo_max( qp, hdr->min_qp_ts );                      /* QpParam Qps[1] (Quant.cpp:103-107) */
    const int per = depQuant ? ( qp + 1 ) / 6 : qp / 6;
    const int rem = depQuant ? ( qp + 1 - 6 * per ) : qp - 6 * per;
    const int needSqrt = !isTS && ( ( lw + lh ) & 1 );                  /* TU::needsSqrt2Scale (UnitTools.cpp:3620) */
    const int trShift  = 15 - bd - ( ( lw + lh ) >> 1 ) - ( needSqrt ? 1 : 0 );
    /* getUseScalingList (Quant.h:103): not for transform skip, optionally not for LFNST blocks */
    const int lfnstApplied = cu->lfnst_idx > 0 && ( cu->tree != VVR_TREE_JOINT || comp == 0 );
    const int useSL = ( hdr->tool_flags & VVR_TOOL_SCALING_LIST ) && g_scaling && !isTS && !( lfnstApplied && ( hdr->tool_flags & VVR_TOOL_SCALING_LIST_NO_LFNST ) );
    const int listType = ( cu->pred_mode == VVR_PRED_INTRA ? 0 : 3 ) + comp;
    const int rightShift = 6 + ( depQuant ? 1 : 0 ) - ( ( isTS ? 0 : trShift ) + per ) + ( useSL ? 4 : 0 );
    const int scaleQP = vvc_inv_quant_scales[needSqrt ? 1 : 0][rem];
    int targetBits = 32 + rightShift - 7; if( targetBits > 16 ) targetBits = 16;
    const int inMax = ( 1 << ( targetBits - 1 ) ) - 1, inMin = -inMax - 1;
    for( int y = 0; y <= maxY; y++ ) for( int x = 0; x <= maxX; x++ )
    {
      const int level = bdpcm ? dq[y * bw + x] : lev[y * levStride + x];
      if( !level ) { if( bdpcm ) dq[y * bw + x] = 0; continue; }
      const int64_t c = vvo_clip3( inMin, inMax, level );
      const int scale = useSL ? scaling_entry( g_scaling, listType, lw, lh, x, y ) * scaleQP : scaleQP;
      int64_t v;
      if( rightShift > 0 ) v = ( c * scale + ( (int64_t) 1 << ( rightShift - 1 ) ) ) >> rightShift;
      else                 v = ( c * scale ) * ( (int64_t) 1 << -rightShift );
      /* the reference computes in 32-bit 'Intermediate_Int'; conformant inputs never exceed it */
      dq[y * bw + x] = vvo_clip3( -32768, 32767, (int) v );
    }
  }

  /* ---- LFNST (TrQuant.cpp:201) */
  if( ( hdr->tool_flags & VVR_TOOL_LFNST ) && cu->lfnst_idx && !isTS && ( cu->tree != VVR_TREE_JOINT || comp == 0 ) )
  {
    const int whge3 = bw >= 8 && bh >= 8;
    int mode;
    if( ( cu->flags & VVR_CU_MIP ) && comp == 0 ) mode = 0;
    else if( comp && cu->intra_dir[1] >= 67 ) mode = cu->lfnst_intra_mode;      /* LM chroma: co-located luma mode */
    else mode = cu->intra_dir[comp ? 1 : 0];
    {
      const int aw = ( cu->isp_mode && !comp ) ? cu->w : bw, ah = ( cu->isp_mode && !comp ) ? cu->h : bh;
      mode = wide_angle_mode( aw, ah, mode );
    }
    int lm = mode < 0 ? mode + 14 + 67 : mode >= 67 ? mode + 14 : mode;          /* getLFNSTIntraMode (TrQuant.cpp:162) */
    const int transpose = ( lm >= 67 && lm >= 67 + 14 ) || ( lm < 67 && lm > 34 );  /* getTransposeFlag (:183) */
    const int sb = whge3 ? 8 : 4;
    const int zeroOut = ( ( bw == 4 && bh == 4 ) || ( bw == 8 && bh == 8 ) ) ? 8 : 16;
    int in[16], out[48];
    for( int i = 0; i < 16; i++ )
    {
      const uint8_t* xy = whge3 ? vvc_lfnst_scan8x8_xy[i] : vvc_lfnst_scan4x4_xy[i];
      in[i] = dq[xy[1] * bw + xy[0]];
    }
    const int set = vvc_lfnst_lut[lm], idx = cu->lfnst_idx - 1, trSize = sb == 8 ? 48 : 16;
    for( int j = 0; j < trSize; j++ )
    {
      int r = 0;
      for( int i = 0; i < zeroOut; i++ ) r += in[i] * ( sb == 8 ? vvc_lfnst8x8[set][idx][j][i] : vvc_lfnst4x4[set][idx][j][i] );
      out[j] = vvo_clip3( -32768, 32767, ( r + 64 ) >> 7 );
    }
    const int* o = out;
    if( transpose )
    {
      if( sb == 4 ) for( int y = 0; y < 4; y++ ) for( int x = 0; x < 4; x++ ) dq[y * bw + x] = out[x * 4 + y];
      else for( int y = 0; y < 8; y++ )
      {
        for( int x = 0; x < 4; x++ ) dq[y * bw + x] = out[x * 8 + y];
        if( y < 4 ) for( int x = 4; x < 8; x++ ) dq[y * bw + x] = out[32 + ( x - 4 ) * 4 + y];
      }
    }
    else
      for( int y = 0; y < sb; y++ ) { const int n = y < 4 ? sb : 4; for( int x = 0; x < n; x++ ) dq[y * bw + x] = *o++; }
    maxX = vvo_max( maxX, vvo_min( bw - 1, 7 ) );
    maxY = vvo_max( maxY, vvo_min( bh - 1, 7 ) );
  }

  if( isTS )
  {   /* xITransformSkip (TrQuant.cpp:489) */
    for( int y = 0; y < bh; y++ ) for( int x = 0; x < bw; x++ ) resi[y * rstride + x] = (int16_t) dq[y * bw + x];
  }
  else
  {   /* xIT (TrQuant.cpp:410) */
    const int trHor = tu->tr_type[comp] & 3, trVer = tu->tr_type[comp] >> 2;
    const int shift1 = 7, shift2 = 20 - bd;
    if( maxX == 0 && maxY == 0 && trHor == 0 && trVer == 0 )
    {
      int dc;
      if( bw > 1 && bh > 1 ) { dc = ( dq[0] * 64 + ( 1 << ( shift1 - 1 ) ) ) >> shift1; dc = ( dc * 64 + ( 1 << ( shift2 - 1 ) ) ) >> shift2; }
      else { const int sh = shift2 + 1; dc = ( dq[0] * 64 + ( 1 << ( sh - 1 ) ) ) >> sh; }
      for( int y = 0; y < bh; y++ ) for( int x = 0; x < bw; x++ ) resi[y * rstride + x] = (int16_t) dc;
    }
    else if( bw > 1 && bh > 1 )
    {
      const int skipW = vvo_max( ( trHor != 0 && bw == 32 ) ? 16 : bw > 32 ? bw - 32 : 0, bw - maxX - 1 );
      const int skipH = vvo_max( ( trVer != 0 && bh == 32 ) ? 16 : bh > 32 ? bh - 32 : 0, bh - maxY - 1 );
      inv_pass( dq,  tmp, tr_matrix( trVer, bh ), bh, bw, skipW, skipH, 1, shift1 );
      inv_pass( tmp, blk, tr_matrix( trHor, bw ), bw, bh, 0,     skipW, 0, shift2 );
      const int rnd = 1 << ( shift2 - 1 );
      for( int y = 0; y < bh; y++ ) for( int x = 0; x < bw; x++ ) resi[y * rstride + x] = (int16_t) vvo_clip3( -32768, 32767, ( blk[y * bw + x] + rnd ) >> shift2 );
    }
    else
    {   /* 1-D blocks (ISP partitions of 4xN / Nx4 CUs): one pass, no intermediate clipping, shift_2nd + 1 (TrQuant.cpp:466-482) */
      const int n = bw == 1 ? bh : bw, tr = bw == 1 ? trVer : trHor, maxPos = bw == 1 ? maxY : maxX;
      const int skip = vvo_max( ( tr != 0 && n == 32 ) ? 16 : n > 32 ? n - 32 : 0, n - maxPos - 1 );
      const int sh = shift2 + 1, rnd = 1 << ( sh - 1 );
      inv_pass( dq, blk, tr_matrix( tr, n ), n, 1, 0, skip, 0, sh );
      for( int y = 0; y < bh; y++ ) for( int x = 0; x < bw; x++ ) resi[y * rstride + x] = (int16_t) vvo_clip3( -32768, 32767, ( blk[y * bw + x] + rnd ) >> sh );
    }
  }
  free( dq ); free( tmp ); free( blk );
  return 0;
}

int vvo_residual( const vvr_pic_header* hdr, const vvr_cu* cu, const vvr_tu* tu, int comp, const int16_t* coef, int16_t* resi )
{
  const int bw = ( comp && cu->isp_mode ) ? cu->w >> 1 : tu->w >> ( comp ? 1 : 0 );
  return vvo_residual_block( hdr, cu, tu, comp, coef, resi, bw );
}
