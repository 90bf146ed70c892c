/* oracle/vvc_oracle.c — CPU restatement (TEST INFRASTRUCTURE): picture-level driver.
 *
 * Order of operations = the stage order of DecLibRecon::ctuTask (DecoderLib/DecLibRecon.cpp:732-1110) run as whole-picture
 * passes, and inside a CU the order of DecCu::predAndReco / finishLMCSAndReco / reconstructResi
 * (DecoderLib/DecCu.cpp:271-481, :483-534, :536-583). */
#include "vvc_oracle_common.h"
#include <stdio.h>

static char g_err[256];
void vvo_set_error( const char* msg ) { snprintf( g_err, sizeof( g_err ), "%s", msg ); }
const char* vvo_last_error( void ) { return g_err; }

int vvo_planes_alloc( vvo_planes* pl, int width, int height, int chroma_format )
{
  memset( pl, 0, sizeof( *pl ) );
  pl->ncomp = chroma_format ? 3 : 1;
  for( int c = 0; c < pl->ncomp; c++ )
  {
    pl->w[c] = c ? width >> 1 : width; pl->h[c] = c ? height >> 1 : height; pl->stride[c] = pl->w[c];
    pl->p[c] = (pel*) calloc( (size_t) pl->w[c] * pl->h[c], sizeof( pel ) );
    if( !pl->p[c] ) return -1;
  }
  return 0;
}
void vvo_planes_free( vvo_planes* pl ) { for( int c = 0; c < 3; c++ ) { free( pl->p[c] ); pl->p[c] = 0; } }

/* residuals of one TU into tight per-component buffers (DecCu::reconstructResi, DecCu.cpp:536); returns mask of components that carry a residual */
static int tu_residuals( const vvr_picture* pic, const vvr_cu* cu, const vvr_tu* tu, int16_t* resi[3], int bw[3], int bh[3] )
{
  /* the header as the block's slice sees it: dependent quantisation and the explicit scaling lists are switches of the slice (Quant.cpp:306,336) */
  vvr_pic_header Hs = pic->hdr; Hs.tool_flags = vvo_flags_at( pic, cu->x, cu->y );
  const vvr_pic_header* H = &Hs;
  const int ncomp = H->chroma_format ? 3 : 1;
  int mask = 0;
  for( int c = 0; c < ncomp; c++ )
  {
    bw[c] = ( c && cu->isp_mode ) ? cu->w >> 1 : tu->w >> ( c ? 1 : 0 );
    bh[c] = ( c && cu->isp_mode ) ? cu->h >> 1 : tu->h >> ( c ? 1 : 0 );
  }
  for( int c = 0; c < ncomp; c++ )
  {
    if( !( tu->comp_mask & ( 1 << c ) ) ) continue;
    if( c && tu->joint_cbcr )
    {
      if( c != 1 ) continue;
      /* joint Cb-Cr: one coded block, the other derived (TrQuant::invTransformICT, TrQuant.cpp:108-126,320) */
      const int codedC = ( tu->joint_cbcr >> 1 ) ? 1 : 2;
      if( vvo_residual_block( H, cu, tu, codedC, pic->coef, resi[codedC], bw[codedC] ) ) return -1;
      static const int ict[2][4] = { { 0, 3, 1, 2 }, { 0, -3, -1, -2 } };               /* g_ictModes (Rom.cpp:409) */
      const int mode = ict[( H->tool_flags & VVR_TOOL_JCCR_SIGN ) ? 1 : 0][tu->joint_cbcr];
      int16_t *cb = resi[1], *cr = resi[2];
      for( int i = 0; i < bw[1] * bh[1]; i++ )
      {
        if(      mode ==  1 ) cr[i] = (int16_t) (  cb[i] >> 1 );
        else if( mode == -1 ) cr[i] = (int16_t) ( -cb[i] >> 1 );
        else if( mode ==  2 ) cr[i] = cb[i];
        else if( mode == -2 ) cr[i] = (int16_t) -cb[i];
        else if( mode ==  3 ) cb[i] = (int16_t) (  cr[i] >> 1 );
        else if( mode == -3 ) cb[i] = (int16_t) ( -cr[i] >> 1 );
      }
      bw[2] = bw[1]; bh[2] = bh[1];
      mask |= 6;
    }
    else if( tu->cbf & ( 1 << c ) )
    {
      if( vvo_residual_block( H, cu, tu, c, pic->coef, resi[c], bw[c] ) ) return -1;
      mask |= 1 << c;
    }
  }
  return mask;
}

/* LMCS chroma residual scaling factor of the VPDU that contains luma position (x, y):
 * Reshape::calculateChromaAdjVpduNei (Reshape.cpp:192-274), getPWLIdxInv (:280), calculateChromaAdj (:182).
 * cuAt: index of the CU covering each 4x4 luma cell.  The neighbouring luma samples are reconstructed (mapped domain) samples of
 * CUs that precede the VPDU in decoding order. */
static int lmcs_chroma_scale( const vvr_picture* pic, const vvo_planes* reco, const int32_t* cuAt, int x, int y )
{
  const vvr_pic_header* H = &pic->hdr;
  const int ctu = 1 << H->log2_ctu, w4 = ( H->width + 3 ) >> 2;
  const int n = ctu < 64 ? ctu : 64, nLog = vvo_log2( n );
  int xPos = x & ~( n - 1 ), yPos = y & ~( n - 1 );
  const int32_t tlIdx = cuAt[(size_t) ( yPos >> 2 ) * w4 + ( xPos >> 2 )];
  const vvr_cu* tl = &pic->cu[tlIdx];
  xPos = tl->x; yPos = tl->y;
  /* CodingStructure::getCURestricted (CodingStructure.cpp:464-499), a neighbour in another slice or tile does not count, one inside the same CTU only
   * if it precedes the CU at the VPDU origin in decoding order */
  const int curCtu = vvo_ctu_of( H, xPos, yPos );
  int hasLeft = xPos > 0 && vvo_same_slice_tile( pic, vvo_ctu_of( H, xPos - 1, yPos ), curCtu ), hasAbove = yPos > 0 && vvo_same_slice_tile( pic, vvo_ctu_of( H, xPos, yPos - 1 ), curCtu );
  if( hasLeft && ( ( xPos - 1 ) >> H->log2_ctu ) == ( xPos >> H->log2_ctu ) && cuAt[(size_t) ( yPos >> 2 ) * w4 + ( ( xPos - 1 ) >> 2 )] > tlIdx ) hasLeft = 0;
  if( hasAbove && ( ( yPos - 1 ) >> H->log2_ctu ) == ( yPos >> H->log2_ctu ) && cuAt[(size_t) ( ( yPos - 1 ) >> 2 ) * w4 + ( xPos >> 2 )] > tlIdx ) hasAbove = 0;
  const pel* Y = reco->p[0]; const int st = reco->stride[0];
  int recLuma = 0, pelnum = 0;
  if( hasLeft )  for( int i = 0; i < n; i++ ) { const int k = ( yPos + i ) >= H->height ? H->height - yPos - 1 : i; recLuma += Y[(size_t) ( yPos + k ) * st + xPos - 1]; pelnum++; }
  if( hasAbove ) for( int i = 0; i < n; i++ ) { const int k = ( xPos + i ) >= H->width  ? H->width  - xPos - 1 : i; recLuma += Y[(size_t) ( yPos - 1 ) * st + xPos + k]; pelnum++; }
  int lumaValue;
  if( pelnum == n ) lumaValue = ( recLuma + ( 1 << ( nLog - 1 ) ) ) >> nLog;
  else if( pelnum == 2 * n ) lumaValue = ( recLuma + ( 1 << nLog ) ) >> ( nLog + 1 );
  else lumaValue = 1 << ( H->bit_depth - 1 );
  int idx = pic->lmcs->min_bin;
  for( ; idx <= pic->lmcs->max_bin; idx++ ) if( lumaValue < pic->lmcs->pivot[idx + 1] ) break;
  if( idx > 15 ) idx = 15;
  return pic->lmcs->chroma_scale[idx];
}

/* AreaBuf::scaleSignal (Buffer.cpp:412) */
static void lmcs_scale_residual( int16_t* r, int n, int scale, int bd )
{
  const int maxAbs = ( 1 << bd ) - 1;
  for( int i = 0; i < n; i++ )
  {
    int v = vvo_clip3( -maxAbs - 1, maxAbs, r[i] );
    const int sign = v >= 0 ? 1 : -1, a = sign * v;
    v = sign * ( ( a * scale + ( 1 << 10 ) ) >> 11 );
    r[i] = (int16_t) vvo_clip3( -32768, 32767, v );
  }
}

int vvo_reconstruct( const vvr_picture* pic, const uint16_t* const* ref_planes, uint16_t* const* out_planes, int flags )
{
  vvo_dmvr_reset();
  vvo_set_scaling_list( pic->scaling );      /* (whether a block uses it is the switch of its slice: vvo_flags_at) */
  const vvr_pic_header* H = &pic->hdr;
  const int W = H->width, Hh = H->height, ncomp = H->chroma_format ? 3 : 1;
  int rc = -1;
  g_err[0] = 0;
  /* reference pictures */
  int numSlots = 0;
  for( int l = 0; l < 2; l++ ) for( int i = 0; i < H->num_ref[l]; i++ ) if( H->ref_slot[l][i] + 1 > numSlots ) numSlots = H->ref_slot[l][i] + 1;
  vvo_planes* refs = (vvo_planes*) calloc( (size_t) ( numSlots + 1 ), sizeof( vvo_planes ) );
  vvo_planes reco, flt;
  memset( &reco, 0, sizeof( reco ) ); memset( &flt, 0, sizeof( flt ) );
  int16_t* resi[3] = { 0, 0, 0 };
  int32_t* order = 0;
  int32_t* cuAt = 0;
  if( H->slice_type != 2 )
    for( int l = 0; l < 2; l++ ) for( int i = 0; i < H->num_ref[l]; i++ )
    {
      const int s = H->ref_slot[l][i];
      if( refs[s].p[0] ) continue;
      /* a scaled reference picture comes with its own size (vvr_rpr_ref): its planes are tight at that size */
      const int rw = pic->rpr ? pic->rpr->ref[l][i].width : W, rh = pic->rpr ? pic->rpr->ref[l][i].height : Hh;
      if( vvo_planes_alloc( &refs[s], rw, rh, H->chroma_format ) ) goto done;
      for( int c = 0; c < ncomp; c++ )
      {
        if( !ref_planes || !ref_planes[s * 3 + c] ) { vvo_set_error( "missing reference plane" ); goto done; }
        for( size_t k = 0; k < (size_t) refs[s].w[c] * refs[s].h[c]; k++ ) refs[s].p[c][k] = (pel) ref_planes[s * 3 + c][k];
      }
    }
  if( vvo_planes_alloc( &reco, W, Hh, H->chroma_format ) || vvo_planes_alloc( &flt, W, Hh, H->chroma_format ) ) goto done;
  for( int c = 0; c < 3; c++ ) resi[c] = (int16_t*) malloc( sizeof( int16_t ) * 128 * 128 );

  /* decode-order index of the transform block covering each 4x4 (luma tree / chroma tree): intra reference availability */
  const int w4 = ( W + 3 ) >> 2, h4 = ( Hh + 3 ) >> 2;
  order = (int32_t*) malloc( sizeof( int32_t ) * (size_t) w4 * h4 * 2 );
  for( size_t k = 0; k < (size_t) w4 * h4 * 2; k++ ) order[k] = 0x7fffffff;
  for( uint32_t i = 0; i < pic->num_cu; i++ )
  {
    const vvr_cu* cu = &pic->cu[i];
    for( uint32_t t = cu->first_tu; t < cu->first_tu + cu->num_tu; t++ )
    {
      const vvr_tu* tu = &pic->tu[t];
      for( int ch = 0; ch < 2; ch++ )
      {
        if( ch == 0 && !( tu->comp_mask & 1 ) ) continue;
        if( ch == 1 && !( tu->comp_mask & 6 ) ) continue;
        int x0 = tu->x, y0 = tu->y, ww = tu->w, hh = tu->h;
        if( ch == 1 && cu->isp_mode ) { x0 = cu->x; y0 = cu->y; ww = cu->w; hh = cu->h; }
        for( int y = y0; y < y0 + hh && y < Hh; y += 4 ) for( int x = x0; x < x0 + ww && x < W; x += 4 )
          order[(size_t) ch * w4 * h4 + ( y >> 2 ) * w4 + ( x >> 2 )] = (int32_t) t;
      }
    }
  }

  /* CU covering every 4x4 luma cell (LMCS chroma scaling looks up the CU at the VPDU origin) */
  cuAt = (int32_t*) malloc( sizeof( int32_t ) * (size_t) w4 * h4 );
  for( uint32_t i = 0; i < pic->num_cu; i++ )
  {
    const vvr_cu* cu = &pic->cu[i];
    if( cu->tree == VVR_TREE_CHROMA ) continue;                      /* dual tree: the luma CUs */
    for( int y = cu->y; y < cu->y + cu->h && y < Hh; y += 4 ) for( int x = cu->x; x < cu->x + cu->w && x < W; x += 4 ) cuAt[(size_t) ( y >> 2 ) * w4 + ( x >> 2 )] = (int32_t) i;
  }
  /* LMCS chroma residual scaling: the picture's flag and the slice's LMCS switch (DecCu.cpp:383,489,618) */
#define CSCALE_AT( x_, y_ ) ( ( vvo_flags_at( pic, x_, y_ ) & VVR_TOOL_LMCS ) && ( vvo_flags_at( pic, x_, y_ ) & VVR_TOOL_LMCS_CSCALE ) && pic->lmcs && ncomp == 3 )
#define CSCALE_TU( tu_, mask_ ) \
  if( CSCALE_AT( tu_->x, tu_->y ) && ( ( mask_ ) & 6 ) && bw[1] * bh[1] > 4 ) \
  { \
    const int sc = lmcs_chroma_scale( pic, &reco, cuAt, tu_->x, tu_->y ); \
    for( int c = 1; c < 3; c++ ) if( ( mask_ ) & ( 1 << c ) ) lmcs_scale_residual( resi[c], bw[c] * bh[c], sc, H->bit_depth ); \
  }

  /* ---- INTER + INTRA stages */
  for( uint32_t i = 0; i < pic->num_cu; i++ )
  {
    const vvr_cu* cu = &pic->cu[i];
    if( cu->pred_mode == VVR_PRED_INTER )
    {
      if( vvo_inter_cu( pic, cu, refs, numSlots, &reco ) ) goto done;
      if( ( vvo_flags_at( pic, cu->x, cu->y ) & VVR_TOOL_LMCS ) && pic->lmcs )
      {   /* forward luma mapping of the inter prediction (DecCu.cpp:458-476, Reshape::rspBufFwd :413, rspFwdCore Buffer.cpp:321) */
        for( int y = 0; y < cu->h; y++ ) for( int x = 0; x < cu->w; x++ )
        {
          pel* d = &reco.p[0][(size_t) ( cu->y + y ) * reco.stride[0] + cu->x + x];
          *d = pic->lmcs->fwd_lut[*d];
        }
      }
      if( cu->flags & VVR_CU_CIIP )
      {
        /* DecCu::predAndReco( cu, doCiipIntra ) (DecCu.cpp:453-470): planar intra prediction of the whole CU from the reconstructed
         * neighbours, blended into the inter prediction, then the residual.  The reference runs this in its intra stage; a single
         * pass in decoding order sees the same neighbours. */
        vvr_cu icu = *cu;
        icu.intra_dir[0] = icu.intra_dir[1] = 0; icu.multi_ref_idx = 0; icu.bdpcm[0] = icu.bdpcm[1] = 0; icu.isp_mode = 0; icu.flags &= (uint16_t) ~VVR_CU_MIP;
        const int wIntra = 1 + ( cu->ciip_neigh_intra & 1 ) + ( ( cu->ciip_neigh_intra >> 1 ) & 1 );
        if( cu->num_tu != 1 )
        {
          /* a CU of several transform units (the split at a largest transform size of 32): predicted and blended as a whole, the residuals are added unit by
           * unit (finishLMCSAndReco, DecCu.cpp:483-520) - here: collected in a residual of the CU's size (zero where a unit has none: pred + 0, already in range) */
          const int nc = H->chroma_format ? 3 : 1;
          int16_t* cuResi[3] = { NULL, NULL, NULL };
          int any[3] = { 0, 0, 0 }, fail = 0;
          for( int c = 0; c < nc; c++ ) cuResi[c] = (int16_t*) calloc( (size_t) ( cu->w >> ( c ? 1 : 0 ) ) * ( cu->h >> ( c ? 1 : 0 ) ), sizeof( int16_t ) );
          for( uint32_t t = cu->first_tu; t < cu->first_tu + cu->num_tu && !fail; t++ )
          {
            const vvr_tu* tu = &pic->tu[t];
            int bw[3], bh[3];
            const int mask = ( cu->flags & VVR_CU_ROOT_CBF ) ? tu_residuals( pic, cu, tu, resi, bw, bh ) : 0;
            if( mask < 0 ) { fail = 1; break; }
            CSCALE_TU( tu, mask )
            for( int c = 0; c < nc; c++ )
            {
              if( !( mask & ( 1 << c ) ) ) continue;
              const int cs = c ? 1 : 0, ox = ( tu->x - cu->x ) >> cs, oy = ( tu->y - cu->y ) >> cs, cw = cu->w >> cs;
              for( int y = 0; y < bh[c]; y++ ) for( int x = 0; x < bw[c]; x++ ) cuResi[c][(size_t) ( oy + y ) * cw + ox + x] = resi[c][y * bw[c] + x];
              any[c] = 1;
            }
          }
          vvr_tu whole = pic->tu[cu->first_tu];
          whole.x = cu->x; whole.y = cu->y; whole.w = cu->w; whole.h = cu->h;
          for( int c = 0; c < nc && !fail; c++ ) if( vvo_intra_tu( pic, &icu, &whole, cu->first_tu, c, &reco, order, cuResi[c], any[c], wIntra ) ) fail = 1;
          for( int c = 0; c < nc; c++ ) free( cuResi[c] );
          if( fail ) goto done;
          continue;
        }
        const vvr_tu* tu = &pic->tu[cu->first_tu];
        int bw[3], bh[3];
        const int mask = ( cu->flags & VVR_CU_ROOT_CBF ) ? tu_residuals( pic, cu, tu, resi, bw, bh ) : 0;
        if( mask < 0 ) goto done;
        for( int c = 0; c < ncomp; c++ )
        {
          if( c == 1 ) { CSCALE_TU( tu, mask ) }            /* after the luma of this TU (finishLMCSAndReco order, DecCu.cpp:498-512) */
          if( c && cu->w == 4 )
          {   /* 2-wide chroma blocks are not blended (predBlendIntraCiip, IntraPrediction.cpp:891): inter prediction + residual */
            if( mask & ( 1 << c ) )
              for( int y = 0; y < bh[c]; y++ ) for( int x = 0; x < bw[c]; x++ )
              {
                pel* d = &reco.p[c][(size_t) ( ( tu->y >> 1 ) + y ) * reco.stride[c] + ( tu->x >> 1 ) + x];
                *d = (pel) vvo_clip_pel( *d + resi[c][y * bw[c] + x], H->bit_depth );
              }
            continue;
          }
          if( vvo_intra_tu( pic, &icu, tu, cu->first_tu, c, &reco, order, resi[c], ( mask >> c ) & 1, wIntra ) ) goto done;
        }
      }
      else if( cu->flags & VVR_CU_ROOT_CBF )
        for( uint32_t t = cu->first_tu; t < cu->first_tu + cu->num_tu; t++ )
        {
          const vvr_tu* tu = &pic->tu[t];
          int bw[3], bh[3];
          const int mask = tu_residuals( pic, cu, tu, resi, bw, bh );
          if( mask < 0 ) goto done;
          for( int c = 0; c < ncomp; c++ )
          {
            if( c == 1 ) { CSCALE_TU( tu, mask ) }
            if( !( mask & ( 1 << c ) ) ) continue;
            /* AreaBuf::reconstruct (Buffer.cpp:482): reco = clip( pred + resi ) */
            const int bx = tu->x >> ( c ? 1 : 0 ), by = tu->y >> ( c ? 1 : 0 );
            for( int y = 0; y < bh[c]; y++ ) for( int x = 0; x < bw[c]; x++ )
            {
              pel* d = &reco.p[c][(size_t) ( by + y ) * reco.stride[c] + bx + x];
              *d = (pel) vvo_clip_pel( *d + resi[c][y * bw[c] + x], H->bit_depth );
            }
          }
        }
    }
    else if( cu->pred_mode == VVR_PRED_INTRA )
    {
      for( uint32_t t = cu->first_tu; t < cu->first_tu + cu->num_tu; t++ )
      {
        const vvr_tu* tu = &pic->tu[t];
        int bw[3], bh[3];
        const int mask = tu_residuals( pic, cu, tu, resi, bw, bh );
        if( mask < 0 ) goto done;
        for( int c = 0; c < ncomp; c++ )
        {
          if( c == 1 ) { CSCALE_TU( tu, mask ) }
          if( !( tu->comp_mask & ( 1 << c ) ) ) continue;
          if( vvo_intra_tu( pic, cu, tu, t, c, &reco, order, resi[c], ( mask >> c ) & 1, 0 ) ) goto done;
        }
      }
    }
    else if( cu->pred_mode == VVR_PRED_IBC )
    {
      /* intra block copy (InterPrediction::xIntraBlockCopy, InterPrediction.cpp:1995; DecCu::predAndReco, DecCu.cpp:442-470): the
       * prediction is a copy of reconstructed, not yet loop-filtered samples of the current picture at the block vector; chroma uses
       * the halved vector.  The reference reads them from the IBC virtual buffer of the CTU row (CodingStructure::fillIBCbuffer,
       * CodingStructure.cpp:550), which holds exactly these picture samples for a valid vector.  No forward luma mapping (the copied
       * samples are in the mapped domain already, DecCu.cpp:460,472); the residual is added as for an inter CU (finishLMCSAndReco). */
      if( !( H->tool_flags & VVR_TOOL_IBC ) ) { vvo_set_error( "IBC CU in a picture without VVR_TOOL_IBC" ); goto done; }
      if( cu->tree == VVR_TREE_CHROMA || cu->w > 64 || cu->h > 64 || ( ( cu->mv[0][0][0] | cu->mv[0][0][1] ) & 15 ) ) { vvo_set_error( "IBC CU: chroma tree, larger than 64, or fractional block vector" ); goto done; }
      const int bvx = cu->mv[0][0][0] >> 4, bvy = cu->mv[0][0][1] >> 4;
      const int nc = ( cu->tree == VVR_TREE_JOINT ) ? ncomp : 1;
      const int ctuMask = ( 1 << H->log2_ctu ) - 1;
      for( int c = 0; c < nc; c++ )
      {
        const int sh = c ? 1 : 0;
        const int bx = cu->x >> sh, by = cu->y >> sh, bw_ = cu->w >> sh, bh_ = cu->h >> sh;
        const int rx = bx + ( c ? bvx >> 1 : bvx ), ry = by + ( c ? bvy >> 1 : bvy );
        const int rowTop = ( cu->y & ~ctuMask ) >> sh, rowEnd = ( ( cu->y & ~ctuMask ) + ctuMask + 1 ) >> sh;
        if( rx < 0 || ry < rowTop || rx + bw_ > reco.w[c] || ry + bh_ > rowEnd || ry + bh_ > reco.h[c] ) { vvo_set_error( "IBC CU: reference block outside the picture or the CTU row" ); goto done; }
        for( int y = 0; y < bh_; y++ )
          memmove( &reco.p[c][(size_t) ( by + y ) * reco.stride[c] + bx], &reco.p[c][(size_t) ( ry + y ) * reco.stride[c] + rx], sizeof( pel ) * (size_t) bw_ );
      }
      if( cu->flags & VVR_CU_ROOT_CBF )
        for( uint32_t t = cu->first_tu; t < cu->first_tu + cu->num_tu; t++ )
        {
          const vvr_tu* tu = &pic->tu[t];
          int bw[3], bh[3];
          const int mask = tu_residuals( pic, cu, tu, resi, bw, bh );
          if( mask < 0 ) goto done;
          for( int c = 0; c < ncomp; c++ )
          {
            if( c == 1 ) { CSCALE_TU( tu, mask ) }
            if( !( mask & ( 1 << c ) ) ) continue;
            const int bx = tu->x >> ( c ? 1 : 0 ), by = tu->y >> ( c ? 1 : 0 );
            for( int y = 0; y < bh[c]; y++ ) for( int x = 0; x < bw[c]; x++ )
            {
              pel* d = &reco.p[c][(size_t) ( by + y ) * reco.stride[c] + bx + x];
              *d = (pel) vvo_clip_pel( *d + resi[c][y * bw[c] + x], H->bit_depth );
            }
          }
        }
    }
    else { vvo_set_error( "unknown prediction mode" ); goto done; }
  }

  if( pic->lmcs )
  {   /* inverse luma mapping, CTU by CTU: where the CTU's slice uses LMCS (Reshape::rspCtuBcw :376-392, applyLutCore Buffer.cpp:200) */
    for( int y = 0; y < reco.h[0]; y++ ) for( int x = 0; x < reco.w[0]; x++ )
      if( vvo_flags_at( pic, x, y ) & VVR_TOOL_LMCS ) { pel* d = &reco.p[0][(size_t) y * reco.stride[0] + x]; *d = pic->lmcs->inv_lut[*d & 4095]; }
  }
  if( !( flags & VVO_STOP_AFTER_RECO ) )
  {
    vvo_deblock( pic, &reco, 0 );
    vvo_deblock( pic, &reco, 1 );
    if( !( flags & VVO_STOP_AFTER_DBK ) )
    {
      if( H->tool_flags & ( VVR_TOOL_SAO_LUMA | VVR_TOOL_SAO_CHROMA ) ) { vvo_sao( pic, &reco, &flt ); vvo_planes tmp = reco; reco = flt; flt = tmp; }
      if( !( flags & VVO_STOP_AFTER_SAO ) && ( H->tool_flags & VVR_TOOL_ALF ) && pic->alf && pic->alf_params ) { vvo_alf( pic, &reco, &flt ); vvo_planes tmp = reco; reco = flt; flt = tmp; }
    }
  }
  for( int c = 0; c < ncomp; c++ ) if( out_planes[c] )
    for( size_t k = 0; k < (size_t) reco.w[c] * reco.h[c]; k++ ) out_planes[c][k] = (uint16_t) reco.p[c][k];
  rc = 0;
done:
  for( int s = 0; s < numSlots; s++ ) vvo_planes_free( &refs[s] );
  free( refs ); vvo_planes_free( &reco ); vvo_planes_free( &flt );
  for( int c = 0; c < 3; c++ ) free( resi[c] );
  free( order ); free( cuAt );
  return rc;
}
