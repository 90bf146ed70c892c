/* oracle/vvc_oracle_common.h — shared declarations of the CPU restatement (test infrastructure). */
#ifndef VVC_ORACLE_COMMON_H
#define VVC_ORACLE_COMMON_H
#include "vvc_oracle.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int16_t pel;

typedef struct {
  pel* p[3];
  int  w[3], h[3], stride[3];
  int  ncomp;
} vvo_planes;

static inline int vvo_clip3( int lo, int hi, int v ) { return v < lo ? lo : v > hi ? hi : v; }
static inline int vvo_clip_pel( int v, int bd ) { return vvo_clip3( 0, ( 1 << bd ) - 1, v ); }
static inline int vvo_log2( int v ) { int l = 0; while( ( 1 << l ) < v ) l++; return l; }
static inline int vvo_abs( int v ) { return v < 0 ? -v : v; }
static inline int vvo_floor_log2( int v ) { int l = 0; while( ( 2 << l ) <= v ) l++; return l; }   /* getLog2 (CommonDef.h:620) */
static inline int vvo_min( int a, int b ) { return a < b ? a : b; }
static inline int vvo_max( int a, int b ) { return a > b ? a : b; }
static inline int vvo_sgn( int v ) { return ( v > 0 ) - ( v < 0 ); }

/* clamped read of a reference plane = reading the border-extended picture (Picture::extendPicBorder, Picture.cpp:400) */
static inline int vvo_ref_at( const vvo_planes* r, int c, int x, int y )
{
  x = vvo_clip3( 0, r->w[c] - 1, x ); y = vvo_clip3( 0, r->h[c] - 1, y );
  return r->p[c][(size_t) y * r->stride[c] + x];
}

/* slices and tiles (vvr_picture.ctu_slice / ctu_tile): luma position -> CTU; nothing is available to intra prediction, CCLM or the LMCS chroma
 * scaling neighbourhood across a slice or tile boundary (CodingStructure::getCURestricted, CodingStructure.cpp:464) */
static inline int vvo_ctu_of( const vvr_pic_header* H, int lx, int ly ) { const int ctu = 1 << H->log2_ctu; return ( ly >> H->log2_ctu ) * ( ( H->width + ctu - 1 ) >> H->log2_ctu ) + ( lx >> H->log2_ctu ); }
static inline int vvo_same_slice_tile( const vvr_picture* pic, int ctuA, int ctuB )
{
  return ( !pic->ctu_slice || pic->ctu_slice[ctuA] == pic->ctu_slice[ctuB] ) && ( !pic->ctu_tile || pic->ctu_tile[ctuA] == pic->ctu_tile[ctuB] );
}
/* slices with headers of their own (vvr_picture.slices): the slice of a luma position, the tool switches that hold there (the slice's value for
 * the switches a slice header carries, the picture's for the rest), the slice's ALF / weighted-prediction tables */
static inline const vvr_slice_header* vvo_slice_at( const vvr_picture* pic, int lx, int ly )
{
  if( !pic->slices || !pic->ctu_slice ) return 0;
  return &pic->slices[pic->ctu_slice[vvo_ctu_of( &pic->hdr, lx, ly )]];
}
static inline uint32_t vvo_flags_at( const vvr_picture* pic, int lx, int ly )
{
  const vvr_slice_header* s = vvo_slice_at( pic, lx, ly );
  return s ? ( pic->hdr.tool_flags & ~(uint32_t) VVR_SLICE_TOOL_MASK ) | ( s->tool_flags & VVR_SLICE_TOOL_MASK ) : pic->hdr.tool_flags;
}
static inline const vvr_alf_params* vvo_alf_set_at( const vvr_picture* pic, int lx, int ly )
{
  const vvr_slice_header* s = vvo_slice_at( pic, lx, ly );
  return pic->alf_params ? &pic->alf_params[s && pic->num_alf_sets > 1 ? s->alf_set : 0] : 0;
}
static inline const vvr_wp_params* vvo_wp_set_at( const vvr_picture* pic, int lx, int ly )
{
  const vvr_slice_header* s = vvo_slice_at( pic, lx, ly );
  return pic->wp ? &pic->wp[s && pic->num_wp_sets > 1 ? s->wp_set : 0] : 0;
}
/* may SAO / ALF of CTU a read samples of CTU b (pps_loop_filter_across_slices / tiles_enabled_flag) */
static inline int vvo_lf_may_cross( const vvr_picture* pic, int a, int b )
{
  if( a == b ) return 1;
  if( ( pic->hdr.tool_flags & VVR_TOOL_NO_LF_ACROSS_SLICES ) && pic->ctu_slice && pic->ctu_slice[a] != pic->ctu_slice[b] ) return 0;
  if( ( pic->hdr.tool_flags & VVR_TOOL_NO_LF_ACROSS_TILES ) && pic->ctu_tile && pic->ctu_tile[a] != pic->ctu_tile[b] ) return 0;
  if( pic->subpics && pic->num_subpics > 1 )
  {
    /* sub-pictures: the flag of the sub-picture the filtered CTU lies in decides (SampleAdaptiveOffset.cpp:806-818, AdaptiveLoopFilter.cpp:183-186) */
    const int ctu = 1 << pic->hdr.log2_ctu, ctusX = ( pic->hdr.width + ctu - 1 ) / ctu;
    int sa = -1, sb = -1;
    for( uint32_t k = 0; k < pic->num_subpics; k++ )
    {
      const vvr_subpic* sp = &pic->subpics[k];
      const int ax = ( a % ctusX ) * ctu, ay = ( a / ctusX ) * ctu, bx = ( b % ctusX ) * ctu, by = ( b / ctusX ) * ctu;
      if( ax >= sp->x0 && ax <= sp->x1 && ay >= sp->y0 && ay <= sp->y1 ) sa = (int) k;
      if( bx >= sp->x0 && bx <= sp->x1 && by >= sp->y0 && by <= sp->y1 ) sb = (int) k;
    }
    if( sa != sb && sa >= 0 && !pic->subpics[sa].lf_across ) return 0;
  }
  return 1;
}

int  vvo_planes_alloc( vvo_planes* pl, int width, int height, int chroma_format );
void vvo_planes_free( vvo_planes* pl );

/* vvc_oracle_trafo.c */
void vvo_set_scaling_list( const vvr_scaling_list* sl );     /* the picture's explicit scaling lists (NULL: flat) */
int  vvo_residual_block( const vvr_pic_header* hdr, const vvr_cu* cu, const vvr_tu* tu, int comp, const int16_t* coef, int16_t* resi, int rstride );
/* vvc_oracle_inter.c */
int  vvo_inter_cu( const vvr_picture* pic, const vvr_cu* cu, const vvo_planes* refs /* [slot] */, int num_slots, vvo_planes* reco );
void vvo_dmvr_reset( void );
/* vvc_oracle_intra.c */
int  vvo_intra_tu( const vvr_picture* pic, const vvr_cu* cu, const vvr_tu* tu, uint32_t tu_idx, int comp, vvo_planes* reco,
                   const int32_t* tu_order_map /* per 4x4 luma units, per channel type */, const int16_t* resi, int has_resi, int ciip_w_intra );
/* vvc_oracle_loopfilter.c */
void vvo_deblock( const vvr_picture* pic, vvo_planes* reco, int dir );
void vvo_sao( const vvr_picture* pic, const vvo_planes* src, vvo_planes* dst );
void vvo_alf( const vvr_picture* pic, const vvo_planes* src, vvo_planes* dst );

void vvo_set_error( const char* msg );
#endif
