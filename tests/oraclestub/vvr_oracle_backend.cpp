// TEST INFRASTRUCTURE (tests/ only; never shipped, never loaded by the product): the ten entry points of include/vvr.h that the drop-in decoder library calls,
// served by the CPU ORACLE (oracle/libvvoracle.so, the plain-C restatement of the reconstruction path that the GPU parity tests pin the HIP kernels to).  With this
// library bound instead of libvvdec_amd.so the reference's application decodes a bitstream END TO END ON THE CPU through integration/DecLibReconDropIn.cpp and
// integration/vvr_extract.h: the output MD5 then checks the flattening of what the real parser left in CodingStructure - and, where the description leaves the
// deblocking edge parameters to the back-end (VVR_TOOL_LFP_ON_DEVICE), vvdec_amd/csrc/vvr_lf_init.h, the source of k_lf_init - against the reference decoder,
// on as many streams as one likes and without a GPU (tests/test_dropin_library.py, tools/fuzz_dropin_on_the_oracle.py).  No sample is computed by product code here.
#include <cstdint>
#include <cstring>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/vvr.h"
#include "../../oracle/vvc_oracle.h"
#include "../../vvdec_amd/csrc/vvr_lf_init.h"

struct OSlot { int w = 0, h = 0; std::vector<uint16_t> p[3]; };
struct vvr_context
{
  vvr_config cfg;
  std::vector<OSlot> slots;
  std::map<int, std::vector<int32_t>> dmvr;     // job -> delta MVs
  std::map<int, int> status;
  int nextJob = 0;
  std::string err;
  std::mutex mu;
};
static std::mutex g_oracle;                     // (the oracle keeps the delta MVs of its last call in a global)

static void sizeSlot( vvr_context* c, int slot, int w, int h )
{
  OSlot& s = c->slots[slot];
  if( s.w == w && s.h == h && !s.p[0].empty() ) return;
  s.w = w; s.h = h;
  s.p[0].assign( (size_t) w * h, 0 );
  for( int k = 1; k < 3; k++ ) s.p[k].assign( c->cfg.chroma_format ? (size_t) ( w >> 1 ) * ( h >> 1 ) : 0, 0 );
}

extern "C" {
#define API __attribute__(( visibility( "default" ) ))
API int vvr_create( const vvr_config* cfg, vvr_context** out )
{
  if( !cfg || !out || cfg->abi_version != VVR_ABI_VERSION ) return VVR_ERR_PARAMETER;
  vvr_context* c = new vvr_context(); c->cfg = *cfg; c->slots.resize( cfg->num_slots );
  *out = c; return VVR_OK;
}
API void vvr_destroy( vvr_context* c ) { delete c; }
API const char* vvr_last_error( const vvr_context* c ) { return c ? c->err.c_str() : "no context"; }
API int vvr_slot_picture_size( vvr_context* c, int slot, int w, int h )
{
  if( !c || slot < 0 || slot >= (int) c->slots.size() || w > c->cfg.max_width || h > c->cfg.max_height ) return VVR_ERR_PARAMETER;
  std::lock_guard<std::mutex> lk( c->mu ); sizeSlot( c, slot, w, h ); return VVR_OK;
}
API int vvr_write_plane( vvr_context* c, int slot, int comp, const uint16_t* src, size_t stride )
{
  if( !c || slot < 0 || slot >= (int) c->slots.size() || comp < 0 || comp > 2 ) return VVR_ERR_PARAMETER;
  std::lock_guard<std::mutex> lk( c->mu );
  OSlot& s = c->slots[slot];
  if( !s.w ) sizeSlot( c, slot, c->cfg.max_width, c->cfg.max_height );
  const int w = comp ? s.w >> 1 : s.w, h = comp ? s.h >> 1 : s.h;
  if( s.p[comp].empty() ) return VVR_ERR_PARAMETER;
  for( int y = 0; y < h; y++ ) memcpy( &s.p[comp][(size_t) y * w], src + (size_t) y * stride, sizeof( uint16_t ) * w );
  return VVR_OK;
}
API int vvr_submit( vvr_context* c, const vvr_picture* pic )
{
  if( !c || !pic ) return VVR_ERR_PARAMETER;
  std::lock_guard<std::mutex> lk( c->mu );
  const vvr_pic_header& h = pic->hdr;
  if( h.out_slot < 0 || h.out_slot >= (int) c->slots.size() ) { c->err = "bad output slot"; return VVR_ERR_PARAMETER; }
  vvr_picture P = *pic;
  std::vector<vvr_cu> cuCopy;
  if( const char* e = getenv( "VVR_ORACLE_NO_DMVR" ) )      // developer aid: 1 = every DMVR CU predicted without the refinement; 2 = only those whose MVs leave the picture by more than a CTU
  {
    cuCopy.assign( pic->cu, pic->cu + pic->num_cu );
    for( auto& u : cuCopy )
    {
      if( u.mc_mode != VVR_MC_DMVR && u.mc_mode != VVR_MC_DMVR_BDOF ) continue;
      bool far = false;
      for( int l = 0; l < 2; l++ ) { const int x = u.x + ( u.mv[l][0][0] >> 4 ); far |= x < -64 - 8 || x + u.w > h.width + 64 + 8; }
      if( atoi( e ) == 1 || far ) u.mc_mode = u.mc_mode == VVR_MC_DMVR_BDOF ? VVR_MC_BDOF : VVR_MC_BI;
    }
    P.cu = cuCopy.data();
  }
  if( const char* e = getenv( "VVR_ORACLE_DUMP_AT" ) )      // developer aid: "poc,x,y" -> the CU that covers the luma sample
  {
    int poc = 0, x = 0, y = 0;
    if( sscanf( e, "%d,%d,%d", &poc, &x, &y ) == 3 && poc == h.poc )
      for( uint32_t k = 0; k < pic->num_cu; k++ )
      {
        const vvr_cu& u = pic->cu[k];
        if( u.tree == VVR_TREE_CHROMA || x < u.x || y < u.y || x >= u.x + u.w || y >= u.y + u.h ) continue;
        fprintf( stderr, "[oracle back-end] POC %d CU %u at (%d,%d) %dx%d pred %d flags %04x mc_mode %d inter_dir %d ref %d %d mv0 (%d,%d) mv1 (%d,%d) bcw %d imv %d qp %d tool_flags %08x wrap %d\n", h.poc, k, u.x, u.y, u.w, u.h,
                 u.pred_mode, u.flags, u.mc_mode, u.inter_dir, u.ref_idx[0], u.ref_idx[1], u.mv[0][0][0], u.mv[0][0][1], u.mv[1][0][0], u.mv[1][0][1], u.bcw_idx, u.imv, u.qp, h.tool_flags, h.wrap_offset );
      }
  }
  // the edge parameters left to the back-end: derived with the source the device kernel is compiled from
  std::vector<vvr_lfp> lf[2];
  if( ( h.tool_flags & VVR_TOOL_LFP_ON_DEVICE ) && !( h.tool_flags & VVR_TOOL_DEBLOCK_OFF ) )
  {
    const int w4 = ( h.width + 3 ) >> 2, h4 = ( h.height + 3 ) >> 2, ctu = 1 << h.log2_ctu, ctusX = ( h.width + ctu - 1 ) / ctu, ctusY = ( h.height + ctu - 1 ) / ctu;
    std::vector<LfCell> cell( (size_t) w4 * h4 ), cellC( (size_t) w4 * h4 );
    std::vector<LfMv> mv( (size_t) w4 * h4 ); std::vector<uint32_t> ref( (size_t) w4 * h4 );
    lf_init_maps_host( h, pic->cu, pic->num_cu, pic->tu, pic->num_tu, cell.data(), cellC.data(), mv.data(), ref.data(), w4, h4 );
    for( uint32_t k = 0; k < pic->num_cu; k++ )
    {
      const vvr_cu& u = pic->cu[k]; vvr_motion m;
      if( lfi_cell_motion( h, u, u.x >> 2, u.y >> 2, m ) != 2 ) continue;
      if( !pic->motion ) { c->err = "missing motion field"; return VVR_ERR_PARAMETER; }
      for( int y = u.y >> 2; y < lfi_min( ( u.y + u.h + 3 ) >> 2, h4 ); y++ ) for( int x = u.x >> 2; x < lfi_min( ( u.x + u.w + 3 ) >> 2, w4 ); x++ )
      { mv[(size_t) y * w4 + x] = lfi_pack_mv( pic->motion[(size_t) y * w4 + x] ); ref[(size_t) y * w4 + x] = lfi_pack_refs( pic->motion[(size_t) y * w4 + x] ); }
    }
    std::vector<uint16_t> ctuSubpic;
    if( pic->subpics && pic->num_subpics > 1 )
    {
      ctuSubpic.assign( (size_t) ctusX * ctusY, 0 );
      for( uint32_t k = 0; k < pic->num_subpics; k++ ) for( int y = pic->subpics[k].y0 >> h.log2_ctu; y <= pic->subpics[k].y1 >> h.log2_ctu; y++ ) for( int x = pic->subpics[k].x0 >> h.log2_ctu; x <= pic->subpics[k].x1 >> h.log2_ctu; x++ ) ctuSubpic[(size_t) y * ctusX + x] = (uint16_t) k;
    }
    LfInitView V; V.hdr = &pic->hdr; V.cell = cell.data(); V.cellC = cellC.data(); V.mv = mv.data(); V.ref = ref.data(); V.ctuSlice = pic->ctu_slice; V.ctuTile = pic->ctu_tile;
    V.ctuSubpic = ctuSubpic.empty() ? nullptr : ctuSubpic.data(); V.subpics = pic->subpics; V.slices = pic->slices; V.w4 = w4; V.h4 = h4; V.ctusX = ctusX;
    lf[0].resize( (size_t) w4 * h4 ); lf[1].resize( (size_t) w4 * h4 );
    lf_init_tables_host( V, lf[0].data(), lf[1].data() );
    P.lfp[0] = lf[0].data(); P.lfp[1] = lf[1].data();
  }
  std::vector<vvr_lfp> none;
  if( !P.lfp[0] ) { none.assign( (size_t) ( ( h.width + 3 ) >> 2 ) * ( ( h.height + 3 ) >> 2 ), vvr_lfp() ); P.lfp[0] = P.lfp[1] = none.data(); }      // (deblocking off)
  // reference planes as the oracle takes them: tight, [slot * 3 + component]
  int numSlots = 0;
  for( int l = 0; l < 2; l++ ) for( int i = 0; i < h.num_ref[l]; i++ ) numSlots = lfi_max( numSlots, h.ref_slot[l][i] + 1 );
  std::vector<const uint16_t*> refs( (size_t) lfi_max( 1, numSlots ) * 3, nullptr );
  if( h.slice_type != 2 )
    for( int l = 0; l < 2; l++ ) for( int i = 0; i < h.num_ref[l]; i++ )
    {
      const int s = h.ref_slot[l][i];
      if( s < 0 || s >= (int) c->slots.size() || c->slots[s].p[0].empty() ) { c->err = "reference slot holds no picture"; return VVR_ERR_PARAMETER; }
      const int rw = pic->rpr ? pic->rpr->ref[l][i].width : h.width, rh = pic->rpr ? pic->rpr->ref[l][i].height : h.height;
      if( c->slots[s].w != rw || c->slots[s].h != rh ) { c->err = "reference slot holds a picture of another size than the description says"; return VVR_ERR_PARAMETER; }
      for( int k = 0; k < 3; k++ ) refs[(size_t) s * 3 + k] = c->slots[s].p[k].empty() ? nullptr : c->slots[s].p[k].data();
    }
  // the picture into a buffer of its own (its slot may be one of its reference slots' ... never, but the oracle reads while it writes nothing outside `out`)
  OSlot out; out.w = h.width; out.h = h.height;
  out.p[0].assign( (size_t) h.width * h.height, 0 );
  for( int k = 1; k < 3; k++ ) out.p[k].assign( h.chroma_format ? (size_t) ( h.width >> 1 ) * ( h.height >> 1 ) : 0, 0 );
  uint16_t* outp[3] = { out.p[0].data(), out.p[1].empty() ? nullptr : out.p[1].data(), out.p[2].empty() ? nullptr : out.p[2].data() };
  const int job = c->nextJob++;
  {
    std::lock_guard<std::mutex> ol( g_oracle );
    // developer aid (VVR_ORACLE_USE_REF=<path of oracle/_ref/libvvref.so>): the reference's own classes, rebuilt from the description by oracle/ref_harness.cpp, instead
    // of the oracle - tells a flaw of the flattening (both differ from the decoder) from one of the oracle's arithmetic (only the oracle differs)
    typedef int ( *RefFn )( const vvr_picture*, const uint16_t* const*, uint16_t* const*, vvr_lfp* const*, int32_t*, int, double* );
    static RefFn refFn = []() -> RefFn { const char* e = getenv( "VVR_ORACLE_USE_REF" ); void* hnd = e ? dlopen( e, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND ) : nullptr; return hnd ? (RefFn) dlsym( hnd, "vvref_reconstruct" ) : nullptr; }();
    if( refFn ) { if( refFn( &P, refs.data(), outp, nullptr, nullptr, 1 /* SIMD */, nullptr ) != 0 ) { typedef const char* ( *ErrFn )(); static ErrFn ef = (ErrFn) dlsym( dlopen( getenv( "VVR_ORACLE_USE_REF" ), RTLD_NOW | RTLD_NOLOAD ), "vvref_last_error" ); c->err = std::string( "reference harness: " ) + ( ef ? ef() : "?" ); c->status[job] = VVR_ERR_PARAMETER; return job; } }
    else if( vvo_reconstruct( &P, refs.data(), outp, 0 ) != 0 ) { c->err = std::string( "oracle: " ) + vvo_last_error(); c->status[job] = VVR_ERR_PARAMETER; return job; }
    std::vector<int32_t>& d = c->dmvr[job];
    d.assign( 2 * (size_t) ( ( h.width / 16 + 1 ) * ( h.height / 16 + 1 ) * 4 ), 0 );
    const uint32_t n = vvo_get_dmvr( d.data(), (uint32_t) ( d.size() / 2 ) );
    d.resize( 2 * (size_t) n );
  }
  if( const char* dir = getenv( "VVR_ORACLE_DUMP_DIR" ) )
  {
    // developer aid: the picture as the back-end got it (arrays as raw files, the layout of tests/golden_io.py), its reference pictures and what came out -
    // tools/replay_oracle_dump.py runs the reference's classes (oracle/_ref harness) and the oracle on it in a process of their own
    const int w4 = ( h.width + 3 ) >> 2, h4 = ( h.height + 3 ) >> 2, ctu = 1 << h.log2_ctu, numCtu = ( ( h.width + ctu - 1 ) / ctu ) * ( ( h.height + ctu - 1 ) / ctu );
    auto put = [&]( const char* name, const void* p, size_t n ) { if( !p || !n ) return; char f[512]; snprintf( f, sizeof( f ), "%s/poc%d_%s.bin", dir, h.poc, name ); FILE* o = fopen( f, "wb" ); if( o ) { fwrite( p, 1, n, o ); fclose( o ); } };
    put( "hdr", &P.hdr, sizeof( P.hdr ) ); put( "cu", P.cu, sizeof( vvr_cu ) * P.num_cu ); put( "tu", P.tu, sizeof( vvr_tu ) * P.num_tu ); put( "ctu_first_cu", P.ctu_first_cu, sizeof( uint32_t ) * ( numCtu + 1 ) );
    put( "coef", P.coef, sizeof( int16_t ) * P.num_coef ); put( "lfp0", P.lfp[0], sizeof( vvr_lfp ) * w4 * h4 ); put( "lfp1", P.lfp[1], sizeof( vvr_lfp ) * w4 * h4 );
    put( "motion", P.motion, sizeof( vvr_motion ) * w4 * h4 ); put( "sao", P.sao, sizeof( vvr_sao_ctu ) * numCtu ); put( "alf", P.alf, sizeof( vvr_alf_ctu ) * numCtu );
    put( "alf_sets", P.alf_params, sizeof( vvr_alf_params ) * ( P.num_alf_sets ? P.num_alf_sets : 1 ) ); put( "lmcs", P.lmcs, sizeof( vvr_lmcs_params ) );
    put( "wp_sets", P.wp, sizeof( vvr_wp_params ) * ( P.num_wp_sets ? P.num_wp_sets : 1 ) ); put( "scaling", P.scaling, sizeof( vvr_scaling_list ) );
    for( int l = 0; l < 2 && h.slice_type != 2; l++ ) for( int i = 0; i < h.num_ref[l]; i++ ) for( int k = 0; k < 3; k++ )
    { const int sl = h.ref_slot[l][i]; char nm[64]; snprintf( nm, sizeof( nm ), "ref_%d_%d", sl, k ); put( nm, c->slots[sl].p[k].data(), sizeof( uint16_t ) * c->slots[sl].p[k].size() ); }
    for( int k = 0; k < 3; k++ ) { char nm[64]; snprintf( nm, sizeof( nm ), "out_%d", k ); put( nm, out.p[k].data(), sizeof( uint16_t ) * out.p[k].size() ); }
  }
  c->slots[h.out_slot] = std::move( out );
  c->status[job] = VVR_OK;
  return job;
}
API int vvr_wait( vvr_context* c, int job ) { if( !c ) return VVR_ERR_PARAMETER; std::lock_guard<std::mutex> lk( c->mu ); auto it = c->status.find( job ); return it == c->status.end() ? VVR_ERR_PARAMETER : it->second; }
API int vvr_test( vvr_context* c, int job ) { return vvr_wait( c, job ); }
API int vvr_read_dmvr( vvr_context* c, int job, int32_t* dst, size_t n )
{
  if( !c ) return VVR_ERR_PARAMETER;
  std::lock_guard<std::mutex> lk( c->mu );
  auto it = c->dmvr.find( job );
  if( it == c->dmvr.end() ) return VVR_ERR_PARAMETER;
  for( size_t i = 0; i < 2 * n; i++ ) dst[i] = i < it->second.size() ? it->second[i] : 0;
  return (int) ( it->second.size() / 2 );      // (as the product: the number of entries the picture has)
}
API int vvr_read_picture( vvr_context* c, int slot, uint16_t* const* dst, const size_t* stride, int )
{
  if( !c || slot < 0 || slot >= (int) c->slots.size() ) return VVR_ERR_PARAMETER;
  std::lock_guard<std::mutex> lk( c->mu );
  const OSlot& s = c->slots[slot];
  for( int k = 0; k < 3; k++ )
  {
    if( s.p[k].empty() || !dst[k] ) continue;
    const int w = k ? s.w >> 1 : s.w, h = k ? s.h >> 1 : s.h;
    for( int y = 0; y < h; y++ ) memcpy( dst[k] + (size_t) y * stride[k], &s.p[k][(size_t) y * w], sizeof( uint16_t ) * w );
  }
  return VVR_OK;
}
}
