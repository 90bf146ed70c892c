// TEST INFRASTRUCTURE (tests/ only; never shipped, never loaded by the product): the ten entry points of include/vvr.h that the drop-in decoder library calls,
// served by the CPU ORACLE (oracle/libvvoracle.so, the plain-C restatement of the reconstruction path that the GPU parity tests pin the HIP kernels to).  With this
// library bound instead of libvvdec_amd.so the reference's application decodes a bitstream END TO END ON THE CPU through integration/DecLibReconDropIn.cpp and
// integration/vvr_extract.h: the output MD5 then checks the flattening of what the real parser left in CodingStructure - and, where the description leaves the
// deblocking edge parameters to the back-end (VVR_TOOL_LFP_ON_DEVICE), vvdec_amd/csrc/vvr_lf_init.h, the source of k_lf_init - against the reference decoder,
// on as many streams as one likes and without a GPU (tests/test_dropin_library.py, tools/fuzz_dropin_on_the_oracle.py).  No sample is computed by product code here.
#include <cstdint>
#include <cstring>
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
    if( vvo_reconstruct( &P, refs.data(), outp, 0 ) != 0 ) { c->err = std::string( "oracle: " ) + vvo_last_error(); c->status[job] = VVR_ERR_PARAMETER; return job; }
    std::vector<int32_t>& d = c->dmvr[job];
    d.assign( 2 * (size_t) ( ( h.width / 16 + 1 ) * ( h.height / 16 + 1 ) * 4 ), 0 );
    const uint32_t n = vvo_get_dmvr( d.data(), (uint32_t) ( d.size() / 2 ) );
    d.resize( 2 * (size_t) n );
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
  return VVR_OK;
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
