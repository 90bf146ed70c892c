// vvdec_amd/csrc/vvr_api.cpp — host side of the reconstruction back-end: the C ABI of include/vvr.h.
//
// Mirrors DecLibRecon (reference: source/Lib/DecoderLib/DecLibRecon.cpp): create() owns the per-instance resources
// (there: per-thread tool objects + scratch, :132-168; here: HIP streams, DPB planes, scratch planes, constant tables),
// vvr_submit() is decompressPicture() (:429) — it turns the parsed picture into device work lists and enqueues the
// kernels in the stage order of the CTU state machine (:732-1110) — and vvr_wait() is waitForPrevDecompressedPic() (:684).
// Several pictures are in flight on separate HIP streams; inter-picture dependencies (reference pictures, :544-581
// "refPicExtDepBarriers") are whole-picture HIP events.  There is NO CPU fallback: without a gfx950 device every entry
// point fails with VVR_ERR_NO_DEVICE.
#include "vvr_device.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

int vvr_upload_tables();

#define HIPCHK( ctx, call ) do { hipError_t e_ = ( call ); if( e_ != hipSuccess ) { ( ctx )->setError( std::string( #call ) + ": " + hipGetErrorString( e_ ) ); return VVR_ERR_DEVICE; } } while( 0 )

namespace {
static inline int ilog2i( int v ) { int l = 0; while( ( 1 << l ) < v ) l++; return l; }

struct Stat { uint64_t launches = 0; double ms = 0, bytes = 0; };

struct PendingTiming { hipEvent_t a, b; int kernel; double bytes; };

enum { K_MC, K_MC_DMVR, K_MC_AFFINE, K_LMCS, K_ITRANS, K_INTRA, K_DEBLOCK_V, K_DEBLOCK_H, K_SAO, K_ALF, K_COPY, K_NUM };
const char* const kKernelNames[K_NUM] = { "k_mc", "k_mc_dmvr", "k_mc_affine", "k_lmcs", "k_itrans", "k_intra", "k_deblock_v", "k_deblock_h", "k_sao", "k_alf", "k_copy" };

struct DevBuf {
  void* p = nullptr; size_t n = 0;
};

}   // namespace

struct vvr_prepared {        // a picture description resident in HBM together with its device work lists
  vvr_pic_header hdr;
  PicDev   pic;
  DevBuf   blob;             // one allocation holding every array
  McItem*  mcItems = nullptr; int numMc = 0;
  McItem*  bdofItems = nullptr; int numBdofItems = 0;      // tiles of CUs in BDOF mode (their own launch: larger LDS footprint)
  McItem*  dmvrItems = nullptr; int numDmvrItems = 0;      // sub-blocks that run decoder-side MV refinement
  McItem*  affItems = nullptr; int numAffItems = 0;        // tiles of affine CUs
  int32_t* dmvrOut = nullptr; uint32_t numDmvr = 0;        // delta MVs, device (inside the blob)
  TbItem*  tbItems[3] = { nullptr, nullptr, nullptr }; int numTb[3] = { 0, 0, 0 };   // size classes 16 / 32 / 64 (TB_ADD: after MC)
  IntraItem* intraItems = nullptr; uint32_t* ctuStart = nullptr; IntraUnit* units = nullptr; int numActive = 0, numIntra = 0;
  std::vector<std::pair<int, int>> intraLevels;     // non-empty: the intra stage runs as one launch per dependency level (first unit, count)
  double   bytes[K_NUM] = { 0 };
  bool     owned = false;
};

struct vvr_context {
  vvr_config cfg;
  int        device = 0;
  std::string err;
  std::vector<hipStream_t> streams;
  std::vector<DevPlanes>   slots;       // DPB
  std::vector<DevPlanes>   scratchB;    // per stream: second picture (SAO output)
  std::vector<DevPlanes>   scratchR;    // per stream: residual planes (intra)
  void*      planeMem = nullptr; bool planeMemOwned = false;
  void*      scratchMem = nullptr;
  std::vector<int*> syncBuf;            // per stream: ticket + one flag per unit of the intra stage
  std::vector<size_t> syncCap;          // ints allocated in syncBuf[lane]; grown when a picture has more units (ensureSync)
  size_t     planeBytes[3] = { 0, 0, 0 }, slotBytes = 0;
  int        stride[3] = { 0, 0, 0 };
  // jobs
  struct Job { int id; int stream; hipEvent_t done; bool waited; vvr_prepared* autoFree; std::vector<PendingTiming> timings;
               vvr_prepared* prepared = nullptr; std::vector<int32_t> dmvr; };     // dmvr: delta MVs copied to the host when the job is waited for
  std::vector<Job> jobs;
  int        nextJob = 0, nextStream = 0;
  std::vector<std::vector<int>> slotUsers;   // job ids that touched a slot since it was last written
  bool       statsOn = false;
  Stat       stats[K_NUM];
  void setError( const std::string& e ) { err = e; }
};

static size_t alignUp( size_t v, size_t a ) { return ( v + a - 1 ) / a * a; }

static void planeGeometry( const vvr_config* cfg, int stride[3], size_t bytes[3], size_t* total )
{
  const int ncomp = cfg->chroma_format ? 3 : 1;
  size_t t = 0;
  for( int c = 0; c < 3; c++ )
  {
    if( c >= ncomp ) { stride[c] = 0; bytes[c] = 0; continue; }
    const int w = c ? cfg->max_width >> 1 : cfg->max_width, h = c ? cfg->max_height >> 1 : cfg->max_height;
    stride[c] = (int) alignUp( (size_t) w, 64 );                  // 128-byte rows
    bytes[c]  = alignUp( (size_t) stride[c] * h * sizeof( pel_t ), 256 );
    t += bytes[c];
  }
  *total = t;
}

static DevPlanes carve( char* base, const vvr_config* cfg, const int stride[3], const size_t bytes[3] )
{
  DevPlanes d; memset( &d, 0, sizeof( d ) );
  const int ncomp = cfg->chroma_format ? 3 : 1;
  size_t off = 0;
  for( int c = 0; c < ncomp; c++ )
  {
    d.p[c] = (pel_t*) ( base + off ); off += bytes[c];
    d.stride[c] = stride[c];
    d.w[c] = c ? cfg->max_width >> 1 : cfg->max_width; d.h[c] = c ? cfg->max_height >> 1 : cfg->max_height;
  }
  return d;
}

extern "C" {

VVR_API const char* vvr_version( void ) { return "vvdec_amd 0.1 (gfx950, ABI 1)"; }

VVR_API size_t vvr_abi_sizeof( int which )
{
  static const size_t sz[] = { sizeof( vvr_pic_header ), sizeof( vvr_cu ), sizeof( vvr_tu ), sizeof( vvr_motion ), sizeof( vvr_lfp ), sizeof( vvr_sao_ctu ),
                               sizeof( vvr_alf_ctu ), sizeof( vvr_alf_params ), sizeof( vvr_lmcs_params ), sizeof( vvr_picture ), sizeof( vvr_config ), sizeof( vvr_kernel_stat ),
                               sizeof( vvr_wp_params ), sizeof( vvr_scaling_list ) };
  return which >= 0 && which < (int) ( sizeof( sz ) / sizeof( sz[0] ) ) ? sz[which] : 0;
}

VVR_API size_t vvr_slot_bytes( const vvr_config* cfg )
{
  int st[3]; size_t b[3], t; planeGeometry( cfg, st, b, &t ); return t;
}

VVR_API int vvr_create( const vvr_config* cfg, vvr_context** out )
{
  if( !cfg || !out || cfg->abi_version != VVR_ABI_VERSION ) return VVR_ERR_PARAMETER;
  // Main 10: 4:0:0 / 4:2:0, 8..10-bit samples (the formats the parity tests cover); CTU 32..128
  if( cfg->chroma_format > 1 || cfg->bit_depth < 8 || cfg->bit_depth > 10 || cfg->log2_ctu < 5 || cfg->log2_ctu > 7 || !cfg->num_slots ) return VVR_ERR_UNSUPPORTED;
  int ndev = 0;
  if( hipGetDeviceCount( &ndev ) != hipSuccess || ndev <= 0 || cfg->device >= ndev ) return VVR_ERR_NO_DEVICE;
  vvr_context* c = new vvr_context();
  c->cfg = *cfg; c->device = cfg->device;
  if( hipSetDevice( cfg->device ) != hipSuccess ) { delete c; return VVR_ERR_NO_DEVICE; }
  {
    hipDeviceProp_t prop;
    if( hipGetDeviceProperties( &prop, cfg->device ) != hipSuccess ) { delete c; return VVR_ERR_NO_DEVICE; }
    if( strncmp( prop.gcnArchName, "gfx950", 6 ) != 0 ) { fprintf( stderr, "vvdec_amd: device %d is %s, this library is built for gfx950 only\n", cfg->device, prop.gcnArchName ); delete c; return VVR_ERR_NO_DEVICE; }
  }
  if( vvr_upload_tables() != 0 ) { delete c; return VVR_ERR_DEVICE; }
  const int ns = std::max<int>( 1, cfg->num_streams );
  c->streams.resize( ns );
  for( int i = 0; i < ns; i++ ) if( hipStreamCreateWithFlags( &c->streams[i], hipStreamNonBlocking ) != hipSuccess ) { delete c; return VVR_ERR_DEVICE; }
  planeGeometry( cfg, c->stride, c->planeBytes, &c->slotBytes );
  if( cfg->ext_planes ) c->planeMem = cfg->ext_planes;
  else { if( hipMalloc( &c->planeMem, c->slotBytes * cfg->num_slots ) != hipSuccess ) { delete c; return VVR_ERR_DEVICE; } c->planeMemOwned = true; hipMemset( c->planeMem, 0, c->slotBytes * cfg->num_slots ); }
  if( hipMalloc( &c->scratchMem, c->slotBytes * 2 * ns ) != hipSuccess ) { delete c; return VVR_ERR_DEVICE; }
  for( int s = 0; s < cfg->num_slots; s++ ) c->slots.push_back( carve( (char*) c->planeMem + c->slotBytes * s, cfg, c->stride, c->planeBytes ) );
  for( int s = 0; s < ns; s++ )
  {
    c->scratchB.push_back( carve( (char*) c->scratchMem + c->slotBytes * ( 2 * s ), cfg, c->stride, c->planeBytes ) );
    c->scratchR.push_back( carve( (char*) c->scratchMem + c->slotBytes * ( 2 * s + 1 ), cfg, c->stride, c->planeBytes ) );
  }
  {
    const int ctu = 1 << cfg->log2_ctu;
    const size_t numCtu = (size_t) ( ( cfg->max_width + ctu - 1 ) / ctu ) * ( ( cfg->max_height + ctu - 1 ) / ctu );
    // sized for the usual pictures (a 4K B picture of the benchmark has about 6 units per CTU, an intra picture 3); pictures with more
    // units than that (many isolated small intra CUs) make the lane's buffer grow when they are submitted
    const size_t perCtu = getenv( "VVR_SYNC_UNITS_PER_CTU" ) ? (size_t) std::max( 1, atoi( getenv( "VVR_SYNC_UNITS_PER_CTU" ) ) ) : 24;
    for( int s = 0; s < ns; s++ ) { int* p = nullptr; const size_t cap = 1 + perCtu * numCtu; if( hipMalloc( (void**) &p, sizeof( int ) * cap ) != hipSuccess ) { delete c; return VVR_ERR_DEVICE; } c->syncBuf.push_back( p ); c->syncCap.push_back( cap ); }
  }
  c->slotUsers.resize( cfg->num_slots );
  *out = c;
  return VVR_OK;
}

VVR_API int vvr_sync( vvr_context* c );

VVR_API void vvr_destroy( vvr_context* c )
{
  if( !c ) return;
  hipSetDevice( c->device );
  vvr_sync( c );
  for( auto& j : c->jobs ) if( j.done ) hipEventDestroy( j.done );
  for( auto s : c->streams ) hipStreamDestroy( s );
  if( c->planeMemOwned && c->planeMem ) hipFree( c->planeMem );
  if( c->scratchMem ) hipFree( c->scratchMem );
  for( auto p : c->syncBuf ) hipFree( p );
  delete c;
}

VVR_API const char* vvr_last_error( const vvr_context* c ) { return c ? c->err.c_str() : "no context"; }

VVR_API int vvr_plane_layout( const vvr_context* c, int comp, size_t* offset, size_t* stride_bytes, int* width, int* height )
{
  if( !c || comp < 0 || comp > 2 ) return VVR_ERR_PARAMETER;
  size_t off = 0; for( int k = 0; k < comp; k++ ) off += c->planeBytes[k];
  if( offset ) *offset = off;
  if( stride_bytes ) *stride_bytes = (size_t) c->stride[comp] * sizeof( pel_t );
  if( width ) *width = comp ? c->cfg.max_width >> 1 : c->cfg.max_width;
  if( height ) *height = comp ? c->cfg.max_height >> 1 : c->cfg.max_height;
  return VVR_OK;
}

VVR_API void* vvr_plane_ptr( vvr_context* c, int slot, int comp )
{
  if( !c || slot < 0 || slot >= (int) c->slots.size() || comp < 0 || comp > 2 ) return nullptr;
  return c->slots[slot].p[comp];
}

VVR_API int vvr_read_plane( vvr_context* c, int slot, int comp, uint16_t* dst, size_t dstStride )
{
  if( !c || slot < 0 || slot >= (int) c->slots.size() || comp < 0 || comp > 2 || !c->slots[slot].p[comp] ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  vvr_sync( c );
  const DevPlanes& d = c->slots[slot];
  HIPCHK( c, hipMemcpy2D( dst, dstStride * 2, d.p[comp], (size_t) d.stride[comp] * 2, (size_t) d.w[comp] * 2, d.h[comp], hipMemcpyDeviceToHost ) );
  return VVR_OK;
}

VVR_API int vvr_read_output( vvr_context* c, int slot, int comp, int x, int y, int w, int h, int bytesPerSample, void* dst, size_t dstStrideBytes )
{
  if( !c || slot < 0 || slot >= (int) c->slots.size() || comp < 0 || comp > 2 || !c->slots[slot].p[comp] || !dst ) return VVR_ERR_PARAMETER;
  const DevPlanes& d = c->slots[slot];
  if( x < 0 || y < 0 || w <= 0 || h <= 0 || x + w > d.w[comp] || y + h > d.h[comp] || ( bytesPerSample != 1 && bytesPerSample != 2 ) || dstStrideBytes < (size_t) w * bytesPerSample )
  { c->setError( "vvr_read_output: window outside the plane, bad sample size or stride" ); return VVR_ERR_PARAMETER; }
  if( bytesPerSample == 1 && c->cfg.bit_depth > 8 ) { c->setError( "vvr_read_output: 8-bit output of a stream with more than 8 bits per sample (only narrowing of 8-bit content, vvdecimpl.cpp:853)" ); return VVR_ERR_PARAMETER; }
  hipSetDevice( c->device );
  vvr_sync( c );
  const pel_t* src = d.p[comp] + (size_t) y * d.stride[comp] + x;
  if( bytesPerSample == 2 )
  {
    HIPCHK( c, hipMemcpy2D( dst, dstStrideBytes, src, (size_t) d.stride[comp] * 2, (size_t) w * 2, h, hipMemcpyDeviceToHost ) );
    return VVR_OK;
  }
  // 8-bit frames: the window comes over as 16-bit samples, the low bytes are packed on the host (what copyComp does with its SSE loop)
  std::vector<uint16_t> tmp( (size_t) w * h );
  HIPCHK( c, hipMemcpy2D( tmp.data(), (size_t) w * 2, src, (size_t) d.stride[comp] * 2, (size_t) w * 2, h, hipMemcpyDeviceToHost ) );
  for( int r = 0; r < h; r++ ) { uint8_t* o = (uint8_t*) dst + (size_t) r * dstStrideBytes; const uint16_t* in = tmp.data() + (size_t) r * w; for( int k = 0; k < w; k++ ) o[k] = (uint8_t) in[k]; }
  return VVR_OK;
}

// ---- decoded picture hash (SEI decoded_picture_hash; reference: PicYuvMD5.cpp).  MD5 after RFC 1321.
namespace {
struct Md5
{
  uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u; uint64_t len = 0; uint8_t buf[64]; size_t fill = 0;
  static uint32_t rol( uint32_t v, int s ) { return ( v << s ) | ( v >> ( 32 - s ) ); }
  void block( const uint8_t* p )
  {
    static const uint32_t K[64] = {
      0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
      0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
      0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
      0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391 };
    static const int S[64] = { 7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22, 5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20, 4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23, 6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21 };
    uint32_t m[16]; for( int i = 0; i < 16; i++ ) m[i] = (uint32_t) p[4 * i] | ( (uint32_t) p[4 * i + 1] << 8 ) | ( (uint32_t) p[4 * i + 2] << 16 ) | ( (uint32_t) p[4 * i + 3] << 24 );
    uint32_t A = a, B = b, C = c, D = d;
    for( int i = 0; i < 64; i++ )
    {
      uint32_t f; int g;
      if( i < 16 ) { f = ( B & C ) | ( ~B & D ); g = i; } else if( i < 32 ) { f = ( D & B ) | ( ~D & C ); g = ( 5 * i + 1 ) & 15; }
      else if( i < 48 ) { f = B ^ C ^ D; g = ( 3 * i + 5 ) & 15; } else { f = C ^ ( B | ~D ); g = ( 7 * i ) & 15; }
      const uint32_t t = D; D = C; C = B; B = B + rol( A + f + K[i] + m[g], S[i] ); A = t;
    }
    a += A; b += B; c += C; d += D;
  }
  void update( const uint8_t* p, size_t n )
  {
    len += n;
    while( n ) { const size_t k = std::min( n, 64 - fill ); memcpy( buf + fill, p, k ); fill += k; p += k; n -= k; if( fill == 64 ) { block( buf ); fill = 0; } }
  }
  void finish( uint8_t out[16] )
  {
    const uint64_t bits = len * 8; const uint8_t one = 0x80, zero = 0;
    update( &one, 1 ); while( fill != 56 ) update( &zero, 1 );
    uint8_t l[8]; for( int i = 0; i < 8; i++ ) l[i] = (uint8_t) ( bits >> ( 8 * i ) );
    update( l, 8 );
    const uint32_t v[4] = { a, b, c, d }; for( int i = 0; i < 16; i++ ) out[i] = (uint8_t) ( v[i >> 2] >> ( 8 * ( i & 3 ) ) );
  }
};
}

VVR_API int vvr_picture_hash( vvr_context* c, int slot, int method, uint8_t* digest, int* digestLen )
{
  if( !c || slot < 0 || slot >= (int) c->slots.size() || !digest || method < VVR_HASH_MD5 || method > VVR_HASH_CHECKSUM ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  vvr_sync( c );
  const DevPlanes& d = c->slots[slot];
  const int nc = c->cfg.chroma_format ? 3 : 1, len = method == VVR_HASH_MD5 ? 16 : method == VVR_HASH_CRC ? 2 : 4;
  const bool two = c->cfg.bit_depth > 8;
  std::vector<uint16_t> pl;
  for( int k = 0; k < nc; k++ )
  {
    const int w = d.w[k], h = d.h[k];
    pl.resize( (size_t) w * h );
    HIPCHK( c, hipMemcpy2D( pl.data(), (size_t) w * 2, d.p[k], (size_t) d.stride[k] * 2, (size_t) w * 2, h, hipMemcpyDeviceToHost ) );
    uint8_t* out = digest + (size_t) k * len;
    if( method == VVR_HASH_MD5 )
    {
      Md5 m; std::vector<uint8_t> row( (size_t) w * 2 );
      for( int y = 0; y < h; y++ )
      {
        const uint16_t* s = pl.data() + (size_t) y * w; size_t n = 0;
        for( int x = 0; x < w; x++ ) { row[n++] = (uint8_t) s[x]; if( two ) row[n++] = (uint8_t) ( s[x] >> 8 ); }
        m.update( row.data(), n );
      }
      m.finish( out );
    }
    else if( method == VVR_HASH_CRC )
    {
      // CRC-16 with polynomial 0x1021 over the bytes of every sample (low byte first), most significant bit first, 16 zero bits appended (:99-137)
      uint32_t crc = 0xffff;
      auto feed = [&]( uint32_t byte ) { for( int bit = 7; bit >= 0; bit-- ) { const uint32_t msb = ( crc >> 15 ) & 1; crc = ( ( ( crc << 1 ) + ( ( byte >> bit ) & 1 ) ) & 0xffff ) ^ ( msb * 0x1021 ); } };
      for( size_t i = 0; i < pl.size(); i++ ) { feed( pl[i] & 0xff ); if( two ) feed( pl[i] >> 8 ); }
      for( int bit = 0; bit < 16; bit++ ) { const uint32_t msb = ( crc >> 15 ) & 1; crc = ( ( crc << 1 ) & 0xffff ) ^ ( msb * 0x1021 ); }
      out[0] = (uint8_t) ( crc >> 8 ); out[1] = (uint8_t) crc;
    }
    else
    {
      // 32-bit sum of the sample bytes, each xor-ed with a mask of its position (:152-181)
      uint32_t sum = 0;
      for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
      {
        const uint32_t mask = ( x & 0xff ) ^ ( y & 0xff ) ^ ( x >> 8 ) ^ ( y >> 8 ), v = pl[(size_t) y * w + x];
        sum += ( ( v & 0xff ) ^ mask ) & 0xff;
        if( two ) sum += ( ( v >> 8 ) ^ mask ) & 0xffffffffu;
      }
      out[0] = (uint8_t) ( sum >> 24 ); out[1] = (uint8_t) ( sum >> 16 ); out[2] = (uint8_t) ( sum >> 8 ); out[3] = (uint8_t) sum;
    }
  }
  if( digestLen ) *digestLen = len;
  return VVR_OK;
}

VVR_API int vvr_write_plane( vvr_context* c, int slot, int comp, const uint16_t* src, size_t srcStride )
{
  if( !c || slot < 0 || slot >= (int) c->slots.size() || comp < 0 || comp > 2 || !c->slots[slot].p[comp] ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  vvr_sync( c );
  const DevPlanes& d = c->slots[slot];
  HIPCHK( c, hipMemcpy2D( d.p[comp], (size_t) d.stride[comp] * 2, src, srcStride * 2, (size_t) d.w[comp] * 2, d.h[comp], hipMemcpyHostToDevice ) );
  return VVR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// vvr_prepare: validation + host glue (work lists) + upload.  This is the only place that touches host arrays.
// ---------------------------------------------------------------------------------------------------------------------
static int validate( vvr_context* c, const vvr_picture* p )
{
  const vvr_pic_header& h = p->hdr;
  if( h.abi_version != VVR_ABI_VERSION ) { c->setError( "abi_version mismatch" ); return VVR_ERR_PARAMETER; }
  if( h.width != c->cfg.max_width || h.height != c->cfg.max_height || h.chroma_format != c->cfg.chroma_format || h.bit_depth != c->cfg.bit_depth || h.log2_ctu != c->cfg.log2_ctu )
  { c->setError( "picture geometry differs from the context configuration" ); return VVR_ERR_PARAMETER; }
  if( ( h.width & 7 ) || ( h.height & 7 ) ) { c->setError( "picture size must be a multiple of 8 (minimum CU size)" ); return VVR_ERR_PARAMETER; }
  if( h.out_slot < 0 || h.out_slot >= c->cfg.num_slots ) { c->setError( "out_slot out of range" ); return VVR_ERR_PARAMETER; }
  if( ( h.tool_flags & VVR_TOOL_LMCS_CSCALE ) && !( h.tool_flags & VVR_TOOL_LMCS ) ) { c->setError( "LMCS chroma residual scaling without LMCS" ); return VVR_ERR_PARAMETER; }
  if( ( h.tool_flags & VVR_TOOL_LMCS ) && !p->lmcs ) { c->setError( "LMCS enabled without tables" ); return VVR_ERR_PARAMETER; }
  const bool wpOn = ( h.tool_flags & VVR_TOOL_WP ) && h.slice_type != 2;
  if( wpOn && !p->wp ) { c->setError( "weighted prediction enabled without the weight table" ); return VVR_ERR_PARAMETER; }
  if( wpOn && ( p->wp->log2_denom[0] > 7 || p->wp->log2_denom[1] > 7 ) ) { c->setError( "weighted prediction: log2 denominator out of range" ); return VVR_ERR_PARAMETER; }
  if( ( h.tool_flags & VVR_TOOL_SCALING_LIST ) && !p->scaling ) { c->setError( "explicit scaling lists enabled without the lists" ); return VVR_ERR_PARAMETER; }
  if( h.tool_flags & VVR_TOOL_SCALING_LIST )
    for( int id = 0; id < 28; id++ ) for( int k = 0; k < ( id < 2 ? 4 : id < 8 ? 16 : 64 ); k++ ) if( !p->scaling->coef[id][k] ) { c->setError( "scaling list entry 0" ); return VVR_ERR_PARAMETER; }
  if( !p->cu || !p->tu || !p->coef || !p->lfp[0] || !p->lfp[1] ) { c->setError( "missing arrays" ); return VVR_ERR_PARAMETER; }
  if( ( h.tool_flags & VVR_TOOL_ALF ) && ( !p->alf || !p->alf_params ) ) { c->setError( "ALF enabled without parameters" ); return VVR_ERR_PARAMETER; }
  if( ( h.tool_flags & ( VVR_TOOL_SAO_LUMA | VVR_TOOL_SAO_CHROMA ) ) && !p->sao ) { c->setError( "SAO enabled without parameters" ); return VVR_ERR_PARAMETER; }
  if( h.slice_type != 2 )
    for( int l = 0; l < 2; l++ ) for( int i = 0; i < h.num_ref[l]; i++ )
      if( h.ref_slot[l][i] < 0 || h.ref_slot[l][i] >= c->cfg.num_slots || h.ref_slot[l][i] == h.out_slot ) { c->setError( "bad reference slot" ); return VVR_ERR_PARAMETER; }
  for( uint32_t i = 0; i < p->num_cu; i++ )
  {
    const vvr_cu& cu = p->cu[i];
    if( cu.x + cu.w > h.width || cu.y + cu.h > h.height || cu.first_tu + cu.num_tu > p->num_tu ) { c->setError( "CU outside the picture / bad TU range" ); return VVR_ERR_PARAMETER; }
    if( cu.pred_mode == VVR_PRED_INTER )
    {
      const bool isDmvr = cu.mc_mode == VVR_MC_DMVR || cu.mc_mode == VVR_MC_DMVR_BDOF;
      const bool isAff = cu.mc_mode == VVR_MC_AFFINE;
      const bool isGeo = cu.mc_mode == VVR_MC_GEO;
      const bool isSbt = cu.mc_mode == VVR_MC_SBTMVP;
      if( cu.mc_mode != VVR_MC_UNI && cu.mc_mode != VVR_MC_BI && cu.mc_mode != VVR_MC_BDOF && !isDmvr && !isAff && !isGeo && !isSbt ) { c->setError( "unknown mc_mode" ); return VVR_ERR_PARAMETER; }
      if( isSbt != ( ( cu.flags & VVR_CU_SBTMVP ) != 0 ) || ( isSbt && ( !p->motion || cu.w < 8 || cu.h < 8 ) ) ) { c->setError( "SbTMVP CU: mc_mode / flag mismatch, missing motion field or CU smaller than 8x8" ); return VVR_ERR_PARAMETER; }
      if( isSbt )
        for( int y = 0; y < cu.h; y += 8 ) for( int x = 0; x < cu.w; x += 8 )
        {
          const vvr_motion& m = p->motion[(size_t) ( ( cu.y + y ) >> 2 ) * ( ( h.width + 3 ) >> 2 ) + ( ( cu.x + x ) >> 2 )];
          if( ( m.ref_idx[0] < 0 && m.ref_idx[1] < 0 ) || m.ref_idx[0] >= h.num_ref[0] || m.ref_idx[1] >= h.num_ref[1] ) { c->setError( "SbTMVP CU: bad sub-block motion" ); return VVR_ERR_PARAMETER; }
        }
      if( isGeo != ( ( cu.flags & VVR_CU_GEO ) != 0 ) ) { c->setError( "GPM CU: mc_mode / flag mismatch" ); return VVR_ERR_PARAMETER; }
      if( isGeo )
      {
        if( cu.w < 8 || cu.h < 8 || cu.w > 64 || cu.h > 64 || cu.w >= 8 * cu.h || cu.h >= 8 * cu.w || cu.geo_split_dir >= 64 ) { c->setError( "GPM CU: size / split direction out of range" ); return VVR_ERR_PARAMETER; }
        for( int k = 0; k < 2; k++ )
        {
          const int l = ( cu.geo_dir_ref[k] >> 4 ) - 1, ri = cu.geo_dir_ref[k] & 15;
          if( l < 0 || l > 1 || ri >= h.num_ref[l] ) { c->setError( "GPM CU: bad reference" ); return VVR_ERR_PARAMETER; }
        }
      }
      if( isAff != ( ( cu.flags & VVR_CU_AFFINE ) != 0 ) || ( isAff && ( !p->motion || cu.w < 8 || cu.h < 8 ) ) ) { c->setError( "affine CU: mc_mode / flag mismatch, missing motion field or CU smaller than 8x8" ); return VVR_ERR_PARAMETER; }
      if( isDmvr && ( !( h.tool_flags & VVR_TOOL_DMVR ) || ( cu.mc_mode == VVR_MC_DMVR_BDOF && !( h.tool_flags & VVR_TOOL_BDOF ) ) || cu.ref_idx[0] < 0 || cu.ref_idx[1] < 0 || cu.w < 8 || cu.h < 8 || cu.w * cu.h < 128 || cu.bcw_idx != 2 ) )
      { c->setError( "mc_mode DMVR on a CU that cannot use DMVR (UnitTools.cpp:1277)" ); return VVR_ERR_PARAMETER; }
      if( cu.mc_mode == VVR_MC_BDOF && ( !( h.tool_flags & VVR_TOOL_BDOF ) || cu.ref_idx[0] < 0 || cu.ref_idx[1] < 0 || cu.w < 8 || cu.h < 8 || cu.w * cu.h < 128 || cu.bcw_idx != 2 ) )
      { c->setError( "mc_mode BDOF on a CU that cannot use BDOF (InterPrediction.cpp:1407-1427)" ); return VVR_ERR_PARAMETER; }
      if( ( cu.flags & VVR_CU_CIIP ) && ( ( cu.mc_mode != VVR_MC_UNI && cu.mc_mode != VVR_MC_BI ) || cu.w * cu.h < 64 || cu.w > 64 || cu.h > 64 || cu.num_tu != 1 ) )
      { c->setError( "CIIP CU: needs plain uni/bi prediction, at least 64 luma samples, sides of at most 64 and one TU" ); return VVR_ERR_PARAMETER; }
      for( int l = 0; l < 2; l++ ) if( cu.ref_idx[l] >= h.num_ref[l] ) { c->setError( "ref_idx out of range" ); return VVR_ERR_PARAMETER; }
      if( cu.ref_idx[0] < 0 && cu.ref_idx[1] < 0 && !isGeo && !isSbt ) { c->setError( "inter CU without reference" ); return VVR_ERR_PARAMETER; }
      if( wpOn && cu.ref_idx[0] >= 0 && cu.ref_idx[1] >= 0 )
      {
        // weighted prediction: BDOF / DMVR only between references with default weights (InterPrediction.cpp:1420, UnitTools.cpp:1297-1302);
        // no identical-motion shortcut (:408)
        bool present = false;
        for( int l = 0; l < 2; l++ ) for( int k = 0; k < 3; k++ ) present |= p->wp->e[l][cu.ref_idx[l]][k].present != 0;
        if( present && ( isDmvr || cu.mc_mode == VVR_MC_BDOF ) ) { c->setError( "mc_mode BDOF / DMVR between references with explicit prediction weights" ); return VVR_ERR_PARAMETER; }
        if( cu.mc_mode == VVR_MC_UNI ) { c->setError( "mc_mode UNI on a bi-predicted CU of a picture with weighted prediction" ); return VVR_ERR_PARAMETER; }
      }
      if( cu.tree != VVR_TREE_JOINT && h.chroma_format ) { c->setError( "inter CU must be single tree" ); return VVR_ERR_PARAMETER; }
      if( cu.w == 4 && cu.h == 4 ) { c->setError( "4x4 inter CU (never inter predicted, InterPrediction.cpp:634)" ); return VVR_ERR_PARAMETER; }
    }
    else if( cu.pred_mode == VVR_PRED_INTRA )
    {
      if( cu.isp_mode )
      {
        // intra sub-partitions (CU::canUseISP, UnitTools.cpp): luma split in four, chroma unsplit in the last TU
        const uint32_t np = ( ( cu.w == 4 && cu.h == 8 ) || ( cu.w == 8 && cu.h == 4 ) ) ? 2 : 4;      // 4x8 / 8x4: two partitions
        if( cu.isp_mode > 2 || cu.multi_ref_idx || cu.bdpcm[0] || ( cu.flags & VVR_CU_MIP ) || cu.num_tu != np || cu.w * cu.h <= 16 ) { c->setError( "ISP CU: bad split mode, combined with MRL / BDPCM / MIP, or wrong number of TUs" ); return VVR_ERR_PARAMETER; }
        for( uint32_t k = 0; k < np; k++ )
        {
          const vvr_tu& t4 = p->tu[cu.first_tu + k];
          const bool ok = cu.isp_mode == 1 ? ( t4.x == cu.x && t4.w == cu.w && t4.h * (int) np == cu.h && t4.y == cu.y + (int) k * t4.h ) : ( t4.y == cu.y && t4.h == cu.h && t4.w * (int) np == cu.w && t4.x == cu.x + (int) k * t4.w );
          if( !ok || ( t4.comp_mask & 6 ) != ( k == np - 1 && h.chroma_format && cu.tree == VVR_TREE_JOINT ? 6 : 0 ) || t4.mts_idx[0] == VVR_MTS_SKIP ) { c->setError( "ISP CU: TU layout" ); return VVR_ERR_PARAMETER; }
        }
      }
      if( cu.flags & VVR_CU_MIP )
      {
        const int sizeId = ( cu.w == 4 && cu.h == 4 ) ? 0 : ( cu.w == 4 || cu.h == 4 || ( cu.w == 8 && cu.h == 8 ) ) ? 1 : 2;
        if( cu.intra_dir[0] >= ( sizeId == 0 ? 16 : sizeId == 1 ? 8 : 6 ) || cu.multi_ref_idx || cu.bdpcm[0] ) { c->setError( "MIP CU: mode index out of range for the block size, or combined with MRL / BDPCM" ); return VVR_ERR_PARAMETER; }
      }
      if( h.chroma_format && cu.tree != VVR_TREE_LUMA && cu.intra_dir[1] > 69 ) { c->setError( "chroma intra mode out of range" ); return VVR_ERR_PARAMETER; }
      {
        // luma-tree CUs of dual-tree pictures go down to 4x4; a 4-wide CU with chroma (2xN chroma blocks / local dual tree) is not in this build
        // luma-tree CUs go down to 4x4; CUs with chroma need 8 luma samples of width (no 2-wide intra chroma blocks) and 4 of height
        const int minW = cu.tree == VVR_TREE_LUMA ? 4 : 8;
        if( cu.w > 64 || cu.h > 64 || cu.w < minW || cu.h < 4 || ( cu.tree != VVR_TREE_LUMA && cu.w * cu.h < 64 ) ) { c->setError( "intra CU size out of range (luma tree 4..64, with chroma at least 8 wide and 16 chroma samples)" ); return VVR_ERR_PARAMETER; }
      }
      if( cu.tree != VVR_TREE_JOINT )
      {
        // dual tree (I slices, qtbtt_dual_tree_intra_flag) and local dual tree (intra-only sub-trees of small blocks in any slice):
        // luma CUs carry luma blocks only, chroma CUs chroma blocks only
        if( cu.tree > VVR_TREE_CHROMA || !h.chroma_format ) { c->setError( "bad tree type" ); return VVR_ERR_PARAMETER; }
        const int want = cu.tree == VVR_TREE_LUMA ? 1 : 6;
        for( uint32_t t = cu.first_tu; t < cu.first_tu + cu.num_tu; t++ ) if( p->tu[t].comp_mask != want ) { c->setError( "separate-tree CU: TU component mask" ); return VVR_ERR_PARAMETER; }
        if( cu.tree == VVR_TREE_CHROMA && ( cu.isp_mode || cu.multi_ref_idx || cu.bdpcm[0] || ( cu.flags & VVR_CU_MIP ) ) ) { c->setError( "chroma-tree CU with luma tools" ); return VVR_ERR_PARAMETER; }
      }
      if( cu.intra_dir[0] > 66 || cu.multi_ref_idx > 2 || cu.bdpcm[1] ) { c->setError( "bad intra mode / chroma BDPCM not implemented" ); return VVR_ERR_UNSUPPORTED; }
    }
    else if( cu.pred_mode == VVR_PRED_IBC )
    {
      // intra block copy (InterPrediction::xIntraBlockCopy, InterPrediction.cpp:1995): integer block vector in mv[0][0], luma at most 64x64
      // (IBC_MAX_CU_SIZE), one TU, no intra / inter tools; sizes as for intra CUs.  That the reference block precedes the CU in decoding order
      // is checked where the work lists are built.
      if( !( h.tool_flags & VVR_TOOL_IBC ) ) { c->setError( "IBC CU in a picture without VVR_TOOL_IBC" ); return VVR_ERR_PARAMETER; }
      const int minW = cu.tree == VVR_TREE_LUMA ? 4 : 8;
      if( cu.tree == VVR_TREE_CHROMA || cu.w > 64 || cu.h > 64 || cu.w < minW || cu.h < 4 || ( cu.tree != VVR_TREE_LUMA && cu.w * cu.h < 64 ) || cu.num_tu != 1 )
      { c->setError( "IBC CU: chroma tree, size out of range or more than one TU" ); return VVR_ERR_PARAMETER; }
      if( ( cu.mv[0][0][0] | cu.mv[0][0][1] ) & 15 ) { c->setError( "IBC CU: fractional block vector" ); return VVR_ERR_PARAMETER; }
      if( cu.isp_mode || cu.bdpcm[0] || cu.bdpcm[1] || cu.lfnst_idx || cu.sbt_info || ( cu.flags & ( VVR_CU_MIP | VVR_CU_CIIP | VVR_CU_AFFINE | VVR_CU_GEO | VVR_CU_SBTMVP ) ) )
      { c->setError( "IBC CU combined with an intra / inter tool" ); return VVR_ERR_PARAMETER; }
      if( cu.tree == VVR_TREE_LUMA && h.chroma_format && p->tu[cu.first_tu].comp_mask != 1 ) { c->setError( "separate-tree CU: TU component mask" ); return VVR_ERR_PARAMETER; }
      const int bvx = cu.mv[0][0][0] >> 4, bvy = cu.mv[0][0][1] >> 4, ctuS = 1 << h.log2_ctu, rowTop = cu.y & ~( ctuS - 1 );
      const int bufW = 256 * 128 / ctuS;                              // width of the IBC virtual buffer (Rom.h:210, CodingStructure.cpp:543)
      bool ok = cu.x + bvx >= 0 && cu.x + bvx + cu.w <= h.width && cu.y + bvy >= rowTop && cu.y + bvy + cu.h <= std::min<int>( h.height, rowTop + ctuS )
             && cu.x + bvx + cu.w <= ( ( cu.x >> h.log2_ctu ) + 1 ) * ctuS && cu.x + bvx >= ( cu.x & ~( ctuS - 1 ) ) - ( bufW - ctuS );
      if( ok && cu.tree == VVR_TREE_JOINT && h.chroma_format )
      {
        const int cxr = ( cu.x >> 1 ) + ( bvx >> 1 ), cyr = ( cu.y >> 1 ) + ( bvy >> 1 );
        ok = cxr >= 0 && cxr + ( cu.w >> 1 ) <= ( h.width >> 1 ) && cyr >= ( rowTop >> 1 ) && cyr + ( cu.h >> 1 ) <= std::min<int>( h.height, rowTop + ctuS ) >> 1 && 2 * cxr >= ( cu.x & ~( ctuS - 1 ) ) - ( bufW - ctuS );
      }
      if( !ok ) { c->setError( "IBC CU: reference block outside the picture, the CTU row or the reach of the IBC buffer" ); return VVR_ERR_PARAMETER; }
    }
    else { c->setError( "unknown prediction mode" ); return VVR_ERR_PARAMETER; }
  }
  return VVR_OK;
}

VVR_API void vvr_free_prepared( vvr_context* c, vvr_prepared* q )
{
  if( !q ) return;
  if( c ) hipSetDevice( c->device );
  if( q->blob.p ) hipFree( q->blob.p );
  delete q;
}

namespace {
// vvr_prepare in stages: everything a picture's device work lists are built from lives in one object; the stages run in the order of
// the member functions below (each keeps the reference citations of the code it holds)
struct PicturePreparer
{
  vvr_context* const c; const vvr_picture* const p; const vvr_pic_header& h;
  const int ncomp; const bool wpOn; const int w4, h4, ctu, ctusX, ctusY, numCtu;
  // ---- host glue: work lists (what DecCu::TaskTrafoCtu / TaskInterCtu iterate over, DecCu.cpp:106-134)
  std::vector<McItem> mc, mcBdof, mcDmvr, mcAff;
  uint32_t numDmvr = 0;
  std::vector<TbItem> tb[3];
  std::vector<IntraItem> intra[3];
  std::vector<uint32_t> ctuStartV;
  double bytes[K_NUM] = { 0 };
  // decode-order index of the transform block covering every 4x4 luma unit (both channel types): reference availability
  // = "inside the picture and reconstructed before me" (CodingStructure::getCURestricted, CodingStructure.cpp:464, and the
  // TU-index test of isAboveAvailable / isLeftAvailable, IntraPrediction.cpp:1343-1400)
  std::vector<int32_t> order;
  std::vector<uint8_t> intraAt;          // per 4x4 luma unit: covered by an intra CU
  // Intra-stage work units: a unit is a set of blocks of one (component, CTU) that are connected through the reference samples they
  // read from each other (a whole CTU in an intra picture, a few blocks around an isolated intra CU in a B picture); one workgroup
  // processes one unit, its blocks in coding order.  Units depend on exactly those other units that produced a sample they read
  // (inter samples are final before the stage starts).  The loop below records per block which blocks it reads from (itemAt[]);
  // the units are formed afterwards.
  struct BBox { int y0 = 255, y1 = 0, c0 = 255, c1 = 0; };   // rows relative to (CTU top - 3), 8-sample chunks relative to (CTU left - 8), chunk index + 1
  struct UnitH { uint32_t comp, ctu, i0, i1, iA = 0; bool hasCs = false; BBox bb; std::vector<uint32_t> deps; bool waited = false; int rank = 0; };
  struct ItemH { uint32_t ctu; BBox bb; std::vector<uint32_t> prod; };      // prod: ( component << 28 ) | item index of the blocks it reads from
  std::vector<UnitH> units;
  std::vector<ItemH> itemH[3];
  std::vector<int32_t> itemAt[3];        // per component and 4x4 luma cell: the block that reconstructs it in the intra stage (-1: none)
  // LMCS chroma residual scaling: per VPDU the luma neighbourhood its factor is averaged over (Reshape::calculateChromaAdjVpduNei,
  // Reshape.cpp:192-274): left column / above row of the CU at the VPDU origin, where that neighbour precedes it in decoding order
  const bool cscale;
  const int vpduLog2, vpdusX, vpdusY;
  std::vector<uint32_t> csVpduV;
  std::vector<IntraItem> intraAll;
  std::vector<IntraUnit> unitsDev;
  std::vector<std::pair<int, int>> intraLevelsV;

  PicturePreparer( vvr_context* c_, const vvr_picture* p_ )
    : c( c_ ), p( p_ ), h( p_->hdr ), ncomp( h.chroma_format ? 3 : 1 ), wpOn( ( h.tool_flags & VVR_TOOL_WP ) && h.slice_type != 2 ),
      w4( ( h.width + 3 ) >> 2 ), h4( ( h.height + 3 ) >> 2 ), ctu( 1 << h.log2_ctu ), ctusX( ( h.width + ctu - 1 ) / ctu ), ctusY( ( h.height + ctu - 1 ) / ctu ), numCtu( ctusX * ctusY ),
      ctuStartV( 3 * (size_t) ( numCtu + 1 ), 0 ),
      cscale( ( h.tool_flags & VVR_TOOL_LMCS ) && ( h.tool_flags & VVR_TOOL_LMCS_CSCALE ) && ncomp == 3 ),
      vpduLog2( std::min<int>( 6, h.log2_ctu ) ), vpdusX( ( h.width + ( 1 << vpduLog2 ) - 1 ) >> vpduLog2 ), vpdusY( ( h.height + ( 1 << vpduLog2 ) - 1 ) >> vpduLog2 ) {}

  int unitAvail( int chn, int x, int y, int32_t cur ) const
  {
    const int cs = chn ? 1 : 0, lx = x << cs, ly = y << cs;
    if( x < 0 || y < 0 || lx >= h.width || ly >= h.height ) return 0;
    return order[(size_t) chn * w4 * h4 + ( ly >> 2 ) * w4 + ( lx >> 2 )] < cur;
  }

  // decoding order of the transform blocks, cells covered by intra CUs, the luma neighbourhood of every VPDU's chroma scaling factor
  int mapDecodingOrder()
  {
    bool anyIntra = ( h.tool_flags & VVR_TOOL_LMCS_CSCALE ) != 0;      // (inter blocks with scaled chroma residuals are intra-stage items)
    for( uint32_t i = 0; i < p->num_cu && !anyIntra; i++ ) anyIntra = p->cu[i].pred_mode == VVR_PRED_INTRA || p->cu[i].pred_mode == VVR_PRED_IBC || ( p->cu[i].flags & VVR_CU_CIIP );
    if( anyIntra )
    {
      order.assign( (size_t) w4 * h4 * 2, 0x7fffffff );
      intraAt.assign( (size_t) w4 * h4, 0 );
      for( int k = 0; k < ncomp; k++ ) itemAt[k].assign( (size_t) w4 * h4, -1 );
      for( uint32_t i = 0; i < p->num_cu; i++ )
      {
        const vvr_cu& cu = p->cu[i];
        // 1: intra CU, 2: CIIP CU (inter prediction blended with planar intra in the intra stage, DecCu.cpp:137-140,453-456)
        const bool ciip = cu.pred_mode == VVR_PRED_INTER && ( cu.flags & VVR_CU_CIIP );
        if( cu.pred_mode == VVR_PRED_INTRA || ciip )
          for( int y = cu.y; y < cu.y + cu.h; y += 4 ) for( int x = cu.x; x < cu.x + cu.w; x += 4 ) intraAt[(size_t) ( y >> 2 ) * w4 + ( x >> 2 )] = ciip ? 2 : 1;
        for( uint32_t t = cu.first_tu; t < cu.first_tu + cu.num_tu; t++ )
        {
          const vvr_tu& tu = p->tu[t];
          for( int chn = 0; chn < 2; chn++ )
          {
            if( chn == 0 && !( tu.comp_mask & 1 ) ) continue;
            if( chn == 1 && !( tu.comp_mask & 6 ) ) continue;
            int ax = tu.x, ay = tu.y, aw = tu.w, ah = tu.h;
            if( chn == 1 && cu.isp_mode ) { ax = cu.x; ay = cu.y; aw = cu.w; ah = cu.h; }      // ISP: the unsplit chroma blocks sit in the last TU
            for( int y = ay; y < ay + ah && y < h.height; y += 4 ) for( int x = ax; x < ax + aw && x < h.width; x += 4 )
              order[(size_t) chn * w4 * h4 + ( y >> 2 ) * w4 + ( x >> 2 )] = (int32_t) t;
          }
        }
      }
    }
    if( cscale )
    {
      std::vector<int32_t> cuAt( (size_t) w4 * h4, -1 );
      for( uint32_t i = 0; i < p->num_cu; i++ )
      {
        const vvr_cu& cu = p->cu[i];
        if( cu.tree == VVR_TREE_CHROMA ) continue;                     // dual tree: the luma CUs
        for( int y = cu.y; y < cu.y + cu.h; y += 4 ) for( int x = cu.x; x < cu.x + cu.w; x += 4 ) cuAt[(size_t) ( y >> 2 ) * w4 + ( x >> 2 )] = (int32_t) i;
      }
      csVpduV.resize( (size_t) vpdusX * vpdusY );
      for( int vy = 0; vy < vpdusY; vy++ ) for( int vx = 0; vx < vpdusX; vx++ )
      {
        const int32_t tl = cuAt[(size_t) ( ( vy << vpduLog2 ) >> 2 ) * w4 + ( ( vx << vpduLog2 ) >> 2 )];
        const int xPos = p->cu[tl].x, yPos = p->cu[tl].y;
        bool hasLeft = xPos > 0, hasAbove = yPos > 0;
        if( hasLeft && ( ( xPos - 1 ) >> h.log2_ctu ) == ( xPos >> h.log2_ctu ) && cuAt[(size_t) ( yPos >> 2 ) * w4 + ( ( xPos - 1 ) >> 2 )] > tl ) hasLeft = false;
        if( hasAbove && ( ( yPos - 1 ) >> h.log2_ctu ) == ( yPos >> h.log2_ctu ) && cuAt[(size_t) ( ( yPos - 1 ) >> 2 ) * w4 + ( xPos >> 2 )] > tl ) hasAbove = false;
        csVpduV[(size_t) vy * vpdusX + vx] = (uint32_t) xPos | ( (uint32_t) yPos << 13 ) | ( hasLeft ? 1u << 26 : 0 ) | ( hasAbove ? 1u << 27 : 0 );
      }
    }
    return VVR_OK;
  }

  // the work lists: intra-stage blocks with the blocks they read from, motion-compensation tiles, transform blocks
  int buildWorkLists()
  {
    uint32_t curCtu = 0;
    for( uint32_t i = 0; i < p->num_cu; i++ )
    {
      const vvr_cu& cu = p->cu[i];
      // CTU bookkeeping for the per-CTU intra lists (CUs arrive in CTU raster order)
      const uint32_t ctuOfCu = (uint32_t) ( ( cu.y >> h.log2_ctu ) * ctusX + ( cu.x >> h.log2_ctu ) );
      {
        if( ctuOfCu < curCtu ) { c->setError( "CUs are not in CTU raster order" ); return VVR_ERR_PARAMETER; }
        while( curCtu < ctuOfCu ) { curCtu++; for( int k = 0; k < 3; k++ ) ctuStartV[(size_t) k * ( numCtu + 1 ) + curCtu] = (uint32_t) intra[k].size(); }
      }
      const bool isCiipCu = cu.pred_mode == VVR_PRED_INTER && ( cu.flags & VVR_CU_CIIP );
      // LMCS chroma residual scaling of an inter block: its factor reads reconstructed luma that the intra stage may still have to
      // produce, and intra blocks next to it read its reconstructed chroma, so the residual add of such a block is an item of the
      // intra stage too (IT_MODE_RESI_ADD: no prediction, scaled residual onto the inter prediction; finishLMCSAndReco, DecCu.cpp:483)
      const bool isCsInterCu = cscale && cu.pred_mode == VVR_PRED_INTER && ( !isCiipCu || cu.w == 4 ) && ( cu.flags & VVR_CU_ROOT_CBF );
      // intra block copy: the block is a copy of reconstructed samples of this picture that the intra stage may still have to produce, so it
      // is an item of the intra stage as well (IT_MODE_IBC; the reference does it in its intra task too, DecCu.cpp:145)
      const bool isIbcCu = cu.pred_mode == VVR_PRED_IBC;
      if( cu.pred_mode == VVR_PRED_INTRA || isCiipCu || isCsInterCu || isIbcCu )
      {
        for( uint32_t t = cu.first_tu; t < cu.first_tu + cu.num_tu; t++ )
        {
          const vvr_tu& tu = p->tu[t];
          for( int comp = 0; comp < ncomp; comp++ )
          {
            if( !( tu.comp_mask & ( 1 << comp ) ) ) continue;
            // the 2-wide chroma blocks of a 4-wide CIIP CU are not blended (predBlendIntraCiip, IntraPrediction.cpp:891): plain inter blocks
            const bool isCiip = isCiipCu && !( comp && cu.w == 4 );
            const bool isCsInter = cscale && cu.pred_mode == VVR_PRED_INTER && !isCiip && ( cu.flags & VVR_CU_ROOT_CBF );
            if( cu.pred_mode != VVR_PRED_INTRA && !isCiip && !isCsInter && !isIbcCu ) continue;
            if( isCsInter && ( !comp || !( ( ( tu.cbf >> comp ) & 1 ) || tu.joint_cbcr ) || ( tu.w >> 1 ) * ( tu.h >> 1 ) <= 4 ) ) continue;
            const int cs = comp ? 1 : 0, chn = comp ? 1 : 0, unit = 4 >> cs;
            // intra sub-partitions: luma partitions are blocks of their own that share the reference line of the whole CU
            // (initIntraPatternChTypeISP, IntraPrediction.cpp:966); partitions narrower than 4 are predicted in pairs (DecCu.cpp:333-371):
            // one item of width 4 carries both; the unsplit chroma blocks come with the last TU
            const bool ispL = cu.isp_mode && !comp, ispC = cu.isp_mode && comp;
            const bool ispPair = ispL && cu.isp_mode == 2 && tu.w < 4;                              // group of 4 / tu.w partitions
            if( ispPair && ( ( tu.x - cu.x ) & 3 ) ) continue;                                      // not the first of its group: part of the group's item
            const int x0 = ( ispC ? cu.x : tu.x ) >> cs, y0 = ( ispC ? cu.y : tu.y ) >> cs, w = ispPair ? 4 : ( ispC ? cu.w : tu.w ) >> cs, hh = ( ispC ? cu.h : tu.h ) >> cs;
            // block whose neighbourhood decides the availability of the reference samples
            const int rx0 = ispL ? cu.x : x0, ry0 = ispL ? cu.y : y0, rw = ispL ? cu.w : w, rh = ispL ? cu.h : hh;
            const int32_t rcur = ispL ? (int32_t) cu.first_tu : (int32_t) t;
            const int totalAbove = ( 2 * rw + unit - 1 ) / unit, totalLeft = ( 2 * rh + unit - 1 ) / unit;
            IntraItem it; memset( &it, 0, sizeof( it ) );
            it.tu = t; it.comp = (uint8_t) comp;
            it.x = (uint16_t) x0; it.y = (uint16_t) y0;
            { int l = 0; while( ( 1 << l ) < w ) l++; it.lw = (uint8_t) l; l = 0; while( ( 1 << l ) < hh ) l++; it.lh = (uint8_t) l; }
            it.mode = isIbcCu ? IT_MODE_IBC : isCsInter ? IT_MODE_RESI_ADD : isCiip ? 0 : cu.intra_dir[chn];       // CIIP: planar
            // IBC: the block vector in samples of the component (chroma: halved, InterPrediction.cpp:2010-2011)
            const int ibcDx = isIbcCu ? ( cu.mv[0][0][0] >> 4 ) >> cs : 0, ibcDy = isIbcCu ? ( cu.mv[0][0][1] >> 4 ) >> cs : 0;
            if( isIbcCu ) it.tu = ( (uint32_t) ibcDx & 0xffff ) | ( (uint32_t) ibcDy << 16 );
            bool hasResi = ( ( tu.cbf >> comp ) & 1 ) || ( comp && tu.joint_cbcr );
            if( ispL )
            {
              // residual flags of the partitions of a group (2 of width 2, or 4 of width 1), geometry of the partition inside its CU
              uint32_t mask = tu.cbf & 1, grp = 0;
              if( ispPair )
              {
                grp = tu.w == 2 ? 1 : 2;
                for( uint32_t k = 1; k < 4u / tu.w && t + k < cu.first_tu + cu.num_tu; k++ ) mask |= (uint32_t) ( p->tu[t + k].cbf & 1 ) << k;
              }
              it.tu = (uint32_t) ( tu.x - cu.x ) | ( (uint32_t) ( tu.y - cu.y ) << 6 ) | ( (uint32_t) ilog2i( cu.w ) << 12 ) | ( (uint32_t) ilog2i( cu.h ) << 15 )
                    | ( (uint32_t) ( cu.isp_mode == 2 ) << 18 ) | ( mask << 19 ) | ( grp << 23 );
              hasResi = mask != 0;
            }
            const int bdp = ( isCiip || isCsInter || isIbcCu ) ? 0 : cu.bdpcm[chn];
            // CIIP blend weight of the intra part (IntraPrediction::predBlendIntraCiip, IntraPrediction.cpp:925-929): 1 + intra neighbours
            const int wIntra = isCiip ? 1 + ( cu.ciip_neigh_intra & 1 ) + ( ( cu.ciip_neigh_intra >> 1 ) & 1 ) : 0;
            it.flags = (uint8_t) ( ( hasResi ? IT_F_RESI : 0 ) | ( bdp == 1 ? IT_F_BDPCM_H : bdp == 2 ? IT_F_BDPCM_V : 0 ) | ( ( comp || isCiip || isCsInter || isIbcCu ? 0 : cu.multi_ref_idx ) << 4 ) | ( wIntra << 6 ) );
            if( !comp && !isCiip && ( cu.flags & VVR_CU_MIP ) ) it.flags = (uint8_t) ( ( hasResi ? IT_F_RESI : 0 ) | IT_F_MIP | ( ( cu.flags & VVR_CU_MIP_TRANSP ) ? 0x10 : 0 ) );
            if( ispL ) it.flags = (uint8_t) ( ( hasResi ? IT_F_RESI : 0 ) | IT_F_ISP );
            const bool noRef = isCsInter || isIbcCu;                                  // no intra reference lines
            if( !noRef ) it.nTL = (uint8_t) unitAvail( chn, rx0 - 1, ry0 - 1, rcur );
            if( !noRef && unitAvail( chn, rx0, ry0 - 1, rcur ) ) { int n = rw / unit; for( int k = 0; k < totalAbove - rw / unit; k++ ) { if( !unitAvail( chn, rx0 + rw + k * unit, ry0 - 1, rcur ) ) break; n++; } it.nA = (uint8_t) n; }
            if( !noRef && unitAvail( chn, rx0 - 1, ry0, rcur ) ) { int n = rh / unit; for( int k = 0; k < totalLeft - rh / unit; k++ ) { if( !unitAvail( chn, rx0 - 1, ry0 + rh + k * unit, rcur ) ) break; n++; } it.nL = (uint8_t) n; }
            int cclmTop = 0, cclmLeft = 0, cclmBLeft = 0; bool isCclm = false;
            const bool csItem = cscale && comp && hasResi && w * hh > 4;             // DecCu.cpp:383-388 / :500-505
            if( csItem ) it.flags |= IT_F_CSCALE;
            if( comp && !isCiip && !isCsInter && !isIbcCu && cu.intra_dir[1] >= 67 )
            {
              // CCLM / MDLM: template sizes and flags of IntraPrediction::xGetLMParameters (:1694-1800) and the border handling of
              // xGetLumaRecPixels (:1403-1470); they ride in the item's `tu` word
              const int mode = cu.intra_dir[1];
              const bool aboveCu = cu.y > 0 || ( y0 << 1 ) > cu.y, leftCu = cu.x > 0 || ( x0 << 1 ) > cu.x;          // cu.above / cu.left (one slice, one tile)
              const int tuWU = w / unit, tuHU = hh / unit;
              const int totA = ( 2 * w + unit - 1 ) / unit, totL = ( 2 * hh + unit - 1 ) / unit;
              int aboveAvail = 0, leftAvail = 0, actualTop = 0, actualLeft = 0;
              if( mode == 69 )
              {
                int avai = 0;
                if( aboveCu ) { avai = tuWU; const int lim = std::min( totA - tuWU, hh / unit ); for( int k = 0; k < lim; k++ ) { if( !unitAvail( 1, x0 + w + k * unit, y0 - 1, (int32_t) t ) ) break; avai++; } }
                aboveAvail = avai >= tuWU; actualTop = unit * avai;
              }
              else if( mode == 68 )
              {
                int avai = 0;
                if( leftCu ) { avai = tuHU; const int lim = std::min( totL - tuHU, w / unit ); for( int k = 0; k < lim; k++ ) { if( !unitAvail( 1, x0 - 1, y0 + hh + k * unit, (int32_t) t ) ) break; avai++; } }
                leftAvail = avai >= tuHU; actualLeft = unit * avai;
              }
              else { aboveAvail = aboveCu; leftAvail = leftCu; actualTop = w; actualLeft = hh; }
              const int bLeft = leftCu ? 1 : 0;                                                          // availlableLeftUnit >= iTUHeightInUnits
              const int firstRow = ( ( y0 << 1 ) & ( ( 1 << h.log2_ctu ) - 1 ) ) == 0;
              it.tu = (uint32_t) actualTop | ( (uint32_t) actualLeft << 8 ) | ( (uint32_t) aboveAvail << 16 ) | ( (uint32_t) leftAvail << 17 ) | ( (uint32_t) bLeft << 18 ) | ( (uint32_t) firstRow << 19 ) | ( (uint32_t) ( aboveCu ? 1 : 0 ) << 20 );
              cclmTop = aboveAvail ? actualTop : 0; cclmLeft = leftAvail ? actualLeft : 0; cclmBLeft = bLeft; isCclm = true;
            }
            // ---- the blocks this one reads from, its part of the CTU tile
            const uint32_t myId = (uint32_t) intra[comp].size();
            intra[comp].push_back( it );
            itemH[comp].emplace_back();
            ItemH& IH = itemH[comp].back();
            IH.ctu = ctuOfCu;
            {
              const int ctuX = cu.x >> h.log2_ctu, ctuY = cu.y >> h.log2_ctu;
              const int mrl = ( comp || isIbcCu ) ? 0 : cu.multi_ref_idx;
              auto touch = [&]( int k, int xc, int yc )   // component k, component coordinates of a sample that is read
              {
                const int sh = k ? 1 : 0, lx = xc << sh, ly = yc << sh;
                if( lx < 0 || ly < 0 || lx >= h.width || ly >= h.height ) return;
                const int32_t d = itemAt[k][(size_t) ( ly >> 2 ) * w4 + ( lx >> 2 )];
                if( d < 0 ) return;
                const uint32_t key = ( (uint32_t) k << 28 ) | (uint32_t) d;
                if( ( k != comp || (uint32_t) d != myId ) && std::find( IH.prod.begin(), IH.prod.end(), key ) == IH.prod.end() ) IH.prod.push_back( key );
              };
              {
                // bounding box of everything the kernel's reference fill may read for this block (whole top / left lines incl. padding sources)
                const int S = ( 1 << h.log2_ctu ) >> cs, ox = ctuX * S, oy = ctuY * S;
                BBox& bb = IH.bb;
                const int bx0 = rx0 - 1 - mrl, bx1 = rx0 + std::max( 2 * rw, 1 ) + 1, by0 = ry0 - 1 - mrl, by1 = ry0 + 2 * rh + 1;
                bb.y0 = std::min( bb.y0, std::max( 0, by0 - ( oy - 3 ) ) );
                bb.y1 = std::max( bb.y1, std::min( S + 3, by1 - ( oy - 3 ) ) );
                bb.c0 = std::min( bb.c0, std::max( 0, ( bx0 - ( ox - 8 ) ) >> 3 ) );
                bb.c1 = std::max( bb.c1, std::min( ( 8 + S + 64 + 7 ) >> 3, ( bx1 - ( ox - 8 ) + 7 ) >> 3 ) );
              }
              if( it.nTL ) touch( comp, rx0 - 1 - mrl, ry0 - 1 - mrl );
              for( int k = 0; k < it.nA * unit; k += unit ) touch( comp, rx0 + k, ry0 - 1 - mrl );
              for( int k = 0; k < it.nL * unit; k += unit ) touch( comp, rx0 - 1 - mrl, ry0 + k );
              if( ispL && ( x0 != rx0 || y0 != ry0 ) ) touch( 0, cu.isp_mode == 2 ? x0 - 1 : x0, cu.isp_mode == 2 ? y0 : y0 - 1 );   // ISP: the previous partition
              if( isIbcCu )
              {
                // the reference block: every cell must precede this block in decoding order; the intra-stage blocks that produce it are producers
                const int qx = x0 + ibcDx, qy = y0 + ibcDy;
                for( int yy = 0; yy < hh + unit - 1; yy += unit ) for( int xx = 0; xx < w + unit - 1; xx += unit )
                {
                  const int sx = qx + std::min( xx, w - 1 ), sy = qy + std::min( yy, hh - 1 );
                  if( !unitAvail( chn, sx, sy, (int32_t) t ) ) { c->setError( "IBC CU: the reference block is not reconstructed before the CU" ); return VVR_ERR_PARAMETER; }
                  touch( comp, sx, sy );
                }
              }
              if( csItem )
              {
                // luma the chroma scaling factor is averaged over (the unit must wait for the luma units that reconstruct it)
                const uint32_t d = csVpduV[(size_t) ( tu.y >> vpduLog2 ) * vpdusX + ( tu.x >> vpduLog2 )];
                const int xPos = d & 0x1fff, yPos = ( d >> 13 ) & 0x1fff, n = 1 << vpduLog2;
                if( ( d >> 26 ) & 1 ) for( int k = 0; k < n; k += 4 ) touch( 0, xPos - 1, std::min( yPos + k, (int) h.height - 1 ) );
                if( ( d >> 27 ) & 1 ) for( int k = 0; k < n; k += 4 ) touch( 0, std::min( xPos + k, (int) h.width - 1 ), yPos - 1 );
              }
              if( isCclm )
              {
                // luma the prediction reads: the co-located block and the template rows / columns around it (luma coordinates)
                const int lx0 = x0 << 1, ly0 = y0 << 1;
                for( int yy = 0; yy < 2 * hh; yy += 4 ) for( int xx = ( cclmBLeft ? -4 : 0 ); xx < 2 * w; xx += 4 ) touch( 0, lx0 + xx, ly0 + yy );
                for( int xx = ( cclmBLeft ? -4 : 0 ); xx < 2 * cclmTop + 4; xx += 4 ) touch( 0, lx0 + xx, ly0 - 1 );
                for( int yy = 0; yy < 2 * cclmLeft + 4; yy += 4 ) touch( 0, lx0 - 1, ly0 + yy );
              }
              // the cells this block reconstructs
              for( int yy = 0; yy < ( hh << cs ); yy += 4 ) for( int xx = 0; xx < ( w << cs ); xx += 4 )
                if( ( x0 << cs ) + xx < h.width && ( y0 << cs ) + yy < h.height ) itemAt[comp][(size_t) ( ( ( y0 << cs ) + yy ) >> 2 ) * w4 + ( ( ( x0 << cs ) + xx ) >> 2 )] = (int32_t) myId;
            }
            bytes[K_INTRA] += (double) w * hh * ( hasResi ? 4 : 2 ) + sizeof( IntraItem );
          }
        }
      }
      if( cu.pred_mode == VVR_PRED_INTER )
      {
        const int nl = cu.mc_mode == VVR_MC_UNI ? 1 : 2;     // (SbTMVP: upper bound, sub-blocks may be uni-directional)
        const bool sbt = cu.mc_mode == VVR_MC_SBTMVP;
        const int ts = sbt ? 8 : 16;                       // SbTMVP: one item per 8x8 sub-block (ATMVP_SUB_BLOCK_SIZE)
        for( int y = 0; y < cu.h; y += ts ) for( int x = 0; x < cu.w; x += ts )
        {
          McItem it; memset( &it, 0, sizeof( it ) );
          it.x = (uint16_t) ( cu.x + x ); it.y = (uint16_t) ( cu.y + y ); it.w = (uint8_t) std::min( ts, cu.w - x ); it.h = (uint8_t) std::min( ts, cu.h - y ); it.flags = sbt ? MC_ITEM_SUBBLOCK : 0; it.cu = i;
          const bool dm = cu.mc_mode == VVR_MC_DMVR || cu.mc_mode == VVR_MC_DMVR_BDOF;
          const bool af = cu.mc_mode == VVR_MC_AFFINE;
          if( !dm && !af )
          {
            // everything k_mc needs about the motion of the tile
            bool uni = cu.mc_mode == VVR_MC_UNI;
            it.ref[0] = cu.ref_idx[0]; it.ref[1] = cu.ref_idx[1];
            for( int l = 0; l < 2; l++ ) { it.mv[l][0] = cu.mv[l][0][0]; it.mv[l][1] = cu.mv[l][0][1]; }
            it.clipX = cu.x; it.clipY = cu.y;
            if( sbt )
            {
              // SbTMVP (xSubPuMC, InterPrediction.cpp:438): the motion of the 8x8 sub-block from the motion field, the identical-motion
              // shortcut (xCheckIdenticalMotion :404, not with weighted bi-prediction :408) decided per sub-block, clipped at its own position
              const vvr_motion& m = p->motion[(size_t) ( it.y >> 2 ) * w4 + ( it.x >> 2 )];
              for( int l = 0; l < 2; l++ ) { it.ref[l] = m.ref_idx[l]; it.mv[l][0] = m.mv[l][0]; it.mv[l][1] = m.mv[l][1]; }
              const bool two = it.ref[0] >= 0 && it.ref[1] >= 0;
              uni = !two || ( h.ref_poc[0][it.ref[0]] == h.ref_poc[1][it.ref[1]] && it.mv[0][0] == it.mv[1][0] && it.mv[0][1] == it.mv[1][1] && !wpOn );
              it.clipX = it.x; it.clipY = it.y;
            }
            it.bcw = cu.bcw_idx;
            it.flags |= ( uni ? MC_ITEM_UNI : 0 ) | ( cu.imv == 3 ? MC_ITEM_HPEL : 0 ) | ( cu.mc_mode == VVR_MC_GEO ? MC_ITEM_GEO : 0 );
          }
          ( dm ? mcDmvr : af ? mcAff : cu.mc_mode == VVR_MC_BDOF ? mcBdof : mc ).push_back( it );
          const double smp = (double) it.w * it.h * ( ncomp == 3 ? 1.5 : 1.0 );
          const int nla = af ? ( ( cu.ref_idx[0] >= 0 && cu.ref_idx[1] >= 0 ) ? 2 : 1 ) : nl;
          bytes[dm ? K_MC_DMVR : af ? K_MC_AFFINE : K_MC] += smp * 2 * nla + smp * 2 + sizeof( McItem ) + ( dm ? 8 : 0 ) + ( af ? it.w * it.h / 16.0 * sizeof( vvr_motion ) : 0 );
        }
        if( cu.mc_mode == VVR_MC_DMVR || cu.mc_mode == VVR_MC_DMVR_BDOF ) numDmvr = std::max<uint32_t>( numDmvr, cu.dmvr_off + ( ( cu.w + 15 ) / 16 ) * ( ( cu.h + 15 ) / 16 ) );
        bytes[K_MC] += sizeof( vvr_cu );
      }
      if( !( cu.flags & VVR_CU_ROOT_CBF ) ) continue;
      for( uint32_t t = cu.first_tu; t < cu.first_tu + cu.num_tu; t++ )
      {
        const vvr_tu& tu = p->tu[t];
        for( int comp = 0; comp < ncomp; comp++ )
        {
          if( !( tu.comp_mask & ( 1 << comp ) ) ) continue;
          TbItem it; it.tu = t; it.comp = (uint8_t) comp; it.ict = 0; it.pad = 0;
          it.mode = ( cu.pred_mode == VVR_PRED_INTER && ( !( cu.flags & VVR_CU_CIIP ) || ( comp && cu.w == 4 ) ) ) ? TB_ADD : TB_STORE;      // (2-wide chroma of a 4-wide CIIP CU: plain inter)
          if( comp && tu.joint_cbcr )
          {
            if( comp != 1 ) continue;
            static const int ict[2][4] = { { 0, 3, 1, 2 }, { 0, -3, -1, -2 } };           // g_ictModes (Rom.cpp:409)
            it.comp = (uint8_t) ( ( tu.joint_cbcr >> 1 ) ? 1 : 2 );
            it.ict = (uint8_t) ( 4 + ict[( h.tool_flags & VVR_TOOL_JCCR_SIGN ) ? 1 : 0][tu.joint_cbcr] );
          }
          else if( !( tu.cbf & ( 1 << comp ) ) ) continue;
          const int bw = ( ( it.comp && cu.isp_mode ) ? cu.w : tu.w ) >> ( it.comp ? 1 : 0 ), bh = ( ( it.comp && cu.isp_mode ) ? cu.h : tu.h ) >> ( it.comp ? 1 : 0 );
          if( ( bw < 2 || bh < 2 ) && !( cu.isp_mode && !it.comp && bw * bh >= 16 ) ) { c->setError( "1-D transform block outside an ISP CU" ); return VVR_ERR_PARAMETER; }
          const int cls = std::max( bw, bh ) <= 16 ? 0 : std::max( bw, bh ) <= 32 ? 1 : 2;
          // LMCS chroma residual scaling of an inter block: the factor needs the reconstructed luma around the VPDU, which the intra stage
          // may still have to produce, so the block's residual is stored and added (scaled) by a residual-add item of the intra stage
          if( cscale && it.comp && it.mode == TB_ADD && bw * bh > 4 ) it.mode = TB_STORE;      // added (scaled) by the intra stage, see isCsInter above
          tb[cls].push_back( it );        // ADD (inter: onto the prediction) and STORE (intra / CIIP: into the residual planes) items share a launch
          const int bdp = it.comp ? cu.bdpcm[1] : cu.bdpcm[0];
          const double ncoef = bdp ? (double) bw * bh : (double) ( tu.max_scan_x[it.comp] + 1 ) * ( tu.max_scan_y[it.comp] + 1 );
          bytes[K_ITRANS] += ncoef * 2 + (double) bw * bh * 4 * ( it.ict ? 2 : 1 ) + sizeof( TbItem ) + sizeof( vvr_tu ) / 3.0;
        }
      }
    }
    return VVR_OK;
  }

  int formUnits()
  {
    // ---- form the units: blocks of one (component, CTU) that read from each other belong together (union-find); the residual-add items of
    // inter blocks (LMCS chroma scaling) of a (component, CTU) form a unit of their own that the kernel processes in parallel
    for( int k = 0; k < ncomp; k++ )
    {
      const size_t n = intra[k].size();
      if( !n ) continue;
      std::vector<uint32_t> parent( n );
      for( size_t i = 0; i < n; i++ ) parent[i] = (uint32_t) i;
      auto find = [&]( uint32_t a ) { while( parent[a] != a ) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
      auto unite = [&]( uint32_t a, uint32_t b ) { a = find( a ); b = find( b ); if( a != b ) parent[std::max( a, b )] = std::min( a, b ); };     // root = first block
      int64_t bulk = -1; uint32_t bulkCtu = 0;
      for( size_t i = 0; i < n; i++ )
      {
        const bool ra = intra[k][i].mode == IT_MODE_RESI_ADD && k;
        if( ra ) { if( bulk >= 0 && bulkCtu == itemH[k][i].ctu ) unite( (uint32_t) bulk, (uint32_t) i ); else { bulk = (int64_t) i; bulkCtu = itemH[k][i].ctu; } continue; }
        for( uint32_t key : itemH[k][i].prod )
        {
          const uint32_t pk = key >> 28, pi = key & 0x0fffffff;
          if( (int) pk == k && itemH[k][pi].ctu == itemH[k][i].ctu && !( intra[k][pi].mode == IT_MODE_RESI_ADD && k ) ) unite( (uint32_t) i, pi );
        }
      }
      // units in the order of their first block; blocks of a unit contiguous and in coding order
      std::vector<int32_t> unitOfRoot( n, -1 );
      std::vector<std::vector<uint32_t>> members;
      std::vector<uint32_t> firstUnit( 1, (uint32_t) units.size() );
      for( size_t i = 0; i < n; i++ )
      {
        const uint32_t r = find( (uint32_t) i );
        if( unitOfRoot[r] < 0 ) { unitOfRoot[r] = (int32_t) members.size(); members.emplace_back(); }
        members[unitOfRoot[r]].push_back( (uint32_t) i );
      }
      std::vector<IntraItem> sorted; sorted.reserve( n );
      std::vector<ItemH> sortedH; sortedH.reserve( n );
      std::vector<uint32_t> newIdx( n );
      for( auto& m : members )
      {
        UnitH u; u.comp = (uint32_t) k; u.ctu = itemH[k][m[0]].ctu; u.i0 = (uint32_t) sorted.size();
        for( uint32_t i : m )
        {
          newIdx[i] = (uint32_t) sorted.size();
          sorted.push_back( intra[k][i] ); sortedH.push_back( std::move( itemH[k][i] ) );
          const BBox& b = sortedH.back().bb;
          u.bb.y0 = std::min( u.bb.y0, b.y0 ); u.bb.y1 = std::max( u.bb.y1, b.y1 ); u.bb.c0 = std::min( u.bb.c0, b.c0 ); u.bb.c1 = std::max( u.bb.c1, b.c1 );
          if( k && ( intra[k][i].flags & IT_F_CSCALE ) ) u.hasCs = true;
        }
        u.i1 = (uint32_t) sorted.size();
        u.iA = ( k && sorted[u.i0].mode == IT_MODE_RESI_ADD ) ? u.i1 : u.i0;
        units.push_back( u );
      }
      intra[k].swap( sorted ); itemH[k].swap( sortedH );
      // item index -> unit, kept for the dependency pass (old index space -> new)
      for( auto& ih : itemH[k] ) for( uint32_t& key : ih.prod ) if( (int) ( key >> 28 ) == k ) key = ( (uint32_t) k << 28 ) | newIdx[key & 0x0fffffff];
      for( int k2 = k + 1; k2 < ncomp; k2++ ) for( auto& ih : itemH[k2] ) for( uint32_t& key : ih.prod ) if( (int) ( key >> 28 ) == k ) key = ( (uint32_t) k << 28 ) | newIdx[key & 0x0fffffff];
      // (chroma never is a producer for luma, and components are processed in ascending order, so every reference to component k is fixed here)
      // the per-CTU offsets follow the new order (units, hence blocks, stay grouped by CTU)
      {
        std::vector<uint32_t> cnt( (size_t) numCtu + 1, 0 );
        for( auto& ih : itemH[k] ) cnt[ih.ctu + 1]++;
        for( int a = 0; a < numCtu; a++ ) cnt[a + 1] += cnt[a];
        for( int a = 0; a <= numCtu; a++ ) ctuStartV[(size_t) k * ( numCtu + 1 ) + a] = cnt[a];
      }
    }
    // dependencies between units
    {
      std::vector<uint32_t> unitOfItem[3];
      for( int k = 0; k < ncomp; k++ ) unitOfItem[k].assign( intra[k].size(), 0 );
      for( size_t u = 0; u < units.size(); u++ ) for( uint32_t i = units[u].i0; i < units[u].i1; i++ ) unitOfItem[units[u].comp][i] = (uint32_t) u;
      for( size_t u = 0; u < units.size(); u++ )
      {
        UnitH& U = units[u];
        for( uint32_t i = U.i0; i < U.i1; i++ ) for( uint32_t key : itemH[U.comp][i].prod )
        {
          const uint32_t d = unitOfItem[key >> 28][key & 0x0fffffff];
          if( d != u && std::find( U.deps.begin(), U.deps.end(), d ) == U.deps.end() ) U.deps.push_back( d );
        }
      }
      // a unit lists at most VVR_INTRA_MAX_DEPS producers: longer lists are folded through empty join units
      for( size_t u = 0; u < units.size(); u++ )
        while( units[u].deps.size() > VVR_INTRA_MAX_DEPS )
        {
          UnitH j; j.comp = units[u].comp; j.ctu = units[u].ctu; j.i0 = j.i1 = j.iA = units[u].i0; j.bb.y0 = j.bb.y1 = 0; j.bb.c0 = j.bb.c1 = 1;
          j.deps.assign( units[u].deps.end() - VVR_INTRA_MAX_DEPS, units[u].deps.end() );
          units[u].deps.resize( units[u].deps.size() - VVR_INTRA_MAX_DEPS );
          units[u].deps.push_back( (uint32_t) units.size() );
          units.push_back( j );
        }
      // rank = length of the longest dependency chain below a unit (the unit graph is acyclic: luma never reads chroma, residual-add
      // units only read luma, other CTUs' units only earlier CTUs'); computed by relaxation in creation order until stable
      bool changed = true;
      for( size_t pass = 0; changed && pass <= units.size(); pass++ )      // (acyclic: stable after at most one pass per level; creation order makes it 2-3)
      {
        changed = false;
        for( auto& U : units ) for( uint32_t d : U.deps ) if( units[d].rank + 1 > U.rank ) { U.rank = units[d].rank + 1; changed = true; }
      }
    }
    return VVR_OK;
  }

  int groupUnits()
  {
    // ---- group the clusters of one (component, CTU) that sit at the same depth of the dependency graph into one unit: they cannot depend
    // on each other, a workgroup start costs more than a few small blocks, and waiting for the union of their producers delays nothing
    // that matters (all of them are less deep).  Residual-add units keep their own (HBM to HBM) workgroup.
    if( !getenv( "VVR_INTRA_NO_GROUPING" ) && !units.empty() )
    {
      std::vector<int32_t> target( units.size(), -1 );            // original unit -> group
      std::vector<std::vector<uint32_t>> parts;                    // groups: original units in creation order
      {
        std::vector<std::pair<uint64_t, uint32_t>> keyed;
        for( size_t u = 0; u < units.size(); u++ )
        {
          const bool own = units[u].iA == units[u].i1;             // residual-add unit (or empty): not grouped
          keyed.emplace_back( own ? ( ( (uint64_t) 1 << 63 ) | u ) : ( ( (uint64_t) units[u].comp << 56 ) | ( (uint64_t) units[u].ctu << 24 ) | (uint64_t) std::min( units[u].rank, 0xffffff ) ), (uint32_t) u );
        }
        std::stable_sort( keyed.begin(), keyed.end(), []( const std::pair<uint64_t, uint32_t>& x, const std::pair<uint64_t, uint32_t>& y ) { return x.first < y.first; } );
        static const uint32_t groupMax = getenv( "VVR_INTRA_GROUP_MAX" ) ? (uint32_t) atoi( getenv( "VVR_INTRA_GROUP_MAX" ) ) : 12;    // blocks per grouped unit
        for( size_t i = 0; i < keyed.size(); )
        {
          size_t j = i; parts.emplace_back();
          uint32_t blocks = 0;
          while( j < keyed.size() && keyed[j].first == keyed[i].first )
          {
            const uint32_t nb = units[keyed[j].second].i1 - units[keyed[j].second].i0;
            if( blocks && blocks + nb > groupMax ) break;                   // a serial workgroup should stay short: start another one
            blocks += nb;
            parts.back().push_back( keyed[j].second ); target[keyed[j].second] = (int32_t) parts.size() - 1; j++;
          }
          i = j;
        }
      }
      // groups in the order of their first original unit (keeps the blocks grouped by CTU)
      std::vector<uint32_t> orderM( parts.size() );
      for( size_t m = 0; m < parts.size(); m++ ) orderM[m] = (uint32_t) m;
      std::stable_sort( orderM.begin(), orderM.end(), [&]( uint32_t x, uint32_t y ) { return parts[x][0] < parts[y][0]; } );
      std::vector<IntraItem> newItems[3];
      std::vector<UnitH> merged;
      std::vector<uint32_t> newIndexOfGroup( parts.size(), 0 );
      for( uint32_t m : orderM )
      {
        const UnitH& f = units[parts[m][0]];
        UnitH U; U.comp = f.comp; U.ctu = f.ctu; U.i0 = (uint32_t) newItems[f.comp].size();
        for( uint32_t u : parts[m] )
        {
          const UnitH& o = units[u];
          newItems[o.comp].insert( newItems[o.comp].end(), intra[o.comp].begin() + o.i0, intra[o.comp].begin() + o.i1 );
          U.bb.y0 = std::min( U.bb.y0, o.bb.y0 ); U.bb.y1 = std::max( U.bb.y1, o.bb.y1 ); U.bb.c0 = std::min( U.bb.c0, o.bb.c0 ); U.bb.c1 = std::max( U.bb.c1, o.bb.c1 );
          U.hasCs = U.hasCs || o.hasCs;
        }
        U.i1 = (uint32_t) newItems[f.comp].size();
        U.iA = f.iA == f.i1 ? U.i1 : U.i0;
        newIndexOfGroup[m] = (uint32_t) merged.size();
        merged.push_back( U );
      }
      for( size_t m = 0; m < parts.size(); m++ )
      {
        UnitH& U = merged[newIndexOfGroup[m]];
        for( uint32_t u : parts[m] ) for( uint32_t d : units[u].deps )
        {
          const uint32_t nd = newIndexOfGroup[target[d]];
          if( nd != newIndexOfGroup[m] && std::find( U.deps.begin(), U.deps.end(), nd ) == U.deps.end() ) U.deps.push_back( nd );
        }
      }
      for( int k = 0; k < ncomp; k++ ) intra[k].swap( newItems[k] );
      units.swap( merged );
      for( size_t u = 0; u < units.size(); u++ )
        while( units[u].deps.size() > VVR_INTRA_MAX_DEPS )
        {
          UnitH j; j.comp = units[u].comp; j.ctu = units[u].ctu; j.i0 = j.i1 = j.iA = units[u].i0; j.bb.y0 = j.bb.y1 = 0; j.bb.c0 = j.bb.c1 = 1;
          j.deps.assign( units[u].deps.end() - VVR_INTRA_MAX_DEPS, units[u].deps.end() );
          units[u].deps.resize( units[u].deps.size() - VVR_INTRA_MAX_DEPS );
          units[u].deps.push_back( (uint32_t) units.size() );
          units.push_back( j );
        }
      for( auto& U : units ) U.rank = 0;
      bool changed = true;
      for( size_t pass = 0; changed && pass <= units.size(); pass++ )
      {
        changed = false;
        for( auto& U : units ) for( uint32_t d : U.deps ) if( units[d].rank + 1 > U.rank ) { U.rank = units[d].rank + 1; changed = true; }
      }
    }

    return VVR_OK;
  }

  int emitUnitTable()
  {
    // one item array for the three components; ctuStart holds offsets into it; active (component, CTU) pairs in raster order
    for( int k = 0; k < 3; k++ )
    {
      const uint32_t base = (uint32_t) intraAll.size();
      intraAll.insert( intraAll.end(), intra[k].begin(), intra[k].end() );
      for( int a = 0; a <= numCtu; a++ ) ctuStartV[(size_t) k * ( numCtu + 1 ) + a] += base;
    }
    // device unit table: units that wait for nothing first (they can never block a resident workgroup slot), then the others in
    // coding order; a unit only ever waits for units created before it, so every dependency holds a lower ticket
    {
      const uint32_t itemBase[3] = { ctuStartV[0], ctuStartV[(size_t) 1 * ( numCtu + 1 )], ctuStartV[(size_t) 2 * ( numCtu + 1 )] };
      std::vector<uint32_t> perm, inv( units.size() );
      // Long dependency chains (an intra picture: one CTU wavefront, ~60 levels): units that spin on their producers would hold most
      // workgroup slots (and their LDS) of the device for milliseconds while other pictures are in flight.  Such a picture runs its
      // intra stage as one launch per dependency level instead - units of one level never depend on each other, the launch boundary is
      // the synchronisation, nothing waits inside a kernel.  Short chains (isolated intra blocks of B pictures) keep the single
      // launch with flags, where a level barrier would cost more than the few waits.
      int maxRank = 0;
      for( auto& u : units ) maxRank = std::max( maxRank, u.rank );
      static const int levelThr = getenv( "VVR_INTRA_LEVEL_THR" ) ? atoi( getenv( "VVR_INTRA_LEVEL_THR" ) ) : 1 << 30;     // measured: slower (9.2 vs 8.0 ms for a 4K I picture, 1457 vs 1508 frames/s), off by default
      const bool byLevel = maxRank > levelThr;
      std::vector<std::pair<int, int>> levels;
      if( byLevel )
      {
        for( size_t t = 0; t < units.size(); t++ ) perm.push_back( (uint32_t) t );
        std::stable_sort( perm.begin(), perm.end(), [&]( uint32_t a, uint32_t b ) { return units[a].rank < units[b].rank; } );
        for( size_t t = 0; t < perm.size(); )
        {
          size_t e = t; while( e < perm.size() && units[perm[e]].rank == units[perm[t]].rank ) e++;
          levels.emplace_back( (int) t, (int) ( e - t ) );
          t = e;
        }
        for( auto& u : units ) u.deps.clear();                  // ordered by the launches
      }
      else
      {
      for( size_t t = 0; t < units.size(); t++ ) if( units[t].deps.empty() ) perm.push_back( (uint32_t) t );
      {
        // dependent units by depth of the dependency graph, then in WAVEFRONT order (key = ctuX + 2 * ctuY): every producer holds a
        // lower ticket, and the workgroups that are resident at any time are the ones on or near the current front
        std::vector<uint32_t> dep;
        for( size_t t = 0; t < units.size(); t++ ) if( !units[t].deps.empty() ) dep.push_back( (uint32_t) t );
        std::stable_sort( dep.begin(), dep.end(), [&]( uint32_t a, uint32_t b )
        {
          const int ka = (int) ( units[a].ctu % ctusX ) + 2 * (int) ( units[a].ctu / ctusX ), kb = (int) ( units[b].ctu % ctusX ) + 2 * (int) ( units[b].ctu / ctusX );
          // by depth first (every producer is less deep, hence holds a lower ticket; units of one depth start together, so few of
          // them find a producer that has not even started), then along the CTU wavefront.  Measured 7 % faster on B pictures than
          // wavefront-major order (VVR_INTRA_KEY_MAJOR), the same on intra pictures where depth and wavefront coincide.
          static const bool keyMajor = getenv( "VVR_INTRA_KEY_MAJOR" ) != nullptr;
          if( !keyMajor ) return units[a].rank != units[b].rank ? units[a].rank < units[b].rank : ka < kb;
          return ka != kb ? ka < kb : units[a].rank < units[b].rank;
        } );
        perm.insert( perm.end(), dep.begin(), dep.end() );
      }
      }
      intraLevelsV = levels;
      for( size_t t = 0; t < perm.size(); t++ ) inv[perm[t]] = (uint32_t) t;
      for( auto& u : units ) for( uint32_t d : u.deps ) units[d].waited = true;
      std::vector<uint32_t> unitCount( 3 * (size_t) numCtu, 0 );
      for( auto& u : units ) unitCount[(size_t) u.comp * numCtu + u.ctu]++;
      unitsDev.resize( units.size() );
      for( size_t t = 0; t < perm.size(); t++ )
      {
        const UnitH& u = units[perm[t]];
        IntraUnit& d = unitsDev[t]; memset( &d, 0, sizeof( d ) );
        // bit 31: the unit is the whole (component, CTU) and every sample of the CTU is intra, so the kernel only stages the reference
        // border and writes the CTU back with 16-byte stores
        bool all = unitCount[(size_t) u.comp * numCtu + u.ctu] == 1;
        {
          const int ctu4 = 1 << ( h.log2_ctu - 2 ), ux = (int) ( u.ctu % ctusX ) * ctu4, uy = (int) ( u.ctu / ctusX ) * ctu4;
          for( int y = uy; y < std::min( uy + ctu4, h4 ) && all; y++ ) for( int x = ux; x < std::min( ux + ctu4, w4 ); x++ ) if( intraAt[(size_t) y * w4 + x] != 1 ) { all = false; break; }
        }
        d.ent = ( u.comp << 24 ) | u.ctu | ( u.hasCs ? 0x20000000u : 0 ) | ( u.waited ? 0x40000000u : 0 ) | ( all ? 0x80000000u : 0 );
        d.i0 = itemBase[u.comp] + u.i0; d.i1 = itemBase[u.comp] + u.i1; d.iA = itemBase[u.comp] + u.iA;
        d.bbox = (uint32_t) u.bb.y0 | ( (uint32_t) u.bb.y1 << 8 ) | ( (uint32_t) u.bb.c0 << 16 ) | ( (uint32_t) u.bb.c1 << 24 );
        d.ndeps = (uint32_t) std::min<size_t>( u.deps.size(), VVR_INTRA_MAX_DEPS );
        if( u.deps.size() > VVR_INTRA_MAX_DEPS ) { c->setError( "internal: intra unit with too many dependencies" ); return VVR_ERR_UNSPECIFIED; }
        for( uint32_t k = 0; k < d.ndeps; k++ ) d.deps[k] = inv[u.deps[k]];
      }
    }
    if( getenv( "VVR_INTRA_STATS" ) )
    {
      size_t nIndep = 0, nBulk = 0, nItems = 0, nResiAdd = 0; int maxRank = 0; size_t perComp[3] = { 0, 0, 0 }, big = 0, maxBlocks = 0;
      for( auto& u : units ) if( u.iA != u.i1 ) maxBlocks = std::max<size_t>( maxBlocks, u.i1 - u.i0 );
      fprintf( stderr, "[vvr] largest serial unit: %zu blocks\n", maxBlocks );
      for( auto& u : units ) { nIndep += u.deps.empty(); nBulk += u.iA == u.i1 && u.i1 > u.i0; maxRank = std::max( maxRank, u.rank ); perComp[u.comp]++; nItems += u.i1 - u.i0; if( u.iA == u.i1 ) nResiAdd += u.i1 - u.i0; big += ( u.i1 - u.i0 ) > 8; }
      fprintf( stderr, "[vvr] POC %d: %zu intra units (Y %zu Cb %zu Cr %zu), %zu independent, %zu residual-add units, %zu blocks (%zu residual-add), %zu units > 8 blocks, longest chain %d\n",
               h.poc, units.size(), perComp[0], perComp[1], perComp[2], nIndep, nBulk, nItems, nResiAdd, big, maxRank );
    }
    return VVR_OK;
  }

  int upload( vvr_prepared** out )
  {
    const double samples = (double) h.width * h.height * ( ncomp == 3 ? 1.5 : 1.0 );
    bytes[K_DEBLOCK_V] = bytes[K_DEBLOCK_H] = samples * 4 + (double) w4 * h4 * sizeof( vvr_lfp );
    bytes[K_SAO] = samples * 4; bytes[K_ALF] = samples * 4; bytes[K_COPY] = samples * 4;

    // ---- one device allocation for everything
    struct Part { const void* src; size_t n; size_t off; };
    std::vector<Part> parts;
    size_t total = 0;
    auto add = [&]( const void* src, size_t n ) { Part q{ src, n, total }; parts.push_back( q ); total += alignUp( std::max<size_t>( n, 16 ), 256 ); return (int) parts.size() - 1; };
    const int iCu = add( p->cu, sizeof( vvr_cu ) * p->num_cu );
    const int iTu = add( p->tu, sizeof( vvr_tu ) * p->num_tu );
    const int iCoef = add( p->coef, sizeof( int16_t ) * (size_t) p->num_coef );
    const int iMot = p->motion ? add( p->motion, sizeof( vvr_motion ) * (size_t) w4 * h4 ) : -1;
    const int iL0 = add( p->lfp[0], sizeof( vvr_lfp ) * (size_t) w4 * h4 );
    const int iL1 = add( p->lfp[1], sizeof( vvr_lfp ) * (size_t) w4 * h4 );
    const int iSao = p->sao ? add( p->sao, sizeof( vvr_sao_ctu ) * numCtu ) : -1;
    const int iAlf = p->alf ? add( p->alf, sizeof( vvr_alf_ctu ) * numCtu ) : -1;
    const int iAlfP = p->alf_params ? add( p->alf_params, sizeof( vvr_alf_params ) ) : -1;
    const bool lmcs = ( h.tool_flags & VVR_TOOL_LMCS ) != 0;
    std::vector<uint8_t> interAtV;
    if( lmcs )
    {
      interAtV.assign( (size_t) w4 * h4 + 8, 0 );
      for( uint32_t i = 0; i < p->num_cu; i++ )
      {
        const vvr_cu& cu = p->cu[i];
        if( cu.pred_mode != VVR_PRED_INTER ) continue;
        for( int y = cu.y; y < cu.y + cu.h; y += 4 ) for( int x = cu.x; x < cu.x + cu.w; x += 4 ) interAtV[(size_t) ( y >> 2 ) * w4 + ( x >> 2 )] = 1;
      }
    }
    const int iLmcs = lmcs ? add( p->lmcs, sizeof( vvr_lmcs_params ) ) : -1;
    const int iSl = ( h.tool_flags & VVR_TOOL_SCALING_LIST ) ? add( p->scaling, sizeof( vvr_scaling_list ) ) : -1;
    const int iWp = ( ( h.tool_flags & VVR_TOOL_WP ) && h.slice_type != 2 ) ? add( p->wp, sizeof( vvr_wp_params ) ) : -1;
    const int iInterAt = lmcs ? add( interAtV.data(), interAtV.size() ) : -1;
    const int iCsVpdu = cscale ? add( csVpduV.data(), sizeof( uint32_t ) * csVpduV.size() ) : -1;
    if( lmcs ) { bytes[K_LMCS] = ( samples / ( ncomp == 3 ? 1.5 : 1.0 ) ) * 4 * 2; }     // forward pass over the inter luma (upper bound) + inverse pass over all luma
    const int iMc = add( mc.data(), sizeof( McItem ) * mc.size() );
    const int iMcB = add( mcBdof.data(), sizeof( McItem ) * mcBdof.size() );
    const int iMcD = add( mcDmvr.data(), sizeof( McItem ) * mcDmvr.size() );
    const int iMcA = add( mcAff.data(), sizeof( McItem ) * mcAff.size() );
    const int iDmvrOut = add( nullptr, sizeof( int32_t ) * 2 * (size_t) numDmvr );
    int iTb[3]; for( int k = 0; k < 3; k++ ) iTb[k] = add( tb[k].data(), sizeof( TbItem ) * tb[k].size() );
    const int iIntra = add( intraAll.data(), sizeof( IntraItem ) * intraAll.size() );
    const int iCtuStart = add( ctuStartV.data(), sizeof( uint32_t ) * ctuStartV.size() );
    const int iActive = add( unitsDev.data(), sizeof( IntraUnit ) * unitsDev.size() );

    vvr_prepared* q = new vvr_prepared();
    q->hdr = h;
    if( hipMalloc( &q->blob.p, total ) != hipSuccess ) { delete q; c->setError( "hipMalloc failed" ); return VVR_ERR_DEVICE; }
    q->blob.n = total;
    // stage through one pinned buffer -> a single H2D copy
    char* staging = nullptr;
    if( hipHostMalloc( (void**) &staging, total, hipHostMallocDefault ) != hipSuccess ) { hipFree( q->blob.p ); delete q; c->setError( "hipHostMalloc failed" ); return VVR_ERR_DEVICE; }
    for( auto& pt : parts ) if( pt.n ) { if( pt.src ) memcpy( staging + pt.off, pt.src, pt.n ); else memset( staging + pt.off, 0, pt.n ); }
    hipError_t e = hipMemcpy( q->blob.p, staging, total, hipMemcpyHostToDevice );
    hipHostFree( staging );
    if( e != hipSuccess ) { hipFree( q->blob.p ); delete q; c->setError( "H2D copy failed" ); return VVR_ERR_DEVICE; }
    char* base = (char*) q->blob.p;
    PicDev& d = q->pic; memset( &d, 0, sizeof( d ) );
    d.hdr = h; d.w4 = w4; d.h4 = h4; d.ctus_x = ctusX; d.ctus_y = ctusY;
    d.cu = (const vvr_cu*) ( base + parts[iCu].off ); d.tu = (const vvr_tu*) ( base + parts[iTu].off ); d.coef = (const int16_t*) ( base + parts[iCoef].off );
    d.motion = iMot >= 0 ? (const vvr_motion*) ( base + parts[iMot].off ) : nullptr;
    d.lfp[0] = (const vvr_lfp*) ( base + parts[iL0].off ); d.lfp[1] = (const vvr_lfp*) ( base + parts[iL1].off );
    d.sao = iSao >= 0 ? (const vvr_sao_ctu*) ( base + parts[iSao].off ) : nullptr;
    d.alf = iAlf >= 0 ? (const vvr_alf_ctu*) ( base + parts[iAlf].off ) : nullptr;
    d.alf_params = iAlfP >= 0 ? (const vvr_alf_params*) ( base + parts[iAlfP].off ) : nullptr;
    d.lmcs = iLmcs >= 0 ? (const vvr_lmcs_params*) ( base + parts[iLmcs].off ) : nullptr;
    d.scaling = iSl >= 0 ? (const vvr_scaling_list*) ( base + parts[iSl].off ) : nullptr;
    d.wp = iWp >= 0 ? (const vvr_wp_params*) ( base + parts[iWp].off ) : nullptr;
    d.interAt = iInterAt >= 0 ? (const uint8_t*) ( base + parts[iInterAt].off ) : nullptr;
    d.csVpdu = iCsVpdu >= 0 ? (const uint32_t*) ( base + parts[iCsVpdu].off ) : nullptr; d.vpdusX = vpdusX; d.vpduLog2 = vpduLog2;
    q->mcItems = (McItem*) ( base + parts[iMc].off ); q->numMc = (int) mc.size();
    q->bdofItems = (McItem*) ( base + parts[iMcB].off ); q->numBdofItems = (int) mcBdof.size();
    q->dmvrItems = (McItem*) ( base + parts[iMcD].off ); q->numDmvrItems = (int) mcDmvr.size();
    q->affItems = (McItem*) ( base + parts[iMcA].off ); q->numAffItems = (int) mcAff.size();
    q->dmvrOut = (int32_t*) ( base + parts[iDmvrOut].off ); q->numDmvr = numDmvr;
    for( int k = 0; k < 3; k++ ) { q->tbItems[k] = (TbItem*) ( base + parts[iTb[k]].off ); q->numTb[k] = (int) tb[k].size(); }
    q->intraItems = (IntraItem*) ( base + parts[iIntra].off ); q->numIntra = (int) intraAll.size();
    q->ctuStart = (uint32_t*) ( base + parts[iCtuStart].off );
    q->units = (IntraUnit*) ( base + parts[iActive].off ); q->numActive = (int) unitsDev.size();
    q->intraLevels = intraLevelsV;
    memcpy( q->bytes, bytes, sizeof( bytes ) );
    *out = q;
    return VVR_OK;
  }
};
}   // namespace

VVR_API int vvr_prepare( vvr_context* c, const vvr_picture* p, vvr_prepared** out )
{
  if( !c || !p || !out ) return VVR_ERR_PARAMETER;
  if( p->resident ) { c->setError( "vvr_prepare needs host arrays" ); return VVR_ERR_PARAMETER; }
  int rc = validate( c, p );
  if( rc != VVR_OK ) return rc;
  hipSetDevice( c->device );
  PicturePreparer P( c, p );
  if( ( rc = P.mapDecodingOrder() ) != VVR_OK || ( rc = P.buildWorkLists() ) != VVR_OK || ( rc = P.formUnits() ) != VVR_OK || ( rc = P.groupUnits() ) != VVR_OK
   || ( rc = P.emitUnitTable() ) != VVR_OK ) return rc;
  return P.upload( out );
}

// ---------------------------------------------------------------------------------------------------------------------
// enqueue one prepared picture
// ---------------------------------------------------------------------------------------------------------------------
static vvr_context::Job* findJob( vvr_context* c, int id ) { for( auto& j : c->jobs ) if( j.id == id ) return &j; return nullptr; }

static int finishJob( vvr_context* c, vvr_context::Job& j )
{
  if( j.waited ) return VVR_OK;
  HIPCHK( c, hipEventSynchronize( j.done ) );
  for( auto& t : j.timings )
  {
    float ms = 0; hipEventElapsedTime( &ms, t.a, t.b );
    c->stats[t.kernel].launches++; c->stats[t.kernel].ms += ms; c->stats[t.kernel].bytes += t.bytes;
    hipEventDestroy( t.a ); hipEventDestroy( t.b );
  }
  j.timings.clear();
  if( j.prepared && j.prepared->numDmvr )
  {
    // refined MVs feed the temporal MV prediction of later pictures on the host (DecCu::TaskFinishMotionInfo, DecCu.cpp:161)
    j.dmvr.resize( 2 * (size_t) j.prepared->numDmvr );
    HIPCHK( c, hipMemcpy( j.dmvr.data(), j.prepared->dmvrOut, sizeof( int32_t ) * j.dmvr.size(), hipMemcpyDeviceToHost ) );
  }
  j.prepared = nullptr;
  if( j.autoFree ) { vvr_free_prepared( c, j.autoFree ); j.autoFree = nullptr; }
  j.waited = true;
  return VVR_OK;
}

VVR_API int vvr_submit_prepared( vvr_context* c, vvr_prepared* q )
{
  if( !c || !q ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  const vvr_pic_header& h = q->hdr;
  const int lane = c->nextStream; c->nextStream = ( c->nextStream + 1 ) % (int) c->streams.size();
  hipStream_t s = c->streams[lane];
  // a lane's scratch planes are reused: the previous job of this lane is ordered before us by the stream itself
  // ---- dependencies: every job that read or wrote one of our slots
  auto waitUsers = [&]( int slot ) { for( int id : c->slotUsers[slot] ) { vvr_context::Job* j = findJob( c, id ); if( j && !j->waited && j->stream != lane ) hipStreamWaitEvent( s, j->done, 0 ); } };
  waitUsers( h.out_slot );
  RefSet refs; memset( &refs, 0, sizeof( refs ) );
  if( h.slice_type != 2 )
    for( int l = 0; l < 2; l++ ) for( int i = 0; i < h.num_ref[l]; i++ )
    {
      const int slot = h.ref_slot[l][i];
      // wait for the writer of the reference (it is the first entry since the slot was last written)
      if( !c->slotUsers[slot].empty() ) { vvr_context::Job* j = findJob( c, c->slotUsers[slot][0] ); if( j && !j->waited && j->stream != lane ) hipStreamWaitEvent( s, j->done, 0 ); }
      for( int k = 0; k < 3; k++ ) refs.p[l * VVR_MAX_REFS + i][k] = c->slots[slot].p[k];
    }
  vvr_context::Job job; job.id = c->nextJob++; job.stream = lane; job.waited = false; job.autoFree = nullptr;
  HIPCHK( c, hipEventCreateWithFlags( &job.done, c->statsOn ? hipEventDefault : hipEventDisableTiming ) );

  DevPlanes A = c->slots[h.out_slot], B = c->scratchB[lane], R = c->scratchR[lane];
  auto timed = [&]( int k, auto&& fn )
  {
    if( c->statsOn ) { PendingTiming t; hipEventCreate( &t.a ); hipEventCreate( &t.b ); t.kernel = k; t.bytes = q->bytes[k]; hipEventRecord( t.a, s ); fn(); hipEventRecord( t.b, s ); job.timings.push_back( t ); }
    else fn();
  };
  // INTER stage: prediction of every inter CU, then residual add (DecLibRecon.cpp:831-874)
  if( q->numMc + q->numBdofItems ) timed( K_MC, [&]{ launch_mc( s, q->pic, refs, A, q->mcItems, q->numMc, 0 ); launch_mc( s, q->pic, refs, A, q->bdofItems, q->numBdofItems, 1 ); } );
  if( q->numDmvrItems ) timed( K_MC_DMVR, [&]{ launch_mc_dmvr( s, q->pic, refs, A, q->dmvrItems, q->numDmvrItems, q->dmvrOut ); } );
  if( q->numAffItems ) timed( K_MC_AFFINE, [&]{ launch_mc_affine( s, q->pic, refs, A, q->affItems, q->numAffItems ); } );
  const bool lmcsOn = ( h.tool_flags & VVR_TOOL_LMCS ) != 0;
  // LMCS: the inter prediction is forward-mapped before any residual is added (DecCu.cpp:458-476); I pictures have no inter prediction
  if( lmcsOn && h.slice_type != 2 && ( q->numMc + q->numBdofItems + q->numDmvrItems + q->numAffItems ) ) timed( K_LMCS, [&]{ launch_lmcs( s, q->pic, A, 0 ); } );
  job.prepared = q;
  if( q->numTb[0] + q->numTb[1] + q->numTb[2] )
    timed( K_ITRANS, [&]{ for( int k = 0; k < 3; k++ ) launch_itrans( s, q->pic, A, R, q->tbItems[k], q->numTb[k], 16 << k ); } );
  // INTRA stage: wavefront over the CTUs that contain intra blocks (DecLibRecon.cpp:876-911)
  {
    // ticket + one flag per unit (or one counter per level): a picture with more units than the lane's buffer holds gets a larger one;
    // work queued on the lane may still use the old buffer, so the lane is drained first (rare: see vvr_create)
    const size_t need = 1 + std::max<size_t>( (size_t) q->numActive, q->intraLevels.size() );
    if( need > c->syncCap[lane] )
    {
      HIPCHK( c, hipStreamSynchronize( s ) );
      int* p = nullptr;
      HIPCHK( c, hipMalloc( (void**) &p, sizeof( int ) * need * 2 ) );
      hipFree( c->syncBuf[lane] );
      c->syncBuf[lane] = p; c->syncCap[lane] = need * 2;
    }
  }
  if( q->numActive ) timed( K_INTRA, [&]
  {
    if( q->intraLevels.empty() ) launch_intra( s, q->pic, A, R, q->intraItems, q->units, q->numActive, c->syncBuf[lane] );
    else launch_intra_levels( s, q->pic, A, R, q->intraItems, q->units, q->intraLevels.data(), (int) q->intraLevels.size(), c->syncBuf[lane] );
  } );
  // LMCS: inverse luma mapping of the reconstructed picture (RSP state, DecLibRecon.cpp:935)
  if( lmcsOn ) timed( K_LMCS, [&]{ launch_lmcs( s, q->pic, A, 1 ); } );
  // in-loop filters: LF_V, LF_H, SAO, ALF (DecLibRecon.cpp:943-1100)
  // debugging aid (like the reference's per-stage CRC traces, LoopFilter.cpp:399-406): VVR_STOP_AFTER=reco|dbk|sao
  const char* stopEnv = getenv( "VVR_STOP_AFTER" );
  const int stopAfter = !stopEnv ? 0 : !strcmp( stopEnv, "reco" ) ? 1 : !strcmp( stopEnv, "dbk" ) ? 2 : !strcmp( stopEnv, "sao" ) ? 3 : 0;
  if( !( h.tool_flags & VVR_TOOL_DEBLOCK_OFF ) && stopAfter != 1 )
  {
    timed( K_DEBLOCK_V, [&]{ launch_deblock( s, q->pic, A, 0 ); } );
    timed( K_DEBLOCK_H, [&]{ launch_deblock( s, q->pic, A, 1 ); } );
  }
  const bool sao = ( h.tool_flags & ( VVR_TOOL_SAO_LUMA | VVR_TOOL_SAO_CHROMA ) ) != 0 && stopAfter != 1 && stopAfter != 2;
  const bool alf = ( h.tool_flags & VVR_TOOL_ALF ) != 0 && stopAfter == 0;
  if( sao && alf ) { timed( K_SAO, [&]{ launch_sao( s, q->pic, A, B ); } ); timed( K_ALF, [&]{ launch_alf( s, q->pic, B, A ); } ); }
  else if( sao )   { timed( K_SAO, [&]{ launch_sao( s, q->pic, A, B ); } ); timed( K_COPY, [&]{ launch_copy_planes( s, B, A ); } ); }
  else if( alf )   { timed( K_COPY, [&]{ launch_copy_planes( s, A, B ); } ); timed( K_ALF, [&]{ launch_alf( s, q->pic, B, A ); } ); }
  HIPCHK( c, hipGetLastError() );
  HIPCHK( c, hipEventRecord( job.done, s ) );
  // bookkeeping
  c->slotUsers[h.out_slot].clear(); c->slotUsers[h.out_slot].push_back( job.id );
  if( h.slice_type != 2 ) for( int l = 0; l < 2; l++ ) for( int i = 0; i < h.num_ref[l]; i++ ) c->slotUsers[h.ref_slot[l][i]].push_back( job.id );
  // retire old finished jobs to keep the table small
  if( c->jobs.size() > 256 )
  {
    std::vector<vvr_context::Job> keep;
    for( auto& j : c->jobs ) { if( j.waited ) { hipEventDestroy( j.done ); } else keep.push_back( j ); }
    c->jobs.swap( keep );
  }
  c->jobs.push_back( job );
  return job.id;
}

VVR_API int vvr_submit( vvr_context* c, const vvr_picture* p )
{
  vvr_prepared* q = nullptr;
  int rc = vvr_prepare( c, p, &q );
  if( rc != VVR_OK ) return rc;
  const int id = vvr_submit_prepared( c, q );
  if( id < 0 ) { vvr_free_prepared( c, q ); return id; }
  findJob( c, id )->autoFree = q;
  return id;
}

VVR_API int vvr_wait( vvr_context* c, int job )
{
  if( !c ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  vvr_context::Job* j = findJob( c, job );
  if( !j ) return VVR_OK;       // already retired
  return finishJob( c, *j );
}

VVR_API int vvr_sync( vvr_context* c )
{
  if( !c ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  for( auto& j : c->jobs ) { int rc = finishJob( c, j ); if( rc != VVR_OK ) return rc; }
  for( auto s : c->streams ) HIPCHK( c, hipStreamSynchronize( s ) );
  return VVR_OK;
}

VVR_API void* vvr_job_stream( vvr_context* c, int job ) { vvr_context::Job* j = c ? findJob( c, job ) : nullptr; return j ? (void*) c->streams[j->stream] : nullptr; }

VVR_API int vvr_read_dmvr( vvr_context* c, int job, int32_t* dst, size_t numEntries )
{
  if( !c || ( !dst && numEntries ) ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  vvr_context::Job* j = findJob( c, job );
  if( !j ) { c->setError( "vvr_read_dmvr: job already retired" ); return VVR_ERR_PARAMETER; }
  const int rc = finishJob( c, *j );
  if( rc != VVR_OK ) return rc;
  const size_t n = std::min( numEntries, j->dmvr.size() / 2 );
  if( n ) memcpy( dst, j->dmvr.data(), sizeof( int32_t ) * 2 * n );
  return (int) ( j->dmvr.size() / 2 );
}

VVR_API int vvr_enable_stats( vvr_context* c, int on ) { if( !c ) return VVR_ERR_PARAMETER; vvr_sync( c ); c->statsOn = on != 0; for( auto& s : c->stats ) s = Stat(); return VVR_OK; }

VVR_API int vvr_get_stats( vvr_context* c, vvr_kernel_stat* out, int maxEntries )
{
  if( !c ) return VVR_ERR_PARAMETER;
  vvr_sync( c );
  int n = 0;
  for( int k = 0; k < K_NUM && n < maxEntries; k++ )
  {
    if( !c->stats[k].launches ) continue;
    memset( &out[n], 0, sizeof( out[n] ) );
    snprintf( out[n].name, sizeof( out[n].name ), "%s", kKernelNames[k] );
    out[n].launches = c->stats[k].launches; out[n].total_ms = c->stats[k].ms; out[n].algo_bytes = c->stats[k].bytes;
    n++;
  }
  return n;
}

VVR_API uint8_t vvr_resolve_tr_type( const vvr_pic_header*, const vvr_cu* cu, const vvr_tu* tu, int comp, int implicit_mts, int explicit_intra, int explicit_inter )
{
  // TrQuant::getTrTypes (TrQuant.cpp:330-407).  0 DCT2, 1 DCT8, 2 DST7; returns (ver << 2) | hor
  int hor = 0, ver = 0;
  const bool intra = cu->pred_mode == VVR_PRED_INTRA, luma = comp == 0;
  const bool isImplicit = intra && luma && implicit_mts && cu->lfnst_idx == 0 && !( cu->flags & VVR_CU_MIP );
  const bool isISP = intra && luma && cu->isp_mode;
  if( isISP && cu->lfnst_idx ) return 0;
  const int lw = tu->w, lh = tu->h;
  if( isImplicit || isISP )
  {
    if( lw >= 4 && lw <= 16 ) hor = 2;
    if( lh >= 4 && lh <= 16 ) ver = 2;
    return (uint8_t) ( ( ver << 2 ) | hor );
  }
  const bool isInterLuma = cu->pred_mode == VVR_PRED_INTER && luma;
  const bool isExplicit = intra ? ( explicit_intra && luma ) : ( explicit_inter && isInterLuma );
  if( isInterLuma && cu->sbt_info )
  {
    const int sbtIdx = cu->sbt_info & 0xf, sbtPos = ( cu->sbt_info >> 4 ) & 0x3;      // CU::getSbtIdx / getSbtPos
    if( sbtIdx == 1 || sbtIdx == 3 )   // SBT_VER_HALF, SBT_VER_QUAD
    { if( lh > 32 ) hor = ver = 0; else if( sbtPos == 0 ) { hor = 1; ver = 2; } else { hor = 2; ver = 2; } }
    else
    { if( lw > 32 ) hor = ver = 0; else if( sbtPos == 0 ) { hor = 2; ver = 1; } else { hor = 2; ver = 2; } }
    return (uint8_t) ( ( ver << 2 ) | hor );
  }
  if( isExplicit && tu->mts_idx[comp] > VVR_MTS_SKIP )
  {
    hor = ( ( tu->mts_idx[comp] - 2 ) & 1 ) ? 1 : 2;
    ver = ( ( tu->mts_idx[comp] - 2 ) >> 1 ) ? 1 : 2;
  }
  return (uint8_t) ( ( ver << 2 ) | hor );
}

}   // extern "C"
