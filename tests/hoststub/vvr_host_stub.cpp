// TEST INFRASTRUCTURE (tests/ only): the product's host glue (vvdec_amd/csrc/vvr_api.cpp: validation, work lists, intra-stage units and
// their dependency graph, job bookkeeping) compiled for the CPU against a stand-in HIP runtime, so that the part of the scheduler that
// lives on the host can be checked without a GPU.  "Device" memory is host memory, streams and events are inert, kernel launches do
// nothing: no sample is ever computed here.  The product library never contains any of this.
#include <hip/hip_runtime_api.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <vector>

// trace of the stream / event operations the host glue issues: { op (0 wait, 1 record), stream number, event number }, numbers in creation order
static std::vector<int> g_trace;
struct StubStream { int id; };
struct StubEvent { int id; int dead; };      // (events are never given back to the allocator: a use after hipEventDestroy is counted, not undefined)
static int g_deadEventUses = 0;
static StubEvent* live( hipEvent_t e ) { StubEvent* p = (StubEvent*) e; if( p->dead ) g_deadEventUses++; return p; }
static int g_numStreams = 0, g_numEvents = 0;

extern "C" {
hipError_t hipGetDeviceCount( int* n ) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice( int ) { return hipSuccess; }
hipError_t hipDeviceGetPCIBusId( char* b, int n, int ) { if( n > 0 ) b[0] = 0; return hipErrorInvalidDevice; }      // (no device: host threads are not pinned)
hipError_t hipGetDeviceProperties( hipDeviceProp_t* p, int ) { memset( p, 0, sizeof( *p ) ); strcpy( p->gcnArchName, "gfx950:host-stub" ); p->multiProcessorCount = 256; return hipSuccess; }
hipError_t hipMalloc( void** p, size_t n ) { *p = calloc( 1, n ? n : 1 ); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree( void* p ) { free( p ); return hipSuccess; }
hipError_t hipHostMalloc( void** p, size_t n, unsigned int ) { *p = calloc( 1, n ? n : 1 ); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree( void* p ) { free( p ); return hipSuccess; }
hipError_t hipMemcpy( void* d, const void* s, size_t n, hipMemcpyKind ) { memcpy( d, s, n ); return hipSuccess; }
hipError_t hipMemcpy2DAsync( void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t ) { for( size_t y = 0; y < h; y++ ) memcpy( (char*) d + y * dp, (const char*) s + y * sp, w ); return hipSuccess; }
hipError_t hipMemcpy2D( void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind ) { for( size_t y = 0; y < h; y++ ) memcpy( (char*) d + y * dp, (const char*) s + y * sp, w ); return hipSuccess; }
hipError_t hipMemset( void* d, int v, size_t n ) { memset( d, v, n ); return hipSuccess; }
hipError_t hipStreamCreateWithFlags( hipStream_t* s, unsigned int ) { StubStream* p = (StubStream*) calloc( 1, sizeof( StubStream ) ); p->id = g_numStreams++; *s = (hipStream_t) p; return hipSuccess; }
hipError_t hipStreamCreateWithPriority( hipStream_t* s, unsigned int f, int ) { return hipStreamCreateWithFlags( s, f ); }
hipError_t hipDeviceGetStreamPriorityRange( int* least, int* greatest ) { *least = 0; *greatest = -1; return hipSuccess; }
static int g_eventsPending = 0;                                      // tests: pretend nothing enqueued has finished yet
hipError_t hipEventQuery( hipEvent_t e ) { live( e ); return g_eventsPending ? hipErrorNotReady : hipSuccess; }       // (by default the stand-in device has finished everything it was given)
hipError_t hipStreamDestroy( hipStream_t s ) { free( s ); return hipSuccess; }
hipError_t hipStreamSynchronize( hipStream_t ) { return hipSuccess; }
hipError_t hipStreamWaitEvent( hipStream_t s, hipEvent_t e, unsigned int ) { g_trace.push_back( 0 ); g_trace.push_back( ( (StubStream*) s )->id ); g_trace.push_back( live( e )->id ); return hipSuccess; }
hipError_t hipEventCreate( hipEvent_t* e ) { StubEvent* p = (StubEvent*) calloc( 1, sizeof( StubEvent ) ); p->id = g_numEvents++; *e = (hipEvent_t) p; return hipSuccess; }
hipError_t hipEventCreateWithFlags( hipEvent_t* e, unsigned ) { return hipEventCreate( e ); }
hipError_t hipEventDestroy( hipEvent_t e ) { live( e )->dead = 1; return hipSuccess; }
hipError_t hipEventRecord( hipEvent_t e, hipStream_t s ) { g_trace.push_back( 1 ); g_trace.push_back( ( (StubStream*) s )->id ); g_trace.push_back( live( e )->id ); return hipSuccess; }
static int g_failLeafWaits = 0;      // vvt_fail_leaf_waits: the next n launches of the intra stage of inter pictures report a wait that gave up (the job's error word)
static int g_delayUs = 0;      // vvt_set_delay: calls that block on a real device take this long here (stress tests of the host pipeline)
hipError_t hipEventSynchronize( hipEvent_t e ) { live( e ); if( g_delayUs ) usleep( g_delayUs ); return hipSuccess; }
hipError_t hipEventElapsedTime( float* ms, hipEvent_t, hipEvent_t ) { *ms = 0.f; return hipSuccess; }
hipError_t hipGetLastError( void ) { return hipSuccess; }
const char* hipGetErrorString( hipError_t ) { return "host stub"; }
static size_t g_h2dCopies = 0, g_h2dBytes = 0; static uint64_t g_h2dHash = 1469598103934665603ull;      // (FNV-1a over everything copied to the device, in copy order)
hipError_t hipMemcpyAsync( void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t ) { memcpy( d, s, n ); if( k == hipMemcpyHostToDevice ) { g_h2dCopies++; g_h2dBytes += n; const uint64_t* w = (const uint64_t*) s; uint64_t hsh = g_h2dHash; for( size_t i = 0; i < n / 8; i++ ) hsh = ( hsh ^ w[i] ) * 1099511628211ull; g_h2dHash = hsh; } return hipSuccess; }
hipError_t hipMemsetAsync( void* d, int v, size_t n, hipStream_t ) { memset( d, v, n ); return hipSuccess; }
}

#define VVT_SLOW_I_PICTURES
#include <chrono>
#include "../../vvdec_amd/csrc/vvr_prepare.cpp"
#include "../../vvdec_amd/csrc/vvr_api.cpp"

// kernel launches: nothing to run on the host; the launch of the intra stage records what it was handed
static int g_lastIntraUnits = -1; static const int* g_lastSync = nullptr;
int  vvr_upload_tables() { return 0; }
void launch_prep( hipStream_t, const PicDev&, const PrepWork& ) {}
void launch_mc( hipStream_t, const PicDev&, const RefSet&, DevPlanes, const McItem*, int, const McItem*, int, int ) { if( g_delayUs ) usleep( g_delayUs ); }
void launch_itrans( hipStream_t, const PicDev&, DevPlanes, DevPlanes, const TbItem*, int, int ) {}
// The one launch that leaves a trace in the "picture": the vertical deblocking pass stamps the first four luma samples of the output slot with
// a hash of the picture's POC and of the stamps found in its reference slots at that moment.  Launches run at submission time here, so
// the stamp of a picture is right exactly if every reference slot held the right picture when it was submitted - which is what the tests
// of the multi-rank scheduler (broadcast of reference pictures between ranks, tests/test_multi_gpu_gloo.py) need to see.
static void stamp_picture( const PicDev& pic, DevPlanes reco )
{
  const vvr_pic_header& h = pic.hdr;
  size_t slotBytes = 0;
  for( int c = 0; c < 3; c++ ) if( reco.p[c] ) slotBytes += ( (size_t) reco.stride[c] * reco.h[c] * sizeof( pel_t ) + 255 ) / 256 * 256;
  uint64_t x = 1469598103934665603ull ^ (uint64_t) (uint32_t) h.poc;
  auto mix = [&]( uint64_t v ) { x = ( x ^ v ) * 1099511628211ull; };
  mix( 0x9e37 );
  if( h.slice_type != 2 )
    for( int l = 0; l < 2; l++ ) for( int i = 0; i < h.num_ref[l]; i++ )
    {
      const pel_t* r = (const pel_t*) ( (const char*) reco.p[0] + ( (ptrdiff_t) h.ref_slot[l][i] - h.out_slot ) * (ptrdiff_t) slotBytes );
      uint64_t v; memcpy( &v, r, 8 ); mix( v ); mix( (uint64_t) h.ref_poc[l][i] );
    }
  for( int k = 0; k < 4; k++ ) reco.p[0][k] = (pel_t) ( ( x >> ( 16 * k ) ) & 0x3ff );
}
// (the stamp needs the DPB slot: a picture with SAO or ALF is reconstructed in the lane's scratch picture and reaches its slot with the fused
// SAO + ALF pass, every other picture is deblocked in its slot)
static bool reaches_slot_with_sao_alf( const PicDev& pic ) { return ( pic.hdr.tool_flags & ( VVR_TOOL_SAO_LUMA | VVR_TOOL_SAO_CHROMA | VVR_TOOL_ALF ) ) != 0; }
void launch_deblock( hipStream_t, const PicDev& pic, DevPlanes reco, int dir ) { if( dir == 0 && !reaches_slot_with_sao_alf( pic ) ) stamp_picture( pic, reco ); }
void launch_deblock_tile( hipStream_t, const PicDev&, DevPlanes, DevPlanes, int, bool ) {}
// LF_INIT has a functional stand-in: the kernels are thin loops around vvr_lf_init.h, which compiles for the host as it is - so the whole path (the
// host's list of sub-block motion, the layout of the device-written parts, the derivation itself) is checked against the reference's tables without a GPU
static vvr_lfp* g_lastLfp[2] = { nullptr, nullptr }; static int g_lastLfpCells = 0;
void launch_lf_init( hipStream_t, const PicDev& pic, uint32_t numCu, uint32_t numTu, LfCell* cell, LfCell* cellC, LfMv* mv, uint32_t* ref, const LfSbCell* sb, int numSb, vvr_lfp* out0, vvr_lfp* out1 )
{
#ifdef VVT_NO_LF_STANDIN      // (tools/host_path_probe.py: time of the host stage alone)
  return;
#endif
  lf_init_maps_host( pic.hdr, pic.cu, numCu, pic.tu, numTu, cell, cellC, mv, ref, pic.w4, pic.h4 );
  for( int i = 0; i < numSb; i++ ) if( sb[i].cell < (uint32_t) ( pic.w4 * pic.h4 ) ) { mv[sb[i].cell] = lfi_pack_mv( sb[i].m ); ref[sb[i].cell] = lfi_pack_refs( sb[i].m ); }
  LfInitView V; V.hdr = &pic.hdr; V.cell = cell; V.cellC = cellC; V.mv = mv; V.ref = ref; V.ctuSlice = pic.ctuSlice; V.ctuTile = pic.ctuTile;
  V.ctuSubpic = pic.ctuSubpic; V.subpics = pic.subpics; V.slices = pic.slices; V.w4 = pic.w4; V.h4 = pic.h4; V.ctusX = pic.ctus_x;
  lf_init_tables_host( V, out0, out1 );
  g_lastLfp[0] = out0; g_lastLfp[1] = out1; g_lastLfpCells = pic.w4 * pic.h4;
}
bool sao_alf_fused( const PicDev& ) { return true; }
void launch_sao_alf( hipStream_t, const PicDev& pic, DevPlanes, DevPlanes dst, bool, bool ) { if( g_delayUs ) usleep( 2 * g_delayUs ); stamp_picture( pic, dst ); }
void launch_sao( hipStream_t, const PicDev&, DevPlanes, DevPlanes ) { if( g_delayUs ) usleep( 2 * g_delayUs ); }
void launch_alf( hipStream_t, const PicDev&, DevPlanes, DevPlanes ) {}
void launch_lmcs( hipStream_t, const PicDev&, DevPlanes, int ) {}
void launch_copy_planes( hipStream_t, DevPlanes, DevPlanes ) {}
void launch_copy_bytes( hipStream_t, const void*, void*, size_t ) {}
void launch_mc_affine( hipStream_t, const PicDev&, const RefSet&, DevPlanes, const McItem*, int ) {}
void launch_mc_rpr( hipStream_t, const PicDev&, const RefSet&, DevPlanes, const McItem*, int ) {}
void launch_mc_dmvr( hipStream_t, const PicDev&, const RefSet&, DevPlanes, const McItem*, int, int32_t* ) {}
static int g_lastIntraWg = 0;
size_t intra_sync_ints( int numUnits, int numItems ) { return ( ( (size_t) 1 + (size_t) numUnits + 63 ) & ~(size_t) 63 ) + (size_t) numItems * 64; }
void launch_intra( hipStream_t, const PicDev&, DevPlanes, DevPlanes, const IntraItem*, int, const IntraUnit*, int numUnits, int ticket0, int ticket1, int numWg, int* sync, int, uint32_t*, size_t, int, int, int* ) { (void) ticket1; if( ticket0 ) return; g_lastIntraWg = numWg; g_lastIntraUnits = numUnits; g_lastSync = sync; for( int i = 0; i <= numUnits; i++ ) sync[i] = 0; /* what the launcher's memset touches */ }
void launch_resi_add( hipStream_t, const PicDev&, DevPlanes, DevPlanes, const IntraItem*, int ) {}
static int g_lastLeafItems = -1;
size_t intra_leaf_map_ints( int w4, int h4, int vpdus ) { return (size_t) 3 * w4 * h4 + 2 * (size_t) vpdus + 64; }
void launch_intra_leaf( hipStream_t, const PicDev&, DevPlanes, DevPlanes, const IntraItem*, int numItems, const IntraItem*, int, uint32_t*, size_t, int, int, int* errWord ) { g_lastLeafItems = numItems; if( g_failLeafWaits > 0 && errWord ) { g_failLeafWaits--; *errWord = 1; } /* what a wavefront of k_intra_leaf does when a bounded wait gives up */ }
// the two output-stage kernels have functional stand-ins (a few plain loops with the kernels' contract: packed window; per row the checksum
// share or the CRC register reached from 0), so that the host half of vvr_read_output / vvr_picture_hash - window geometry, chaining the rows'
// CRC pieces - is checked against the reference's own functions without a GPU
void launch_output_window( hipStream_t, const pel_t* src, int stride, int w, int h, int bps, void* dst )
{
  for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ )
  {
    const uint16_t v = (uint16_t) src[(size_t) y * stride + x];
    if( bps == 2 ) ( (uint16_t*) dst )[(size_t) y * w + x] = v; else ( (uint8_t*) dst )[(size_t) y * w + x] = (uint8_t) v;
  }
}
void launch_plane_hash_rows( hipStream_t, const pel_t* plane, int stride, int w, int h, int two, int crcMode, uint32_t* out )
{
  for( int y = 0; y < h; y++ )
  {
    uint32_t acc = 0;
    for( int x = 0; x < w; x++ )
    {
      const uint32_t v = (uint16_t) plane[(size_t) y * stride + x];
      if( !crcMode ) { const uint32_t mask = ( ( x & 0xff ) ^ ( y & 0xff ) ^ ( x >> 8 ) ^ ( y >> 8 ) ) & 0xff; acc += ( v & 0xff ) ^ mask; if( two ) acc += ( v >> 8 ) ^ mask; }
      else for( int b = 0; b < ( two ? 2 : 1 ); b++ ) for( int bit = 7; bit >= 0; bit-- )
      {
        const uint32_t byte = b ? v >> 8 : v & 0xff, msb = ( acc >> 15 ) & 1;
        acc = ( ( ( acc << 1 ) + ( ( byte >> bit ) & 1 ) ) & 0xffff ) ^ ( msb * 0x1021 );
      }
    }
    out[y] = acc;
  }
}

extern "C" {
// the intra-stage tables of a prepared picture (host memory in this build): units in ticket order, items, counts
__attribute__(( visibility( "default" ) )) int vvt_intra_tables( const vvr_prepared* q, const IntraUnit** units, int* numUnits, const IntraItem** items, int* numItems )
{
  if( !q ) return -1;
  *units = q->units; *numUnits = q->numActive; *items = q->intraItems; *numItems = q->numIntra;
  return 0;
}
// the recorded stream / event operations (triples), cleared by the call
__attribute__(( visibility( "default" ) )) int vvt_take_trace( int* dst, int maxInts )
{
  const int n = (int) g_trace.size() < maxInts ? (int) g_trace.size() : maxInts;
  for( int i = 0; i < n; i++ ) dst[i] = g_trace[i];
  g_trace.clear();
  return n;
}
// everything vvr_prepare uploaded for one picture (work lists, tables, the description's arrays): one allocation
__attribute__(( visibility( "default" ) )) int vvt_blob( const vvr_prepared* q, const void** p, size_t* n ) { if( !q ) return -1; *p = q->blob; *n = q->blobBytes; return 0; }
// one logical table of a prepared picture (developer regression check of the host glue, tools/host_tables_hash.py): 0..3 MC tile lists (plain, BDOF,
// DMVR, affine), 4..6 transform block lists by size class, 7 intra-stage blocks, 8 intra-stage units, 9 residual-add blocks
__attribute__(( visibility( "default" ) )) int vvt_table( const vvr_prepared* q, int which, const void** p, size_t* n )
{
  if( !q ) return -1;
  switch( which )
  {
  case 0: *p = q->mcItems;   *n = sizeof( McItem ) * q->numMc; break;
  case 1: *p = nullptr; *n = 0; break;        // (BDOF and DMVR tiles: written by the device, see 10)
  case 2: *p = nullptr; *n = 0; break;
  case 3: *p = q->affItems;  *n = sizeof( McItem ) * q->numAffItems; break;
  case 4: case 5: case 6: *p = q->tbItems[which - 4]; *n = sizeof( TbItem ) * q->numTb[which - 4]; break;
  case 7: *p = q->intraItems; *n = sizeof( IntraItem ) * q->numIntra; break;
  case 8: *p = q->units; *n = sizeof( IntraUnit ) * q->numActive; break;
  case 10: *p = q->mcCus; *n = sizeof( McCuRef ) * q->numMcCus; break;      // CUs whose MC tiles the device writes
  case 9: *p = q->resiItems; *n = sizeof( IntraItem ) * q->numResi; break;      // residual-add blocks (k_resi_add)
  case 11: *p = q->rprItems; *n = sizeof( McItem ) * q->numRprItems; break;      // tiles of CUs with a scaled reference picture (k_mc_rpr)
  default: return -1;
  }
  return 0;
}
// asynchronous host-to-device copies issued so far: count and bytes (cleared by the call)
__attribute__(( visibility( "default" ) )) unsigned long long vvt_take_h2d_hash( void ) { const uint64_t v = g_h2dHash; g_h2dHash = 1469598103934665603ull; return v; }
__attribute__(( visibility( "default" ) )) void vvt_take_h2d( size_t* copies, size_t* bytes ) { *copies = g_h2dCopies; *bytes = g_h2dBytes; g_h2dCopies = g_h2dBytes = 0; }
__attribute__(( visibility( "default" ) )) void vvt_set_delay( int us ) { g_delayUs = us; }
__attribute__(( visibility( "default" ) )) void vvt_fail_leaf_waits( int n ) { g_failLeafWaits = n; }
// the edge-parameter tables the last launch_lf_init derived ("device" memory of the picture's image: valid until the ring entry is reused)
__attribute__(( visibility( "default" ) )) int vvt_last_lfp( const vvr_lfp** d0, const vvr_lfp** d1 ) { *d0 = g_lastLfp[0]; *d1 = g_lastLfp[1]; return g_lastLfpCells; }
__attribute__(( visibility( "default" ) )) void vvt_slow_i_pictures( int us ) { g_vvtSlowIUs = us; }
__attribute__(( visibility( "default" ) )) void vvt_slow_b_pictures( int us ) { g_vvtSlowBUs = us; }
__attribute__(( visibility( "default" ) )) void vvt_events_pending( int on ) { g_eventsPending = on; }
__attribute__(( visibility( "default" ) )) int vvt_band_pictures( void ) { return vvr_host_band_pictures(); }      // pictures with inter CUs whose work lists were built in bands by the workers together
__attribute__(( visibility( "default" ) )) int vvt_dead_event_uses( void ) { return g_deadEventUses; }      // uses of an event after hipEventDestroy since the library was loaded
__attribute__(( visibility( "default" ) )) unsigned long long vvt_overtakes( vvr_context* c ) { return c->overtakes; }
__attribute__(( visibility( "default" ) )) size_t vvt_sizeof( int which ) { return which == 0 ? sizeof( IntraUnit ) : which == 1 ? sizeof( IntraItem ) : 0; }
// the stage's launches of a prepared picture: number of luma units that go first when the picture has residual-add blocks (else 0), workgroups of both launches
__attribute__(( visibility( "default" ) )) void vvt_intra_launches( const vvr_prepared* q, int* numLumaUnits, int* wg0, int* wg1 ) { *numLumaUnits = q->numLumaUnits; *wg0 = q->intraWorkgroups; *wg1 = q->intraWorkgroupsChroma; }
__attribute__(( visibility( "default" ) )) int vvt_last_intra_launch( void ) { return g_lastIntraUnits; }
__attribute__(( visibility( "default" ) )) int vvt_last_leaf_items( void ) { return g_lastLeafItems; }      // items of the last k_intra_leaf launch
__attribute__(( visibility( "default" ) )) int vvt_is_leaf( const vvr_prepared* q ) { return q && q->intraLeaf ? 1 : 0; }
__attribute__(( visibility( "default" ) )) int vvt_last_intra_wg( void ) { return g_lastIntraWg; }
// pretend the lane's flag buffer is small (the product sizes it for ordinary pictures; the growth path needs a picture with more units than that)
__attribute__(( visibility( "default" ) )) void vvt_shrink_sync( vvr_context* c, int lane, size_t cap ) { if( c && lane < (int) c->syncCap.size() && cap < c->syncCap[lane] ) c->syncCap[lane] = cap; }
__attribute__(( visibility( "default" ) )) size_t vvt_sync_capacity( const vvr_context* c, int lane ) { return c && lane < (int) c->syncCap.size() ? c->syncCap[lane] : 0; }
}
