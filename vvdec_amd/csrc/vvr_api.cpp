// vvdec_amd/csrc/vvr_api.cpp — host side of the reconstruction back-end: the C ABI of include/vvr.h.
//
// Mirrors DecLibRecon (reference: source/Lib/DecoderLib/DecLibRecon.cpp): create() owns the per-instance resources
// (there: per-thread tool objects + scratch, :132-168; here: HIP streams, DPB planes, scratch planes, constant tables, the upload ring and the
// host worker threads), vvr_submit() is decompressPicture() (:429) and vvr_wait() is waitForPrevDecompressedPic() (:684).
//
// A submitted picture goes through a three-stage pipeline, several pictures deep:
//   1. prepare (host, vvr_prepare.cpp): validation, then the device work lists, built by one of `host_threads` worker threads (or by the
//      submitting thread when host_threads is 0) and packed into the pinned half of a ring entry — the reference does the same set-up inline
//      in decompressPicture (:429-682) and spreads it over its thread pool;
//   2. commit (by the launcher thread in submission order, except that a ready picture passes pictures still in stage 1 that it shares no DPB
//      slot with, nextToCommitLocked - by the submitting thread when host_threads is 0): asynchronous H2D copies on the copy
//      stream (the ring entry; record arrays in pinned caller memory straight from where they are), then the kernels of the picture on one of
//      `num_streams` HIP streams (an I picture: on the high-priority stream when that is free), ordered against other pictures by whole-picture HIP events (reference pictures: the "refPicExtDepBarriers" of
//      :544-581; slot reuse: write-after-read);
//   3. completion: two events per picture, one later pictures' streams wait for and one host threads wait for (hipEventSynchronize holds the event's
//      lock while it waits: a hipStreamWaitEvent on the same event would stall the launcher behind it); DMVR delta MVs and the collocated motion
//      are written by the DMVR kernel straight into device-mapped pinned memory (no device-to-host copy call: it blocked the calling thread).
// Nothing is allocated, freed or synchronised device-wide on this path: the ring is allocated when the context is created.
// There is NO CPU fallback: without a gfx950 device every entry point fails with VVR_ERR_NO_DEVICE.
#include "vvr_host.h"
#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <sched.h>
#include <pthread.h>

int vvr_upload_tables();

// Host threads of a context stay on the NUMA node its GPU hangs off: the work lists are written into pinned memory the device reads, and a
// worker that wanders to the other socket of a two-socket host takes 6-9 instead of 4.4 ms for a 4K picture (round 2: the spread of the
// benchmark line).  The node comes from sysfs (PCI bus id of the HIP device -> numa_node -> cpulist); VVR_NO_PIN=1 leaves the threads alone.
static std::vector<int> gpuNodeCpus( int device )
{
  std::vector<int> cpus;
  if( const char* e = getenv( "VVR_NO_PIN" ) ) if( atoi( e ) ) return cpus;
  char bus[64] = { 0 };
  if( hipDeviceGetPCIBusId( bus, sizeof( bus ), device ) != hipSuccess ) return cpus;
  for( char* q = bus; *q; q++ ) *q = (char) tolower( *q );
  char path[256]; snprintf( path, sizeof( path ), "/sys/bus/pci/devices/%s/numa_node", bus );
  int node = -1;
  if( FILE* f = fopen( path, "r" ) ) { if( fscanf( f, "%d", &node ) != 1 ) node = -1; fclose( f ); }
  if( node < 0 ) return cpus;
  snprintf( path, sizeof( path ), "/sys/devices/system/node/node%d/cpulist", node );
  if( FILE* f = fopen( path, "r" ) )
  {
    int a, b; char sep;
    while( fscanf( f, "%d", &a ) == 1 )
    {
      b = a;
      const int ch = fgetc( f );
      if( ch == '-' ) { if( fscanf( f, "%d", &b ) != 1 ) b = a; sep = (char) fgetc( f ); (void) sep; }
      for( int k = a; k <= b && k < CPU_SETSIZE; k++ ) cpus.push_back( k );
      if( ch == EOF || ch == '\n' ) break;
    }
    fclose( f );
  }
  return cpus;
}
static void pinToCpus( const std::vector<int>& cpus )
{
  if( cpus.empty() ) return;
  cpu_set_t set; CPU_ZERO( &set );
  for( int k : cpus ) CPU_SET( k, &set );
  pthread_setaffinity_np( pthread_self(), sizeof( set ), &set );      // (a failure leaves the thread where it was: nothing depends on it)
}

#define HIPCHK( ctx, call ) do { hipError_t e_ = ( call ); if( e_ != hipSuccess ) { ( ctx )->setError( std::string( #call ) + ": " + hipGetErrorString( e_ ) ); return VVR_ERR_DEVICE; } } while( 0 )

#ifdef VVR_WATCHDOG
#include <atomic>
#include <chrono>
static std::atomic<uint64_t> g_wdProgress{ 0 };
static double g_wdEnqMs = 0, g_wdEnqMax = 0; static uint64_t g_wdEnqN = 0;      // time the committing thread spends enqueuing pictures
#define WD_PROGRESS() g_wdProgress.fetch_add( 1 )
static double wdNow() { return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count(); }
#define WD_STAMP( job, field ) ( job ).field = wdNow()
static double g_wdSum[6]; static uint64_t g_wdJobs; static double g_wdPart[4], g_wdPartMax[4]; static uint64_t g_wdPartN; static double g_wdCall[K_NUM][3];
#else
#define WD_PROGRESS() do {} while( 0 )
#define WD_STAMP( job, field ) do {} while( 0 )
#endif

struct Stat { uint64_t launches = 0; double ms = 0, bytes = 0; };
struct PendingTiming { hipEvent_t a, b; int kernel; double bytes; };
const char* const kKernelNames[K_NUM] = { "k_mc", "k_mc_dmvr", "k_mc_affine", "k_lmcs", "k_itrans", "k_intra", "k_resi_add", "k_deblock_v", "k_deblock_h", "k_sao", "k_alf", "k_copy", "k_output", "k_lf_init", "k_intra_leaf", "k_deblock4", "k_alf_planes" };

// One slot of the upload ring: pinned staging memory and its image in HBM (grown on demand, never freed while the context lives), the device
// pointers of the picture that currently sits in it, and the pinned landing area of its DMVR delta MVs.
struct RingEntry {
  char* host = nullptr; size_t hostCap = 0;
  char* dev = nullptr;  size_t devCap = 0;
  int32_t* dmvrHost = nullptr; size_t dmvrCap = 0;      // ints
  vvr_motion* colHost = nullptr; size_t colCap = 0;     // records: collocated motion of the picture (VVR_TOOL_COL_MOTION), pinned + device-mapped
  hipEvent_t copied = nullptr;
  vvr_prepared q;
  size_t stagedBegin = 0, stagedEnd = 0;                // the part of the image that goes through `host`
  std::vector<DirectCopy> direct;                       // arrays copied straight from the caller's pinned memory
  struct Job* owner = nullptr;                          // the job whose picture sits in the entry (nullptr: free)
  uint64_t turn = 0;                                    // ring number (Job::ringSeq) of the streaming job the entry serves next: entry i serves i, i + R, i + 2R, ...
};

enum { J_QUEUED, J_PREPARING, J_READY, J_FAILED, J_COMMITTED };

#define VVR_ERR_WORDS 1024
struct Job {
  int id = 0; uint64_t seq = 0;
  uint64_t ringSeq = 0;             // streaming jobs: position in the sequence of ring users (entry = ringSeq % ring size)
  int state = J_QUEUED;
  int rc = VVR_OK; std::string err;
  vvr_picture pic;                  // shallow copy: the arrays stay the caller's until the job is prepared
  vvr_prepared* q = nullptr;        // what the kernels read (ring entry or resident handle)
  RingEntry* ring = nullptr;
  int lane = -1;
  hipEvent_t done = nullptr;          // what later pictures' streams wait for
  hipEvent_t doneHost = nullptr;      // what host threads wait for: hipEventSynchronize holds the event's lock for as long as it waits, and a
                                      // hipStreamWaitEvent on the same event (the launcher ordering a later picture) would block behind it
  bool completed = false, waited = false;
  std::vector<PendingTiming> timings;
  int* errWord = nullptr;           // a word of pinned host memory the intra stage writes when one of its bounded waits gave up: the job fails when it completes
  std::vector<int32_t> dmvr;        // delta MVs, copied out of pinned memory when the job completes
  std::vector<vvr_motion> col;      // collocated motion (VVR_TOOL_COL_MOTION), likewise
  bool handled = false;             // committed (or failed) ahead of its turn: its bySeq entry stays until the commit front passes it
#ifdef VVR_WATCHDOG
  double tSubmit = 0, tPrep = 0, tBuilt = 0, tRing = 0, tReady = 0, tCommit0 = 0, tCommit1 = 0; hipEvent_t tlA = nullptr, tlB = nullptr; int tlPoc = 0, tlType = 0; unsigned long long tlSeq = 0;      // developer build: where a picture spends its time on the host
#endif
};

struct vvr_context {
  vvr_config cfg;
  int        device = 0;
  std::string err;
  std::vector<hipStream_t> streams;
  hipStream_t copyStream = nullptr;
  std::vector<DevPlanes>   slots;       // DPB
  std::vector<std::pair<uint16_t, uint16_t>> slotDim;      // luma size of the picture last submitted into each slot (the context's size before that): what vvr_picture_hash covers
  std::vector<DevPlanes>   scratchB;    // per stream: second picture (SAO output)
  std::vector<DevPlanes>   scratchR;    // per stream: residual planes (intra)
  void*      planeMem = nullptr; bool planeMemOwned = false;
  void*      scratchMem = nullptr;
  int        partsComing = 0;           // workers that have taken an I picture and are about to publish its parts in `subtasks`: the others wait for the parts instead of starting on another picture (guarded by mu)
  std::deque<std::function<void( PrepScratch& )>> subtasks;      // parts of a picture's host stage that any worker may run (an I picture is prepared by all of them together); guarded by mu, served before `queue`
  std::vector<int*> syncBuf;            // per stream: ticket + one flag per unit of the intra stage
  std::vector<size_t> syncCap;          // ints allocated in syncBuf[lane]; grown when a picture has more units
  std::vector<uint32_t*> leafMaps;      // per stream: the cell maps, VPDU flags and factors of k_intra_leaf (all zero between launches), the ticket at the end
  int*       errHost = nullptr;         // VVR_ERR_WORDS words of pinned host memory: the error word of job j is errHost[j % VVR_ERR_WORDS] (far more than pictures in flight)
  size_t     leafMapInts = 0; int leafW4 = 0, leafH4 = 0;
  bool       intraFine = false;         // VVR_INTRA_FINE=1: pictures of intra CTUs resolve their CTU wavefront block by block (k_intra<.., FINE>).  Measured in round 5 and left off:
                                        // an I picture alone 4.23 - 4.46 ms against 4.56 (its blocks read far down the left CTU's last column, the chain of 1442 blocks stays), and
                                        // with other pictures in flight the 256 polling workgroups of nine wavefronts cost more than they gain (4K RA 1555 against 1801 frames/s,
                                        // all-intra 547 against 944)
  bool       leafByLevel = false;       // ... their blocks listed by level instead of decoding order (VVR_LEAF_BY_LEVEL=1; vvr_prepare.cpp, `leafSort`): measured, no gain
  bool       intraLeaf = true;          // pictures with scattered intra blocks take the one-wavefront-per-block path (VVR_INTRA_LEAF=0: the CTU-tile path for everything)
  size_t     planeBytes[3] = { 0, 0, 0 }, slotBytes = 0;
  int        stride[3] = { 0, 0, 0 };
  // output stage scratch (device + pinned), grown on demand
  void*      outDev = nullptr; size_t outDevCap = 0;
  void*      outHost = nullptr; size_t outHostCap = 0;
  hipStream_t outStream = nullptr;                 // device-to-host copies of vvr_read_picture
  char*      prepStage = nullptr; size_t prepStageCap = 0;      // pinned staging of vvr_prepare
  std::vector<void*> stagePool;                    // pinned staging buffers of vvr_read_picture (one picture each), handed out under `mu`
  // ---- job pipeline (everything below is guarded by mu)
  std::mutex mu, commitMu;              // commitMu: one committing thread at a time (it takes mu only around its bookkeeping)
  std::condition_variable cv;
  std::map<int, std::unique_ptr<Job>> jobs;
  std::deque<Job*> queue;               // submitted, not yet taken by a worker
  int        nextJob = 0, nextStream = 0;
  int        numLanesRR = 1;               // lanes the pictures go round: all but the priority lane
  int        prioLane = -1, prioJob = -1;  // lane with a high-priority stream for I pictures (planCommitLocked), the last picture that took it
  uint64_t   nextSeq = 0, nextCommit = 0, nextRingSeq = 0;
  uint64_t   overtakes = 0;                // pictures enqueued ahead of a picture submitted before them that was still being prepared (nextToCommitLocked)
  std::map<uint64_t, Job*> bySeq;       // jobs that have not been committed yet
  std::vector<std::vector<int>> slotUsers;   // job ids that touched a slot since it was last written (first entry: the writer)
  std::vector<std::vector<hipEvent_t>> slotExt;   // events of external work on a slot (vvr_slot_external_event): pictures that use the slot wait for them
  std::vector<RingEntry> ring;
  size_t     ringLargest = 0;           // bytes of the largest picture image seen (+ 25 %): what a ring entry grows to
  size_t     ringLargestHost = 0;       // ... and of its uploaded part (the pinned half of an entry)
  std::vector<char*> retiredHost, retiredDev;   // outgrown ring buffers, freed with the context
  std::vector<hipEvent_t> eventPool;
  std::vector<int> nodeCpus;            // CPUs of the NUMA node the device is attached to (empty: unknown / pinning off)
  std::vector<std::thread> workers;
  std::thread launcher;                 // commits prepared pictures (contexts with worker threads), see nextToCommitLocked for the order
#ifdef VVR_WATCHDOG
  std::thread watchdog;
#endif
  bool       stop = false;
  PrepScratch* inlineScratch = nullptr; // host_threads == 0, and vvr_prepare
  PinnedRanges pinned;                  // vvr_host_alloc
  bool       statsOn = false;
  int        partsPolicy = 2;              // pictures with inter CUs built in bands by the workers together: 0 never, 1 always, 2 while the device is short of work (VVR_PARTS)
#ifdef VVR_WATCHDOG
  hipEvent_t tlBase = nullptr; double tlBaseHost = 0, tl0 = 0;      // developer build, VVR_TIMELINE: device times of a picture relative to the first commit of a burst
#endif
  Stat       stats[K_NUM];
  void setError( const std::string& e ) { err = e; }
};

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

// ---------------------------------------------------------------------------------------------------------------------
// job pipeline
// ---------------------------------------------------------------------------------------------------------------------
static hipEvent_t takeEvent( vvr_context* c )
{
  if( !c->eventPool.empty() ) { hipEvent_t e = c->eventPool.back(); c->eventPool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if( hipEventCreateWithFlags( &e, hipEventDisableTiming ) != hipSuccess ) return nullptr;
  return e;
}

// the job's picture is reconstructed (its `done` event has been waited for): statistics, delta MVs, the ring entry is free again.  mu held.
static void completeLocked( vvr_context* c, Job& j )
{
  if( j.completed ) return;
#ifdef VVR_WATCHDOG
  if( j.tlA && j.tlB && c->tlBase )
  {
    float a = 0, b = 0; hipEventElapsedTime( &a, c->tlBase, j.tlA ); hipEventElapsedTime( &b, c->tlBase, j.tlB );
    fprintf( stderr, "[vvr timeline] device seq %3llu poc %4d type %d lane %d: %6.2f .. %6.2f  (base event recorded at host %6.2f)\n", j.tlSeq, j.tlPoc, j.tlType, j.lane, a, b, c->tlBaseHost - c->tl0 );
  }
  if( j.tlA ) { hipEventDestroy( j.tlA ); j.tlA = nullptr; } if( j.tlB ) { hipEventDestroy( j.tlB ); j.tlB = nullptr; }
#endif
  for( auto& t : j.timings )
  {
    float ms = 0; hipEventElapsedTime( &ms, t.a, t.b );
    c->stats[t.kernel].launches++; c->stats[t.kernel].ms += ms; c->stats[t.kernel].bytes += t.bytes;
    hipEventDestroy( t.a ); hipEventDestroy( t.b );
  }
  j.timings.clear();
  if( j.errWord && *j.errWord )
  {
    // k_intra_leaf bounds every wait for a neighbouring block: a wavefront that gave up has reconstructed from whatever was there - the picture is wrong
    *j.errWord = 0;
    if( j.state == J_COMMITTED ) { j.state = J_FAILED; j.rc = VVR_ERR_DEVICE; j.err = "intra stage: a block waited for its neighbours beyond the bound (dependency that never completed)"; }
  }
  if( j.q && j.q->numDmvr && j.state == J_COMMITTED )
  {
    // refined MVs feed the temporal MV prediction of later pictures on the host (DecCu::TaskFinishMotionInfo, DecCu.cpp:161)
    const int32_t* src = j.ring ? j.ring->dmvrHost : j.q->dmvrHost;
    j.dmvr.assign( src, src + 2 * (size_t) j.q->numDmvr );
  }
  if( j.q && j.q->colHost && j.state == J_COMMITTED ) j.col.assign( j.q->colHost, j.q->colHost + j.q->numCol );
  if( j.ring && j.ring->owner == &j ) { j.ring->owner = nullptr; j.ring->turn += c->ring.size(); }
  if( j.done ) { c->eventPool.push_back( j.done ); j.done = nullptr; }
  if( j.doneHost ) { c->eventPool.push_back( j.doneHost ); j.doneHost = nullptr; }
  j.q = nullptr;
  j.completed = true;
  WD_PROGRESS();
  c->cv.notify_all();
}

// What the committer decides about a picture while it holds mu: its lane, the events of the pictures it has to be ordered behind, its
// reference planes.  The HIP calls themselves (enqueuePicture) run WITHOUT mu: a launch blocks when the device's queues are full, and the
// worker threads must be able to go on preparing pictures meanwhile.
struct CommitPlan { int lane; std::vector<hipEvent_t> waits; RefSet refs; std::vector<int> waitInfo; std::vector<hipEvent_t> outExt /* external events of the output slot this picture waits for */; };

// External events of a slot (vvr_slot_external_event) that the device has passed are dropped: nothing has to wait for them any more, and the caller
// may destroy an event once it is complete and the back-end has been through vvr_sync (vvr.h).  Called with mu held - by vvr_sync ONLY: hipEventQuery says
// hipSuccess for an event that has not been recorded yet as well, so between two vvr_sync calls every registered event is waited for by whoever uses the slot
// (a wait for a complete event costs nothing on the device); the events of a slot are dropped unasked when a picture overwrites the slot (it has waited for them,
// and everybody after it waits for the picture) or an external writer registers (vvr_slot_external_event, writes).
static void pruneExternalEventsLocked( vvr_context* c, int slot )
{
  auto& v = c->slotExt[slot];
  v.erase( std::remove_if( v.begin(), v.end(), []( hipEvent_t ev ) { return hipEventQuery( ev ) == hipSuccess; } ), v.end() );
}

static void planCommitLocked( vvr_context* c, Job& job, CommitPlan& plan )
{
  const vvr_pic_header& h = job.q->hdr;
  // An I picture of a stream with inter pictures takes the priority lane when that is free: its intra stage is one 8 ms chain of dependent blocks
  // (36 workgroups at 4K) that the whole next GOP waits for, while the kernels of the B pictures in flight are bulk work that fills whatever is
  // left - queued behind them (20 4K pictures arriving at once: 15 of them enqueued before it) it was seen to finish 5 ms later in one run out of
  // two.  The hardware queue of a high-priority stream is served first when workgroup slots free up; nothing that runs is pre-empted.  A stream of
  // I pictures only (all-intra) finds the lane taken by the picture before and goes round the other lanes as ever.
  int lane = -1;
  if( c->prioLane >= 0 && h.slice_type == 2 )
  {
    auto it = c->jobs.find( c->prioJob );
    bool idle = it == c->jobs.end() || it->second->completed;
    // (the event the streams wait on, not the one host threads wait on: hipEventSynchronize holds the event's lock while it waits - see the
    // two completion events of a job - and this thread holds mu)
    if( !idle && it->second->state == J_COMMITTED && it->second->done ) idle = hipEventQuery( it->second->done ) == hipSuccess;
    if( idle ) { lane = c->prioLane; c->prioJob = job.id; }
  }
  if( lane < 0 ) { lane = c->nextStream; c->nextStream = ( c->nextStream + 1 ) % c->numLanesRR; }
  plan.lane = lane; plan.waits.clear(); plan.waitInfo.clear();
  job.lane = lane;
  // a lane's scratch planes are reused: the previous job of this lane is ordered before us by the stream itself
  // ---- dependencies: every job that read or wrote one of our slots
  auto waitFor = [&]( int id ) { auto it = c->jobs.find( id ); if( it != c->jobs.end() ) { Job& j = *it->second; if( !j.completed && j.state == J_COMMITTED && j.lane != lane ) { plan.waits.push_back( j.done ); plan.waitInfo.push_back( j.q ? ( j.id * 16 + j.q->hdr.slice_type * 4 ) : -1 ); plan.waitInfo.push_back( j.lane ); } } };
  for( int id : c->slotUsers[h.out_slot] ) waitFor( id );
  // external work on our slots (a collective that wrote a reference slot, or still reads the slot we overwrite)
  // (this picture waits for them, and whoever uses the slot afterwards waits for this picture: they are dropped when the picture HAS been enqueued - exactly these
  // handles, the lock is released in between and another may have been registered; a picture that could not be enqueued leaves them where they are)
  plan.outExt.clear();
  for( hipEvent_t ev : c->slotExt[h.out_slot] ) { plan.waits.push_back( ev ); plan.waitInfo.push_back( -2 ); plan.waitInfo.push_back( -1 ); plan.outExt.push_back( ev ); }
  memset( &plan.refs, 0, sizeof( plan.refs ) );
  if( h.slice_type != 2 )
    for( int l = 0; l < 2; l++ ) for( int i = 0; i < h.num_ref[l]; i++ )
    {
      const int slot = h.ref_slot[l][i];
      // wait for the writer of the reference (it is the first entry since the slot was last written)
      if( !c->slotUsers[slot].empty() ) waitFor( c->slotUsers[slot][0] );
      for( hipEvent_t ev : c->slotExt[slot] ) if( std::find( plan.waits.begin(), plan.waits.end(), ev ) == plan.waits.end() ) { plan.waits.push_back( ev ); plan.waitInfo.push_back( -2 ); plan.waitInfo.push_back( -1 ); }
      for( int k = 0; k < 3; k++ ) plan.refs.p[l * VVR_MAX_REFS + i][k] = c->slots[slot].p[k];
    }
  job.done = takeEvent( c ); job.doneHost = takeEvent( c );
}

// enqueue one prepared picture: H2D copy of its ring entry, dependencies, kernels.  Called by the one committing thread, without mu.
static int enqueuePicture( vvr_context* c, Job& job, const CommitPlan& plan, std::string& err )
{
#undef HIPCHK
  // a failure after the first copy / kernel of the picture has been enqueued leaves work in flight: the copy stream still reads the ring entry, the
  // lane still writes the output slot.  Every failure exit therefore drains both before it returns - the caller marks the job failed, gives the
  // ring entry to the next picture and does not record this one as a user of its slots, all of which is only safe once nothing of it runs.
#define HIPCHK( ctx, call ) do { hipError_t e_ = ( call ); if( e_ != hipSuccess ) { err = std::string( #call ) + ": " + hipGetErrorString( e_ ); hipStreamSynchronize( ( ctx )->copyStream ); hipStreamSynchronize( ( ctx )->streams[plan.lane] ); return VVR_ERR_DEVICE; } } while( 0 )
  vvr_prepared* q = job.q;
  const vvr_pic_header& h = q->hdr;
  const int lane = plan.lane;
  hipStream_t s = c->streams[lane];
  if( !job.done || !job.doneHost ) { err = "hipEventCreate failed"; return VVR_ERR_DEVICE; }
#ifdef VVR_WATCHDOG
  const double wdA = wdNow();
#endif
  if( job.ring )
  {
    RingEntry& e = *job.ring;
    HIPCHK( c, hipMemcpyAsync( e.dev + e.stagedBegin, e.host + e.stagedBegin, e.stagedEnd - e.stagedBegin, hipMemcpyHostToDevice, c->copyStream ) );
    for( auto& d : e.direct ) HIPCHK( c, hipMemcpyAsync( e.dev + d.off, d.src, d.n, hipMemcpyHostToDevice, c->copyStream ) );
    HIPCHK( c, hipEventRecord( e.copied, c->copyStream ) );
    HIPCHK( c, hipStreamWaitEvent( s, e.copied, 0 ) );
  }
#ifdef VVR_WATCHDOG
  const double wdB = wdNow(); g_wdPart[0] += wdB - wdA; g_wdPartMax[0] = std::max( g_wdPartMax[0], wdB - wdA );
#endif
#ifdef VVR_WATCHDOG
  for( size_t wi = 0; wi < plan.waits.size(); wi++ )
  {
    const double w0 = wdNow(); HIPCHK( c, hipStreamWaitEvent( s, plan.waits[wi], 0 ) ); const double w = wdNow() - w0;
    if( w > 0.5 ) fprintf( stderr, "[vvr] slow wait: picture %d (type %d, lane %d) wait %zu of %zu on job %d (type %d, lane %d): %.2f ms\n", job.id, (int) h.slice_type, lane, wi, plan.waits.size(), plan.waitInfo[2 * wi] >> 4, ( plan.waitInfo[2 * wi] >> 2 ) & 3, plan.waitInfo[2 * wi + 1], w );
  }
#else
  for( hipEvent_t ev : plan.waits ) HIPCHK( c, hipStreamWaitEvent( s, ev, 0 ) );      // (a picture that cannot be ordered behind its references must not run)
#endif
#ifdef VVR_WATCHDOG
  const double wdC = wdNow(); g_wdPart[1] += wdC - wdB; g_wdPartMax[1] = std::max( g_wdPartMax[1], wdC - wdB );
  // VVR_TIMELINE: when the picture's kernels start and end on the device, relative to an event recorded with the first commit of a burst
  static const bool tlOn = getenv( "VVR_TIMELINE" ) != nullptr;
  if( tlOn && job.ring )
  {
    if( job.tSubmit - c->tl0 > 50 )
    {
      static hipStream_t tlStream = nullptr; if( !tlStream ) hipStreamCreate( &tlStream );      // (an idle stream: the event completes when it is recorded)
      c->tl0 = job.tSubmit; hipEventCreate( &c->tlBase ); hipEventRecord( c->tlBase, tlStream ); c->tlBaseHost = wdNow();
      fprintf( stderr, "[vvr timeline] ---- (host: ms since the first submit of the burst; device: ms since the base event)\n" );
    }
    hipEventCreate( &job.tlA ); hipEventCreate( &job.tlB ); job.tlPoc = h.poc; job.tlType = h.slice_type; job.tlSeq = job.seq;
    hipEventRecord( job.tlA, s );
  }
#endif
  job.errWord = c->errHost + ( (unsigned) job.id % VVR_ERR_WORDS ); *job.errWord = 0;
  const RefSet& refs = plan.refs;
  DevPlanes A = c->slots[h.out_slot], B = c->scratchB[lane], R = c->scratchR[lane];
  // the picture's own size (it may be smaller than the context's pictures: it lies in the top left corner of its slot and of the scratch planes)
  for( int k = 0; k < 3; k++ ) { const int w = k ? h.width >> 1 : h.width, hh = k ? h.height >> 1 : h.height; A.w[k] = B.w[k] = R.w[k] = w; A.h[k] = B.h[k] = R.h[k] = hh; }
  // SAO and ALF in one pass (k_sao_alf): the picture is reconstructed and deblocked in the lane's scratch picture, the pass writes the DPB slot; else
  // everything up to SAO works in the slot, SAO writes the scratch picture and ALF the slot (the reference's m_fltBuf round trip)
  const int stopAfter = c->cfg.stop_after;       // conformance aid: vvr_config.stop_after
  const bool sao = ( h.tool_flags & ( VVR_TOOL_SAO_LUMA | VVR_TOOL_SAO_CHROMA ) ) != 0 && stopAfter != 1 && stopAfter != 2;
  const bool alf = ( h.tool_flags & VVR_TOOL_ALF ) != 0 && stopAfter == 0;
  const bool fused = ( sao || alf ) && sao_alf_fused( q->pic );
  // With the fused SAO + ALF pass (scratch -> slot) behind them the two deblocking passes run out of place, a tile per workgroup: vertical edges slot -> scratch
  // picture (the inverse luma mapping in the load), horizontal edges scratch picture -> the residual planes (free once the picture is reconstructed), SAO + ALF from
  // there into the slot.  Without deblocking: reconstructed in the scratch picture.
  const bool dbOn = !( h.tool_flags & VVR_TOOL_DEBLOCK_OFF ) && stopAfter != 1;
  const bool hop = fused && dbOn;
  const DevPlanes P = ( fused && !hop ) ? B : A;
  auto timedOn = [&]( int k, hipStream_t st, double algoBytes, auto&& fn )
  {
#ifdef VVR_WATCHDOG
    // developer experiment: what a stage costs in throughput (VVR_SKIP_KERNELS = bit mask over the kernel ids; the pictures are wrong, of course)
    static const int skipMask = getenv( "VVR_SKIP_KERNELS" ) ? atoi( getenv( "VVR_SKIP_KERNELS" ) ) : 0;
    if( skipMask & ( 1 << k ) ) return;
    const double w0 = wdNow();
#endif
    // (the events bracket the launch on the stream it is issued to)
    if( c->statsOn ) { PendingTiming t; hipEventCreate( &t.a ); hipEventCreate( &t.b ); t.kernel = k; t.bytes = algoBytes; hipEventRecord( t.a, st ); fn(); hipEventRecord( t.b, st ); job.timings.push_back( t ); }
    else fn();
#ifdef VVR_WATCHDOG
    const double w = wdNow() - w0; g_wdCall[k][0] += w; g_wdCall[k][1] = std::max( g_wdCall[k][1], w ); g_wdCall[k][2] += 1;
#endif
  };
  auto timed = [&]( int k, auto&& fn ) { timedOn( k, s, q->bytes[k], fn ); };
  // INTER stage: prediction of every inter CU, then residual add (DecLibRecon.cpp:831-874)
  // (the four launches write disjoint tiles.  Running them side by side on extra streams of the lane, forked and joined with events, was measured:
  // device-only throughput fell from 1870 to 1240 pictures/s with 4 lanes, to 970 with 8 - the cross-stream waits cost more than the overlap gives)
  // (the tiles of plain, BDOF and DMVR CUs are written on the device from the CU records: the host only counted them)
  {
    PrepWork w;
    if( q->numMcCus ) { w.mcCus = q->mcCus; w.numMcCus = q->numMcCus; w.plain = q->mcDev; w.bdof = q->bdofItems; w.dmvr = q->dmvrItems; }
    if( q->lfpOnDevice && dbOn ) { w.lfMaps = true; w.numCu = q->numCu; w.numTu = q->numTu; w.cell = q->lfCell; w.cellC = q->lfCellC; w.mv = q->lfMv; w.ref = q->lfRef; w.sb = q->lfSb; w.numSb = q->numLfSb; }
    if( q->intraLeaf && q->numIntra ) { w.items = q->intraItems; w.numItems = q->numIntra; w.resi = q->resiItems; w.numResi = q->numResi; w.maps = c->leafMaps[lane]; w.mapInts = c->leafMapInts; w.mapW4 = c->leafW4; w.mapH4 = c->leafH4; }
    launch_prep( s, q->pic, w );
  }
  // LF_INIT (DecLibRecon.cpp:807-829): the edge parameters of the deblocking passes from the CU / TU records, where the caller leaves them to the back-end
  if( q->lfpOnDevice && dbOn ) timed( K_LF_INIT, [&]{ launch_lf_init( s, q->pic, q->numCu, q->numTu, q->lfCell, q->lfCellC, q->lfMv, q->lfRef, q->lfSb, q->numLfSb, q->lfpDev[0], q->lfpDev[1] ); } );
  if( q->numMc + q->numMcDev ) timedOn( K_MC, s, q->bytes[K_MC] - q->bytesBdof, [&]{ launch_mc( s, q->pic, refs, P, q->mcItems, q->numMc, q->mcDev, q->numMcDev, 0 ); } );
  if( q->numBdofItems ) timedOn( K_MC, s, q->bytesBdof, [&]{ launch_mc( s, q->pic, refs, P, nullptr, 0, q->bdofItems, q->numBdofItems, 1 ); } );
  if( q->numDmvrItems )
  {
    // the delta MVs go straight into pinned host memory (device-mapped): a few bytes per 16x16 sub-block, and no copy call on the
    // launcher's path - hipMemcpyAsync device-to-host was found to block the calling thread until the stream had drained
    int32_t* out = job.ring ? job.ring->dmvrHost : q->dmvrHost;
    memset( out, 0, sizeof( int32_t ) * 2 * (size_t) q->numDmvr );        // (offsets no CU owns stay zero)
    timed( K_MC_DMVR, [&]{ launch_mc_dmvr( s, q->pic, refs, P, q->dmvrItems, q->numDmvrItems, out ); } );
  }
  if( q->numAffItems ) timed( K_MC_AFFINE, [&]{ launch_mc_affine( s, q->pic, refs, P, q->affItems, q->numAffItems ); } );
  if( q->numRprItems ) launch_mc_rpr( s, q->pic, refs, P, q->rprItems, q->numRprItems );      // CUs that read a scaled reference picture
  const bool lmcsOn = ( h.tool_flags & VVR_TOOL_LMCS ) != 0;
  // LMCS: the inter prediction is forward-mapped before any residual is added (DecCu.cpp:458-476) - by the motion-compensation kernels themselves
  // where they store their luma samples (lmcs_fwd_luma): no pass over the picture
  for( int k = 0; k < 3; k++ ) if( q->numTb[k] ) timedOn( K_ITRANS, s, q->bytesTb[k], [&]{ launch_itrans( s, q->pic, P, R, q->tbItems[k], q->numTb[k], 16 << k ); } );
  // INTRA stage: wavefront over the CTUs that contain intra blocks (DecLibRecon.cpp:876-911)
  {
    // ticket + one flag per unit: a picture with more units than the lane's buffer holds gets a larger one; work queued on the lane may
    // still use the old buffer, so the lane is drained first (rare: see vvr_create)
    const size_t need = q->intraLeaf ? 0 : intra_sync_ints( q->numActive, q->numIntra );
    if( need > c->syncCap[lane] )
    {
      HIPCHK( c, hipStreamSynchronize( s ) );
      int* p = nullptr;
      HIPCHK( c, hipMalloc( (void**) &p, sizeof( int ) * need * 2 ) );
      hipFree( c->syncBuf[lane] );
      c->syncBuf[lane] = p; c->syncCap[lane] = need * 2;
    }
  }
  const int wideIntra = h.slice_type == 2 && ( lane == c->prioLane || c->numLanesRR == 1 );      // an I picture the pictures behind it wait for (launch_intra)
  // A picture with scattered intra blocks: one wavefront per block, luma, the scaled residuals of inter chroma blocks and chroma in ONE launch (vvr_intra_leaf.inc)
  if( q->intraLeaf ) { if( q->numIntra ) timed( K_INTRA_LEAF, [&]{ launch_intra_leaf( s, q->pic, P, R, q->intraItems, q->numIntra, q->resiItems, q->numResi, c->leafMaps[lane], c->leafMapInts, c->leafW4, c->leafH4, job.errWord ); } ); }
  // A picture whose inter blocks carry scaled chroma residuals (LMCS): luma units, the residual-add blocks, chroma units
  else if( q->numResi )
  {
    if( q->numLumaUnits ) timedOn( K_INTRA, s, q->bytesIntraLuma, [&]{ launch_intra( s, q->pic, P, R, q->intraItems, q->numIntra, q->units, q->numActive, 0, q->numLumaUnits, q->intraWorkgroups, c->syncBuf[lane], wideIntra ); } );
    timed( K_RESI_ADD, [&]{ launch_resi_add( s, q->pic, P, R, q->resiItems, q->numResi ); } );
    if( q->numActive > q->numLumaUnits ) timedOn( K_INTRA, s, q->bytes[K_INTRA] - q->bytesIntraLuma, [&]{ launch_intra( s, q->pic, P, R, q->intraItems, q->numIntra, q->units, q->numActive, q->numLumaUnits, q->numActive, q->intraWorkgroupsChroma, c->syncBuf[lane], wideIntra ); } );
  }
  else if( q->numActive )
  {
    const bool fine = q->intraFine && c->intraFine;
    timed( K_INTRA, [&]{ launch_intra( s, q->pic, P, R, q->intraItems, q->numIntra, q->units, q->numActive, 0, q->numActive, q->intraWorkgroups, c->syncBuf[lane], wideIntra, fine ? c->leafMaps[lane] : nullptr, c->leafMapInts, c->leafW4, c->leafH4, job.errWord ); } );
  }
  // LMCS: inverse luma mapping of the reconstructed picture (RSP state, DecLibRecon.cpp:935)
  if( lmcsOn && !hop ) timed( K_LMCS, [&]{ launch_lmcs( s, q->pic, P, 1 ); } );
  // in-loop filters: LF_V, LF_H, SAO, ALF (DecLibRecon.cpp:943-1100)
  if( hop )
  {
    timed( K_DEBLOCK_V, [&]{ launch_deblock_tile( s, q->pic, A, B, 0, lmcsOn ); } );
    timed( K_DEBLOCK_H, [&]{ launch_deblock_tile( s, q->pic, B, R, 1, false ); } );
  }
  else if( dbOn )
  {
    timedOn( K_DEBLOCK4, s, q->bytes[K_DEBLOCK_V], [&]{ launch_deblock( s, q->pic, P, 0 ); } );
    timedOn( K_DEBLOCK4, s, q->bytes[K_DEBLOCK_H], [&]{ launch_deblock( s, q->pic, P, 1 ); } );
  }
  if( fused ) timed( K_ALF, [&]{ launch_sao_alf( s, q->pic, hop ? R : B, A, sao, alf ); } );
  else if( sao && alf ) { timed( K_SAO, [&]{ launch_sao( s, q->pic, A, B ); } ); timedOn( K_ALF_PLANES, s, q->bytes[K_ALF], [&]{ launch_alf( s, q->pic, B, A ); } ); }
  else if( sao )   { timed( K_SAO, [&]{ launch_sao( s, q->pic, A, B ); } ); timed( K_COPY, [&]{ launch_copy_planes( s, B, A ); } ); }
  else if( alf )   { timed( K_COPY, [&]{ launch_copy_planes( s, A, B ); } ); timedOn( K_ALF_PLANES, s, q->bytes[K_ALF], [&]{ launch_alf( s, q->pic, B, A ); } ); }
#ifdef VVR_WATCHDOG
  const double wdD = wdNow(); g_wdPart[2] += wdD - wdC; g_wdPartMax[2] = std::max( g_wdPartMax[2], wdD - wdC );
#endif
#ifdef VVR_WATCHDOG
  if( job.tlB ) hipEventRecord( job.tlB, s );
#endif
  hipError_t le = hipGetLastError();
  if( le == hipSuccess ) le = hipEventRecord( job.done, s );
  if( le == hipSuccess ) le = hipEventRecord( job.doneHost, s );
#ifdef VVR_WATCHDOG
  g_wdPart[3] += wdNow() - wdD; if( job.ring ) g_wdPartN++;
#endif
  if( le != hipSuccess )
  {
    // nothing may keep running behind a failed submission: drain the lane
    hipStreamSynchronize( s );
    err = std::string( "kernel launch: " ) + hipGetErrorString( le );
    return VVR_ERR_DEVICE;
  }
  return VVR_OK;
#undef HIPCHK
#define HIPCHK( ctx, call ) do { hipError_t e_ = ( call ); if( e_ != hipSuccess ) { ( ctx )->setError( std::string( #call ) + ": " + hipGetErrorString( e_ ) ); return VVR_ERR_DEVICE; } } while( 0 )
}

// commit every job that is next in submission order and ready.  One thread at a time (commitMu): the launcher thread of a context with worker
// threads, else the thread inside vvr_submit / vvr_submit_prepared.  mu is only held around the bookkeeping, never across a HIP call.
// The next picture to enqueue on the device.  Pictures are committed in submission order - except that a picture whose work lists are ready may pass
// pictures whose host stage is still running if it has nothing to do with them.  Two cases matter: the pictures behind an I picture (the longest
// host stage of the stream, and a host that parses ahead hands it over early) need not wait for its work lists; and an I picture, which depends on
// nothing and whose 8 ms intra stage everything of the next GOP waits for, need not wait for the work lists of the B pictures submitted before it
// (workers finish in any order; measured on 20 4K pictures arriving at once: the I picture was ready at 5.9 ms and enqueued at 9.3 ms).  "Nothing to
// do with them" is decided from the picture headers alone (slots known at submission): the later picture reads no slot an overtaken picture writes,
// writes no slot an overtaken picture reads or writes - transitively, since a picture that may not pass joins the overtaken ones.  mu held.
static const vvr_pic_header& hdrOfJob( const Job& j ) { return j.q ? j.q->hdr : j.pic.hdr; }
static bool slotConflict( const vvr_pic_header& later, const vvr_pic_header& earlier )
{
  if( later.out_slot == earlier.out_slot ) return true;
  if( later.slice_type != 2 ) for( int l = 0; l < 2; l++ ) for( int i = 0; i < later.num_ref[l]; i++ ) if( later.ref_slot[l][i] == earlier.out_slot ) return true;
  if( earlier.slice_type != 2 ) for( int l = 0; l < 2; l++ ) for( int i = 0; i < earlier.num_ref[l]; i++ ) if( earlier.ref_slot[l][i] == later.out_slot ) return true;
  return false;
}
static Job* nextToCommitLocked( vvr_context* c )
{
  Job* overtaken[24]; int n = 0;
  for( uint64_t seq = c->nextCommit; seq < c->nextCommit + 24; seq++ )
  {
    auto it = c->bySeq.find( seq );
    if( it == c->bySeq.end() ) break;               // not submitted yet
    Job* j = it->second;
    if( j->handled ) continue;
    const bool ready = j->state == J_READY || j->state == J_FAILED;
    if( ready )
    {
      bool free = true;
      for( int k = 0; k < n && free; k++ ) free = !slotConflict( hdrOfJob( *j ), hdrOfJob( *overtaken[k] ) );
      if( free ) return j;
    }
    overtaken[n++] = j;
  }
  return nullptr;
}

static void commitReady( vvr_context* c )
{
  std::lock_guard<std::mutex> cm( c->commitMu );
  CommitPlan plan;
  for( ;; )
  {
    Job* j = nullptr;
    {
      std::lock_guard<std::mutex> lk( c->mu );
      j = nextToCommitLocked( c );
      if( !j ) break;
      if( j->seq != c->nextCommit ) c->overtakes++;
      if( j->state == J_READY ) planCommitLocked( c, *j, plan );
    }
    int rc = VVR_OK; std::string err;
#ifdef VVR_WATCHDOG
    const auto wdT0 = std::chrono::steady_clock::now();
    WD_STAMP( *j, tCommit0 );
#endif
    if( j->state == J_READY ) rc = enqueuePicture( c, *j, plan, err );
#ifdef VVR_WATCHDOG
    { const double ms = std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - wdT0 ).count(); g_wdEnqMs = g_wdEnqMs + ms; g_wdEnqMax = std::max( g_wdEnqMax, ms ); g_wdEnqN++; WD_STAMP( *j, tCommit1 );
      if( j->ring ) { g_wdSum[0] += j->tPrep - j->tSubmit; g_wdSum[1] += j->tBuilt - j->tPrep; g_wdSum[2] += j->tRing - j->tBuilt; g_wdSum[3] += j->tReady - j->tRing; g_wdSum[4] += j->tCommit0 - j->tReady; g_wdSum[5] += j->tCommit1 - j->tCommit0; g_wdJobs++; }
      // developer build: one line per picture with the times of its host stages (VVR_TIMELINE)
      static const bool tl = getenv( "VVR_TIMELINE" ) != nullptr; const double tl0 = c->tl0;
      if( tl && j->ring ) {
        fprintf( stderr, "[vvr timeline] seq %3llu poc %4d type %d: submit %6.2f prepare %6.2f built %6.2f ring %6.2f ready %6.2f commit %6.2f..%6.2f\n", (unsigned long long) j->seq, j->pic.hdr.poc, j->pic.hdr.slice_type,
                 j->tSubmit - tl0, j->tPrep - tl0, j->tBuilt - tl0, j->tRing - tl0, j->tReady - tl0, j->tCommit0 - tl0, j->tCommit1 - tl0 ); } }
#endif
    {
      std::lock_guard<std::mutex> lk( c->mu );
      if( j->state == J_READY )
      {
        if( rc == VVR_OK )
        {
          const vvr_pic_header& h = j->q->hdr;
          c->slotUsers[h.out_slot].clear(); c->slotUsers[h.out_slot].push_back( j->id );
          { auto& v = c->slotExt[h.out_slot]; for( hipEvent_t ev : plan.outExt ) v.erase( std::remove( v.begin(), v.end(), ev ), v.end() ); }      // (what this writer waited for is ordered before it now)
          if( h.slice_type != 2 ) for( int l = 0; l < 2; l++ ) for( int i = 0; i < h.num_ref[l]; i++ ) c->slotUsers[h.ref_slot[l][i]].push_back( j->id );
          j->state = J_COMMITTED;
        }
        else
        {
          for( auto& t : j->timings ) { hipEventDestroy( t.a ); hipEventDestroy( t.b ); }
          j->timings.clear();
          if( j->done ) { c->eventPool.push_back( j->done ); j->done = nullptr; }
          if( j->doneHost ) { c->eventPool.push_back( j->doneHost ); j->doneHost = nullptr; }
          j->state = J_FAILED; j->rc = rc; j->err = err; c->setError( err );
        }
      }
      if( j->state == J_FAILED )
      {
        // a failed picture holds nothing: its ring entry is free again, waiting for it returns the error
        if( j->ring && j->ring->owner == j ) { j->ring->owner = nullptr; j->ring->turn += c->ring.size(); }
        j->q = nullptr; j->completed = true;
      }
      j->handled = true;
      for( auto it = c->bySeq.find( c->nextCommit ); it != c->bySeq.end() && it->second->handled; it = c->bySeq.find( c->nextCommit ) ) { c->bySeq.erase( it ); c->nextCommit++; }
      c->cv.notify_all();
    }
  }
}

// the committing thread of a context with worker threads
static void launcherMain( vvr_context* c )
{
  hipSetDevice( c->device );
  pinToCpus( c->nodeCpus );
  for( ;; )
  {
    {
      std::unique_lock<std::mutex> lk( c->mu );
      c->cv.wait( lk, [&]{ return c->stop || nextToCommitLocked( c ) != nullptr; } );
      if( c->stop ) break;
    }
    commitReady( c );
  }
}

// stage 1 of a streaming job: work lists into scratch, then packed into the job's ring entry.  Called without mu.
#ifdef VVT_SLOW_I_PICTURES
static int g_vvtSlowIUs = 0;       // stand-in runtime (tests): extra time the host stage of an I picture other than the first of the stream takes
static int g_vvtSlowBUs = 0;       // ... and of every picture that is not an I picture
#endif
static void prepareJob( vvr_context* c, Job& job, PrepScratch& S, HostHelpers* helpers = nullptr )
{
#ifdef VVT_SLOW_I_PICTURES
  if( g_vvtSlowIUs && job.pic.hdr.slice_type == 2 && job.pic.hdr.poc != 0 ) std::this_thread::sleep_for( std::chrono::microseconds( g_vvtSlowIUs ) );
  if( g_vvtSlowBUs && job.pic.hdr.slice_type != 2 ) std::this_thread::sleep_for( std::chrono::microseconds( g_vvtSlowBUs ) );
#endif
  size_t total = 0; std::string err;
  // A picture with inter CUs is normally one worker's job (pictures side by side: no joining, no waiting for the slowest band).  While the device has
  // next to nothing to do - the first pictures of a burst, a stream that is submitted picture by picture - the workers build it together, in bands
  // (vvr_host_build): what counts then is when the picture reaches the device, 4.5 ms for a 4K B picture on one thread.
  if( helpers )
  {
    int policy = c->partsPolicy;
    bool parts = policy == 1;
    if( policy == 2 )
    {
      std::lock_guard<std::mutex> lk( c->mu );
      int ahead = 0;        // pictures the device has or is about to get
      for( auto& kv : c->jobs ) { const Job& j = *kv.second; if( !j.completed && j.seq < job.seq && ( j.state == J_COMMITTED || j.state == J_READY ) ) ahead++; }
      parts = ahead < c->numLanesRR;
    }
    vvr_scratch_parts_for_all( &S, parts );
  }
  WD_STAMP( job, tPrep );
  // (the CU / TU records are checked here - on the way, by whoever builds the part - unless there are no workers: then vvr_submit has checked them)
#ifdef VVR_DEV_ENV
  // developer build: where the host stage of a picture spends its time (VVR_PHASES; vvr_host_build prints its own phases)
  auto devNow = []{ return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count(); };
  const bool devPhases = getenv( "VVR_PHASES" ) != nullptr;
  double devT[4] = { devNow(), 0, 0, 0 };
#endif
  int rc = vvr_host_build( &job.pic, S, &total, err, &c->pinned, helpers, !c->workers.empty() );
#ifdef VVR_DEV_ENV
  devT[1] = devNow();
#endif
  WD_STAMP( job, tBuilt );
  RingEntry& e = c->ring[job.ringSeq % c->ring.size()];
  {
    // The ring entry is ours when it is this job's turn: the streaming job ring.size() places ahead of us has used it and its picture is
    // reconstructed (that job is committed before us and never waits for us).  Strictly by turn - a later job that finds the entry free
    // must not take it, or the job whose turn it is could wait for it forever while everything behind waits for that job's commit.  A job
    // whose work lists could not be built takes its turn all the same (and gives the entry back when its place in the commit order comes).
    std::unique_lock<std::mutex> lk( c->mu );
    for( ;; )
    {
      if( e.owner )
      {
        Job* prev = e.owner;
        c->cv.wait( lk, [&]{ return e.owner != prev || prev->state == J_COMMITTED || prev->completed; } );
        if( e.owner != prev || prev->completed ) continue;
        hipEvent_t ev = prev->doneHost;
        lk.unlock();
        hipEventSynchronize( ev );
        lk.lock();
        if( e.owner == prev ) completeLocked( c, *prev );
        continue;
      }
      if( e.turn == job.ringSeq ) break;
      c->cv.wait( lk, [&]{ return e.owner != nullptr || e.turn == job.ringSeq; } );
    }
    e.owner = &job; job.ring = &e;
  }
  WD_STAMP( job, tRing );
  if( rc == VVR_OK )
  {
    // Buffers only ever grow, and an entry that has to grow takes the size of the largest picture seen so far (an intra picture needs several
    // times the bytes of a B picture: after the first one has come by, no entry is reallocated when the next one lands on it).  The old
    // buffers are kept until the context goes: hipFree would wait for the device, i.e. for every picture in flight.
    size_t want, wantHost;
    const size_t staged = vvr_host_staged_bytes( S );      // (what the device writes itself - MC tiles, the edge-parameter tables - has no pinned side)
    {
      std::lock_guard<std::mutex> lk( c->mu );
      c->ringLargest = std::max( c->ringLargest, alignUp( total + total / 4, 1 << 16 ) ); want = c->ringLargest;
      c->ringLargestHost = std::max( c->ringLargestHost, alignUp( staged + staged / 4, 1 << 16 ) ); wantHost = c->ringLargestHost;
    }
    if( staged > e.hostCap )
    {
      if( e.host ) { std::lock_guard<std::mutex> lk( c->mu ); c->retiredHost.push_back( e.host ); }
      e.host = nullptr; e.hostCap = 0;
      if( hipHostMalloc( (void**) &e.host, wantHost, hipHostMallocDefault ) != hipSuccess ) { rc = VVR_ERR_DEVICE; err = "hipHostMalloc failed"; } else e.hostCap = wantHost;
    }
    if( rc == VVR_OK && total > e.devCap )
    {
      if( e.dev ) { std::lock_guard<std::mutex> lk( c->mu ); c->retiredDev.push_back( e.dev ); }
      e.dev = nullptr; e.devCap = 0;
      if( hipMalloc( (void**) &e.dev, want ) != hipSuccess ) { rc = VVR_ERR_DEVICE; err = "hipMalloc failed"; } else e.devCap = want;
    }
  }
  if( rc == VVR_OK )
  {
#ifdef VVR_DEV_ENV
    devT[2] = devNow();
#endif
    vvr_host_pack( S, e.host );
#ifdef VVR_DEV_ENV
    devT[3] = devNow();
    if( devPhases ) fprintf( stderr, "[vvr] host stage (ms): build %.2f, ring %.2f, pack %.2f (%.2f MB staged)\n", devT[1] - devT[0], devT[2] - devT[1], devT[3] - devT[2], vvr_host_staged_bytes( S ) / 1e6 );
#endif
    vvr_host_bind( S, e.q, e.dev );
    vvr_host_upload_plan( S, e.direct, &e.stagedBegin, &e.stagedEnd );
    e.q.colHost = nullptr; e.q.numCol = 0; e.q.pic.colMotion = nullptr; e.q.pic.colStride = 0;
    if( job.pic.hdr.tool_flags & VVR_TOOL_COL_MOTION )
    {
      // collocated motion: the unrefined subsample is written here by this thread, the DMVR kernel replaces the MVs it refines
      const size_t n = vvr_host_num_col( &job.pic );
      if( n > e.colCap )
      {
        if( e.colHost ) { std::lock_guard<std::mutex> lk( c->mu ); c->retiredHost.push_back( (char*) e.colHost ); }
        e.colHost = nullptr; e.colCap = 0;
        const size_t cap = std::max( n, ( ( (size_t) c->cfg.max_width + 7 ) >> 3 ) * ( ( (size_t) c->cfg.max_height + 7 ) >> 3 ) );
        if( hipHostMalloc( (void**) &e.colHost, sizeof( vvr_motion ) * cap, hipHostMallocDefault ) != hipSuccess ) { rc = VVR_ERR_DEVICE; err = "hipHostMalloc failed"; } else e.colCap = cap;
      }
      if( rc == VVR_OK )
      {
        vvr_host_gather_col( &job.pic, e.colHost );
        e.q.colHost = e.colHost; e.q.numCol = n; e.q.pic.colMotion = e.colHost; e.q.pic.colStride = (int) ( ( ( ( job.pic.hdr.width + 3 ) >> 2 ) + 1 ) >> 1 );
      }
    }
    const size_t nInts = 2 * (size_t) e.q.numDmvr;
    if( rc == VVR_OK && nInts > e.dmvrCap )
    {
      if( e.dmvrHost ) { std::lock_guard<std::mutex> lk( c->mu ); c->retiredHost.push_back( (char*) e.dmvrHost ); }
      e.dmvrHost = nullptr; e.dmvrCap = 0;
      const size_t cap = std::max( nInts, 2 * ( (size_t) c->cfg.max_width * c->cfg.max_height / 128 + 1 ) );      // (no picture of this size has more DMVR sub-blocks)
      if( hipHostMalloc( (void**) &e.dmvrHost, sizeof( int32_t ) * cap, hipHostMallocDefault ) != hipSuccess ) { rc = VVR_ERR_DEVICE; err = "hipHostMalloc failed"; } else e.dmvrCap = cap;
    }
  }
  {
    std::lock_guard<std::mutex> lk( c->mu );
    WD_STAMP( job, tReady );
    if( rc == VVR_OK ) { job.q = &e.q; job.ring = &e; job.state = J_READY; }
    else { job.rc = rc; job.err = err; job.state = J_FAILED; }
    c->cv.notify_all();                                 // (the launcher, if there is one)
  }
#ifdef VVR_DEV_ENV
  const double devC0 = devNow();
#endif
  if( c->workers.empty() ) commitReady( c );            // no worker threads: the submitting thread commits
#ifdef VVR_DEV_ENV
  if( devPhases && c->workers.empty() ) fprintf( stderr, "[vvr] host stage (ms): commit %.2f, after pack %.2f\n", devNow() - devC0, devC0 - devT[3] );
#endif
}

#ifdef VVR_WATCHDOG
// developer build (make watchdog): when nothing completes for 15 s, the state of the pipeline goes to stderr and the process aborts
static void watchdogMain( vvr_context* c )
{
  uint64_t last = g_wdProgress.load(); int idle = 0;
  while( !c->stop )
  {
    std::this_thread::sleep_for( std::chrono::seconds( 1 ) );
    const uint64_t now = g_wdProgress.load();
    bool pending = false;
    if( c->mu.try_lock() ) { for( auto& kv : c->jobs ) if( !kv.second->completed ) pending = true; c->mu.unlock(); } else pending = true;
    if( now != last || !pending ) { last = now; idle = 0; continue; }
    if( ++idle < 15 ) continue;
    fprintf( stderr, "[vvr watchdog] no progress for 15 s; mu %s, commitMu %s\n", c->mu.try_lock() ? ( c->mu.unlock(), "free" ) : "HELD", c->commitMu.try_lock() ? ( c->commitMu.unlock(), "free" ) : "HELD" );
    fprintf( stderr, "  nextSeq %llu nextCommit %llu queue %zu workers %zu\n", (unsigned long long) c->nextSeq, (unsigned long long) c->nextCommit, c->queue.size(), c->workers.size() );
    for( auto& kv : c->jobs ) { Job& j = *kv.second; if( j.completed && j.waited ) continue; fprintf( stderr, "  job %d seq %llu state %d completed %d waited %d lane %d ring %d done %p\n", j.id, (unsigned long long) j.seq, j.state, (int) j.completed, (int) j.waited, j.lane, j.ring ? (int) ( j.ring - c->ring.data() ) : -1, (void*) j.done ); }
    for( size_t i = 0; i < c->ring.size(); i++ ) fprintf( stderr, "  ring %zu owner %d\n", i, c->ring[i].owner ? c->ring[i].owner->id : -1 );
    for( size_t i = 0; i < c->streams.size(); i++ ) fprintf( stderr, "  stream %zu query %d\n", i, (int) hipStreamQuery( c->streams[i] ) );
    fprintf( stderr, "  copyStream query %d\n", (int) hipStreamQuery( c->copyStream ) );
    fflush( stderr );
    abort();
  }
}
#endif

// The workers as helpers of the one among them that prepares an I picture (HostHelpers): the parts go into `subtasks`, which every worker serves before it
// takes another picture; the caller runs part 0, then whatever parts are still unclaimed (on a second scratch of its own: its first one holds part 0),
// and returns when all parts are done.  A part may block until the parts before it are done (they are appended in order): parts are claimed in order, and
// the caller claims too, so the earliest part not yet done is always running or about to.
struct WorkerHelpers : HostHelpers
{
  vvr_context* c; PrepScratch*& spare;
  bool announced = false;               // this worker counts in c->partsComing (it took an I picture and has not published the parts yet)
  WorkerHelpers( vvr_context* c_, PrepScratch*& spare_ ) : c( c_ ), spare( spare_ ) {}
  int width() const override { return c->cfg.host_threads; }
  void notInParts() override { if( announced ) { std::lock_guard<std::mutex> lk( c->mu ); c->partsComing--; announced = false; c->cv.notify_all(); } }
  void run( int n, PrepScratch& own, const std::function<void( int, PrepScratch& )>& fn ) override
  {
    struct State { std::mutex mu; std::condition_variable cv; int remaining; } st; st.remaining = n - 1;
    {
      std::lock_guard<std::mutex> lk( c->mu );
      // (the parts of an I picture go IN FRONT of the parts of B pictures that are waiting there: everything of the next GOP hangs on the I picture's chain on the device)
      const bool first = announced;
      for( int part = first ? n - 1 : 1; first ? part >= 1 : part < n; part += first ? -1 : 1 )
      {
        auto task = [&st, &fn, part]( PrepScratch& R ) { fn( part, R ); std::lock_guard<std::mutex> l2( st.mu ); st.remaining--; st.cv.notify_all(); };      // (notified under the lock: `st` lives on the caller's stack and is gone once the caller has seen remaining == 0)
        if( first ) c->subtasks.push_front( task ); else c->subtasks.push_back( task );
      }
      if( announced ) { c->partsComing--; announced = false; }
      c->cv.notify_all();
    }
    fn( 0, own );
    for( ;; )
    {
      std::function<void( PrepScratch& )> task;
      { std::lock_guard<std::mutex> lk( c->mu ); if( !c->subtasks.empty() ) { task = std::move( c->subtasks.front() ); c->subtasks.pop_front(); } }
      if( !task ) break;
      if( !spare ) { spare = vvr_scratch_create(); vvr_scratch_intra_leaf( spare, c->intraLeaf, c->leafByLevel ); vvr_scratch_warm( spare, c->cfg ); }
      task( *spare );
    }
    std::unique_lock<std::mutex> lk( st.mu );
    st.cv.wait( lk, [&]{ return st.remaining == 0; } );
  }
};

static void workerMain( vvr_context* c )
{
  hipSetDevice( c->device );
  pinToCpus( c->nodeCpus );
  PrepScratch* S = vvr_scratch_create();
  vvr_scratch_intra_leaf( S, c->intraLeaf, c->leafByLevel );
  vvr_scratch_warm( S, c->cfg );
  // (a second scratch for the parts of its own I picture that nobody else takes, see WorkerHelpers::run: allocated and touched now, not inside a picture)
  PrepScratch* spare = nullptr;
  if( c->cfg.host_threads > 1 ) { spare = vvr_scratch_create(); vvr_scratch_intra_leaf( spare, c->intraLeaf, c->leafByLevel ); vvr_scratch_warm( spare, c->cfg ); }
  WorkerHelpers helpers( c, spare );
  for( ;; )
  {
    Job* job = nullptr;
    {
      std::unique_lock<std::mutex> lk( c->mu );
      // (while a worker is about to publish the parts of an I picture - a fraction of a millisecond, the scan of its CUs - the others do not start on
      // another picture: the I picture's chain on the device is the longest thing in the stream, and a worker that has just taken a B picture would be
      // busy with it for the 2 ms in which the I picture needs everybody.  Measured on the driver's 20-picture window: the IRAP built 2.3 ms after it was
      // taken, by its own worker alone, because the other seven had each taken a B picture 0.05 ms before its parts appeared)
      c->cv.wait( lk, [&]{ return c->stop || !c->subtasks.empty() || ( !c->queue.empty() && c->partsComing == 0 ); } );
      if( !c->subtasks.empty() )
      {
        // a part of a picture another worker is preparing: before anything else (that picture is an I picture: the next GOP waits for it)
        std::function<void( PrepScratch& )> task = std::move( c->subtasks.front() ); c->subtasks.pop_front();
        lk.unlock();
        task( *S );
        continue;
      }
      if( c->queue.empty() ) break;       // (stop, and nothing left to do)
      // An I picture among the next few waiting pictures goes first: its intra stage is the longest thing the device does for the stream (8 ms at
      // 4K against 0.7 ms for a whole B picture), everything of the next GOP waits for it, and it waits for nothing itself, so it should not queue
      // behind the host stage of pictures that do not depend on it.  Only as far ahead as the upload ring reaches: the
      // ring entry of a picture that far down is free as soon as pictures already handed to workers are done, never one still in this queue.
      size_t pick = 0;
      for( size_t k = 1; k < c->queue.size() && k < 16; k++ )
      {
        Job* cand = c->queue[k];
        if( cand->pic.hdr.slice_type != 2 ) continue;
        if( cand->ringSeq - c->queue.front()->ringSeq < c->ring.size() && c->queue.front()->pic.hdr.slice_type != 2 ) pick = k;
        break;
      }
      job = c->queue[pick]; c->queue.erase( c->queue.begin() + pick );
      job->state = J_PREPARING;
      if( job->pic.hdr.slice_type == 2 && c->cfg.host_threads > 1 ) { c->partsComing++; helpers.announced = true; }
      c->cv.notify_all();                 // (a submitter may be waiting for room in the queue)
    }
    prepareJob( c, *job, *S, c->cfg.host_threads > 1 ? &helpers : nullptr );
    if( helpers.announced ) { std::lock_guard<std::mutex> lk( c->mu ); c->partsComing--; helpers.announced = false; c->cv.notify_all(); }      // (the picture was not built in parts after all)
  }
  vvr_scratch_destroy( S );
  if( spare ) vvr_scratch_destroy( spare );
}

// wait until the job's picture is reconstructed (or has failed); returns its status
static int finishJob( vvr_context* c, int id )
{
  std::unique_lock<std::mutex> lk( c->mu );
  auto it = c->jobs.find( id );
  if( it == c->jobs.end() ) return VVR_OK;        // already retired
  Job& j = *it->second;
  c->cv.wait( lk, [&]{ return j.state == J_COMMITTED || j.completed; } );
  if( !j.completed )
  {
    hipEvent_t ev = j.doneHost;
    lk.unlock();
    const hipError_t e = hipEventSynchronize( ev );
    lk.lock();
    if( e != hipSuccess ) { c->setError( std::string( "hipEventSynchronize: " ) + hipGetErrorString( e ) ); return VVR_ERR_DEVICE; }
    // mu was released: another thread may have finished, waited for and - through a later vvr_submit - retired the job meanwhile
    it = c->jobs.find( id );
    if( it == c->jobs.end() ) return VVR_OK;
    Job& jj = *it->second;
    if( !jj.completed ) completeLocked( c, jj );
    jj.waited = true;
    if( jj.state == J_FAILED ) { c->setError( jj.err ); return jj.rc; }
    return VVR_OK;
  }
  j.waited = true;
  if( j.state == J_FAILED ) { c->setError( j.err ); return j.rc; }
  return VVR_OK;
}

// forget jobs nobody can ask about any more.  mu held.
static void retireLocked( vvr_context* c )
{
  if( c->jobs.size() <= 256 ) return;
  for( auto it = c->jobs.begin(); it != c->jobs.end() && c->jobs.size() > 128; )
  {
    if( it->second->completed && it->second->waited ) it = c->jobs.erase( it ); else ++it;
  }
}

static Job* newJobLocked( vvr_context* c )
{
  std::unique_ptr<Job> j( new Job() );
  j->id = c->nextJob++; j->seq = c->nextSeq++;
  Job* p = j.get();
  c->bySeq[p->seq] = p;
  c->jobs[p->id] = std::move( j );
  return p;
}

extern "C" {

#define VVR_STR2( x ) #x
#define VVR_STR( x ) VVR_STR2( x )
VVR_API const char* vvr_version( void ) { return "vvdec_amd 0.3 (gfx950, ABI " VVR_STR( VVR_ABI_VERSION ) ")"; }

VVR_API size_t vvr_abi_sizeof( int which )
{
  static const size_t sz[] = { sizeof( vvr_pic_header ), sizeof( vvr_cu ), sizeof( vvr_tu ), sizeof( vvr_motion ), sizeof( vvr_lfp ), sizeof( vvr_sao_ctu ),
                               sizeof( vvr_alf_ctu ), sizeof( vvr_alf_params ), sizeof( vvr_lmcs_params ), sizeof( vvr_picture ), sizeof( vvr_config ), sizeof( vvr_kernel_stat ),
                               sizeof( vvr_wp_params ), sizeof( vvr_scaling_list ), sizeof( vvr_subpic ), sizeof( vvr_slice_header ),
                               sizeof( vvr_rpr_ref ), sizeof( vvr_rpr_params ) };
  return which >= 0 && which < (int) ( sizeof( sz ) / sizeof( sz[0] ) ) ? sz[which] : 0;
}

VVR_API size_t vvr_slot_bytes( const vvr_config* cfg )
{
  int st[3]; size_t b[3], t; planeGeometry( cfg, st, b, &t ); return t;
}

VVR_API void vvr_destroy( vvr_context* c );

VVR_API int vvr_create( const vvr_config* cfg, vvr_context** out )
{
  if( !cfg || !out || cfg->abi_version != VVR_ABI_VERSION ) return VVR_ERR_PARAMETER;
  // Main 10: 4:0:0 / 4:2:0, 8..10-bit samples (the formats the parity tests cover); CTU 32..128
  if( cfg->chroma_format > 1 || cfg->bit_depth < 8 || cfg->bit_depth > 10 || cfg->log2_ctu < 5 || cfg->log2_ctu > 7 || !cfg->num_slots || cfg->host_threads > 64 || cfg->stop_after > VVR_STOP_SAO ) return VVR_ERR_UNSUPPORTED;
  int ndev = 0;
  if( hipGetDeviceCount( &ndev ) != hipSuccess || ndev <= 0 || cfg->device >= ndev ) return VVR_ERR_NO_DEVICE;
  if( hipSetDevice( cfg->device ) != hipSuccess ) return VVR_ERR_NO_DEVICE;
  {
    hipDeviceProp_t prop;
    if( hipGetDeviceProperties( &prop, cfg->device ) != hipSuccess ) return VVR_ERR_NO_DEVICE;
    if( strncmp( prop.gcnArchName, "gfx950", 6 ) != 0 ) { fprintf( stderr, "vvdec_amd: device %d is %s, this library is built for gfx950 only\n", cfg->device, prop.gcnArchName ); return VVR_ERR_NO_DEVICE; }
  }
  if( vvr_upload_tables() != 0 ) return VVR_ERR_DEVICE;
  vvr_context* c = new vvr_context();
  c->cfg = *cfg; c->device = cfg->device;
  const int ns = std::max<int>( 1, cfg->num_streams );
  // lanes: num_streams of them taken in turn, plus - whenever there are at least two - one with a high-priority stream for I pictures, see planCommitLocked
  // (also without worker threads: vvr_submit_prepared and inline submission order pictures the same way)
  const int nl = ns + ( ns >= 2 ? 1 : 0 );
  c->numLanesRR = ns; c->prioLane = nl > ns ? ns : -1;
  if( const char* e = getenv( "VVR_PARTS" ) ) c->partsPolicy = atoi( e );      // 0 / 1 / 2: see partsPolicy
  if( const char* e = getenv( "VVR_INTRA_LEAF" ) ) c->intraLeaf = atoi( e ) != 0;
#ifdef VVR_DEV_ENV
  // two experiments of round 5, measured and left off: switchable in the developer build only
  if( const char* e = getenv( "VVR_LEAF_BY_LEVEL" ) ) c->leafByLevel = atoi( e ) != 0;
  if( const char* e = getenv( "VVR_INTRA_FINE" ) ) c->intraFine = atoi( e ) != 0;
#endif
  c->streams.resize( nl, nullptr );
  bool ok = true;
  for( int i = 0; i < ns && ok; i++ ) ok = hipStreamCreateWithFlags( &c->streams[i], hipStreamNonBlocking ) == hipSuccess;
  if( ok && c->prioLane >= 0 )
  {
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange( &least, &greatest );                    // (numerically lower = higher priority)
    ok = hipStreamCreateWithPriority( &c->streams[c->prioLane], hipStreamNonBlocking, greatest ) == hipSuccess;
  }
  ok = ok && hipStreamCreateWithFlags( &c->copyStream, hipStreamNonBlocking ) == hipSuccess;
  planeGeometry( cfg, c->stride, c->planeBytes, &c->slotBytes );
  if( ok )
  {
    if( cfg->ext_planes ) c->planeMem = cfg->ext_planes;
    else { ok = hipMalloc( &c->planeMem, c->slotBytes * cfg->num_slots ) == hipSuccess; if( ok ) { c->planeMemOwned = true; hipMemset( c->planeMem, 0, c->slotBytes * cfg->num_slots ); } }
  }
  ok = ok && hipMalloc( &c->scratchMem, c->slotBytes * 2 * nl ) == hipSuccess;
  if( ok )
  {
    c->slotDim.assign( cfg->num_slots, std::make_pair( cfg->max_width, cfg->max_height ) );
    for( int k = 0; k < std::min<int>( cfg->read_buffers, 8 ) && ok; k++ ) { void* p = nullptr; ok = hipHostMalloc( &p, c->slotBytes, hipHostMallocDefault ) == hipSuccess; if( ok ) c->stagePool.push_back( p ); }
    for( int s = 0; s < cfg->num_slots; s++ ) c->slots.push_back( carve( (char*) c->planeMem + c->slotBytes * s, cfg, c->stride, c->planeBytes ) );
    for( int s = 0; s < nl; s++ )
    {
      c->scratchB.push_back( carve( (char*) c->scratchMem + c->slotBytes * ( 2 * s ), cfg, c->stride, c->planeBytes ) );
      c->scratchR.push_back( carve( (char*) c->scratchMem + c->slotBytes * ( 2 * s + 1 ), cfg, c->stride, c->planeBytes ) );
    }
    const int ctu = 1 << cfg->log2_ctu;
    const size_t numCtu = (size_t) ( ( cfg->max_width + ctu - 1 ) / ctu ) * ( ( cfg->max_height + ctu - 1 ) / ctu );
    // sized for the usual pictures (a 4K B picture of the benchmark has about 6 units per CTU, an intra picture 3, and about 100 blocks per CTU
    // with a 256-byte parameter record each: 13 MB per lane at 4K); pictures with more units or blocks than that (many isolated small intra
    // CUs) make the lane's buffer grow when they are submitted
    for( int s = 0; s < nl && ok; s++ ) { int* p = nullptr; const size_t cap = intra_sync_ints( (int) ( 24 * numCtu ), (int) ( 100 * numCtu ) ); ok = hipMalloc( (void**) &p, sizeof( int ) * cap ) == hipSuccess; if( ok ) { c->syncBuf.push_back( p ); c->syncCap.push_back( cap ); } }
    // the per-cell words of k_intra_leaf: three maps of the largest picture, a flag and a factor per VPDU; zero from here on (every launch leaves them so)
    {
      c->leafW4 = ( cfg->max_width + 3 ) >> 2; c->leafH4 = ( cfg->max_height + 3 ) >> 2;
      const int vl = std::min<int>( 6, cfg->log2_ctu ), vpdus = ( ( cfg->max_width + ( 1 << vl ) - 1 ) >> vl ) * ( ( cfg->max_height + ( 1 << vl ) - 1 ) >> vl );
      c->leafMapInts = intra_leaf_map_ints( c->leafW4, c->leafH4, vpdus );
      for( int s = 0; s < nl && ok; s++ ) { uint32_t* p = nullptr; ok = hipMalloc( (void**) &p, sizeof( uint32_t ) * c->leafMapInts ) == hipSuccess && hipMemset( p, 0, sizeof( uint32_t ) * c->leafMapInts ) == hipSuccess; if( p ) c->leafMaps.push_back( p ); }
      if( ok ) { ok = hipHostMalloc( (void**) &c->errHost, sizeof( int ) * VVR_ERR_WORDS, hipHostMallocDefault ) == hipSuccess; if( ok ) memset( c->errHost, 0, sizeof( int ) * VVR_ERR_WORDS ); }
    }
  }
  if( ok )
  {
    // upload ring: one entry per picture that can be between "being prepared" and "reconstructed"
    // (a picture holds its entry from the moment a worker starts packing it until the device has finished it: the pictures in the workers' hands,
    // those waiting for their turn to be committed, and those in flight on the device - the lanes and what is queued behind them.  An entry too few
    // makes a worker wait for the device instead of preparing ahead.)
    c->ring.resize( cfg->ring_entries ? std::max<size_t>( cfg->ring_entries, 2 ) : 2 * (size_t) ns + 2 * (size_t) cfg->host_threads + 4 );
    // Every entry starts with room for an ordinary picture of this size (the two edge-parameter tables, CU / TU records and levels of a
    // moderately split picture, the work lists): allocating pinned and device memory takes milliseconds and must not happen while a stream runs.
    // Larger pictures (an intra picture with small CUs) make the entries grow, see prepareJob.
    const size_t w4 = ( cfg->max_width + 3 ) >> 2, h4 = ( cfg->max_height + 3 ) >> 2;
    const size_t estimate = alignUp( w4 * h4 * ( 2 * sizeof( vvr_lfp ) + 14 ) + ( 1u << 20 ), 1 << 16 );
    const size_t dmvrInts = 2 * ( (size_t) cfg->max_width * cfg->max_height / 128 + 1 );
    // (device side: room for the cell maps and the motion of sub-block CUs as well, should the pictures leave the edge parameters to the back-end)
    const size_t estimateDev = estimate + alignUp( w4 * h4 * ( 2 * 16 + sizeof( vvr_motion ) ), 1 << 16 );
    c->ringLargest = estimateDev; c->ringLargestHost = estimate;
    for( size_t i = 0; i < c->ring.size() && ok; i++ )
    {
      RingEntry& e = c->ring[i];
      e.turn = i;
      ok = hipEventCreateWithFlags( &e.copied, hipEventDisableTiming ) == hipSuccess
        && hipHostMalloc( (void**) &e.host, estimate, hipHostMallocDefault ) == hipSuccess && hipMalloc( (void**) &e.dev, estimateDev ) == hipSuccess
        && hipHostMalloc( (void**) &e.dmvrHost, sizeof( int32_t ) * dmvrInts, hipHostMallocDefault ) == hipSuccess;
      if( ok ) { e.hostCap = estimate; e.devCap = estimateDev; e.dmvrCap = dmvrInts; }
    }
  }
  if( !ok ) { vvr_destroy( c ); return VVR_ERR_DEVICE; }
  c->slotUsers.resize( cfg->num_slots ); c->slotExt.resize( cfg->num_slots );
  c->inlineScratch = vvr_scratch_create();
  vvr_scratch_intra_leaf( c->inlineScratch, c->intraLeaf, c->leafByLevel );
  if( cfg->host_threads <= 0 ) vvr_scratch_warm( c->inlineScratch, c->cfg );
  if( cfg->host_threads ) c->nodeCpus = gpuNodeCpus( c->device );
  for( int t = 0; t < cfg->host_threads; t++ ) c->workers.emplace_back( workerMain, c );
  if( cfg->host_threads ) c->launcher = std::thread( launcherMain, c );
#ifdef VVR_WATCHDOG
  c->watchdog = std::thread( watchdogMain, c );
#endif
  *out = c;
  return VVR_OK;
}

VVR_API int vvr_sync( vvr_context* c );

VVR_API void vvr_destroy( vvr_context* c )
{
  if( !c ) return;
  hipSetDevice( c->device );
  if( !c->streams.empty() && c->inlineScratch ) vvr_sync( c );
  { std::lock_guard<std::mutex> lk( c->mu ); c->stop = true; c->cv.notify_all(); }
  for( auto& t : c->workers ) t.join();
  if( c->launcher.joinable() ) c->launcher.join();
#ifdef VVR_WATCHDOG
  if( c->watchdog.joinable() ) c->watchdog.join();
  if( g_wdJobs ) fprintf( stderr, "[vvr] per streamed picture (ms): queued %.2f, work lists %.2f, wait for ring entry %.2f, pack %.2f, wait for commit %.2f, enqueue %.2f (%llu pictures)\n",
                          g_wdSum[0] / g_wdJobs, g_wdSum[1] / g_wdJobs, g_wdSum[2] / g_wdJobs, g_wdSum[3] / g_wdJobs, g_wdSum[4] / g_wdJobs, g_wdSum[5] / g_wdJobs, (unsigned long long) g_wdJobs );
  for( int k = 0; k < K_NUM; k++ ) if( g_wdCall[k][2] ) fprintf( stderr, "[vvr] host time of %-12s: %5.0f calls, %.1f us on average, %.0f us at most\n", kKernelNames[k], g_wdCall[k][2], 1e3 * g_wdCall[k][0] / g_wdCall[k][2], 1e3 * g_wdCall[k][1] );
  if( g_wdEnqN ) fprintf( stderr, "[vvr] enqueue parts, ms per picture over ALL %llu pictures: upload calls %.3f, event waits %.3f, kernel launches %.3f, record %.3f\n", (unsigned long long) g_wdEnqN, g_wdPart[0] / g_wdEnqN, g_wdPart[1] / g_wdEnqN, g_wdPart[2] / g_wdEnqN, g_wdPart[3] / g_wdEnqN );
  if( g_wdEnqN ) fprintf( stderr, "[vvr] enqueue parts at most (ms): upload calls %.3f, event waits %.3f, kernel launches %.3f\n", g_wdPartMax[0], g_wdPartMax[1], g_wdPartMax[2] );
  if( g_wdEnqN ) fprintf( stderr, "[vvr] enqueue: %llu pictures, %.3f ms each on average, %.3f ms at most\n", (unsigned long long) g_wdEnqN, g_wdEnqMs / g_wdEnqN, g_wdEnqMax );
#endif
  for( auto& kv : c->jobs ) { Job& j = *kv.second; for( auto& t : j.timings ) { hipEventDestroy( t.a ); hipEventDestroy( t.b ); } if( j.done ) hipEventDestroy( j.done ); if( j.doneHost ) hipEventDestroy( j.doneHost ); }
  for( auto e : c->eventPool ) hipEventDestroy( e );
  for( auto& e : c->ring ) { if( e.host ) hipHostFree( e.host ); if( e.dev ) hipFree( e.dev ); if( e.dmvrHost ) hipHostFree( e.dmvrHost ); if( e.colHost ) hipHostFree( e.colHost ); if( e.copied ) hipEventDestroy( e.copied ); }
  for( auto p : c->retiredHost ) hipHostFree( p );
  for( auto p : c->retiredDev ) hipFree( p );
  for( auto s : c->streams ) if( s ) hipStreamDestroy( s );
  if( c->copyStream ) hipStreamDestroy( c->copyStream );
  if( c->planeMemOwned && c->planeMem ) hipFree( c->planeMem );
  if( c->scratchMem ) hipFree( c->scratchMem );
  if( c->outDev ) hipFree( c->outDev );
  if( c->outHost ) hipHostFree( c->outHost );
  for( void* p : c->stagePool ) hipHostFree( p );
  if( c->prepStage ) hipHostFree( c->prepStage );
  if( c->outStream ) hipStreamDestroy( c->outStream );
  for( auto p : c->syncBuf ) hipFree( p );
  for( auto p : c->leafMaps ) hipFree( p );
  if( c->errHost ) hipHostFree( c->errHost );
  if( c->inlineScratch ) vvr_scratch_destroy( c->inlineScratch );
  for( auto& e : c->pinned.r ) hipHostFree( (void*) e.first );
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

VVR_API int vvr_slot_picture_size( vvr_context* c, int slot, int width, int height )
{
  if( !c || slot < 0 || slot >= (int) c->slots.size() || width <= 0 || height <= 0 || width > c->cfg.max_width || height > c->cfg.max_height || ( width & 7 ) || ( height & 7 ) ) return VVR_ERR_PARAMETER;
  c->slotDim[slot] = std::make_pair( (uint16_t) width, (uint16_t) height );
  return VVR_OK;
}
// the picture in a slot: its planes at its own size
static DevPlanes pictureIn( const vvr_context* c, int slot )
{
  DevPlanes d = c->slots[slot];
  for( int k = 0; k < 3; k++ ) { d.w[k] = k ? c->slotDim[slot].first >> 1 : c->slotDim[slot].first; d.h[k] = k ? c->slotDim[slot].second >> 1 : c->slotDim[slot].second; }
  return d;
}

VVR_API int vvr_read_plane( vvr_context* c, int slot, int comp, uint16_t* dst, size_t dstStride )
{
  if( !c || slot < 0 || slot >= (int) c->slots.size() || comp < 0 || comp > 2 || !c->slots[slot].p[comp] ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  const int rc = vvr_sync( c ); if( rc != VVR_OK ) return rc;
  const DevPlanes d = pictureIn( c, slot );
  HIPCHK( c, hipMemcpy2D( dst, dstStride * 2, d.p[comp], (size_t) d.stride[comp] * 2, (size_t) d.w[comp] * 2, d.h[comp], hipMemcpyDeviceToHost ) );
  return VVR_OK;
}

VVR_API int vvr_write_plane( vvr_context* c, int slot, int comp, const uint16_t* src, size_t srcStride )
{
  if( !c || slot < 0 || slot >= (int) c->slots.size() || comp < 0 || comp > 2 || !c->slots[slot].p[comp] ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  const int rc = vvr_sync( c ); if( rc != VVR_OK ) return rc;
  const DevPlanes d = pictureIn( c, slot );
  HIPCHK( c, hipMemcpy2D( d.p[comp], (size_t) d.stride[comp] * 2, src, srcStride * 2, (size_t) d.w[comp] * 2, d.h[comp], hipMemcpyHostToDevice ) );
  return VVR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// submission
// ---------------------------------------------------------------------------------------------------------------------
VVR_API int vvr_submit( vvr_context* c, const vvr_picture* p )
{
  if( !c || !p ) return VVR_ERR_PARAMETER;
  if( p->resident ) { c->setError( "vvr_submit needs host arrays" ); return VVR_ERR_PARAMETER; }
  hipSetDevice( c->device );
  {
    // What is wrong with the header, the tables or the set of arrays is reported here.  The CU / TU records (0.3 ms for a 4K picture) are checked
    // here as well when there are no worker threads; with worker threads that is the first thing the picture's worker does - the submitting
    // thread is the decoder's parser, and a picture further down its queue should not wait behind the checks of the ones before it - and what it
    // finds comes back from vvr_wait, like everything that only shows while the work lists are built and like device errors (the reference
    // parks exceptions on reconDone in the same way).
    std::string err;
    const int rc = c->workers.empty() ? vvr_host_validate( c->cfg, p, err ) : vvr_host_validate_header( c->cfg, p, err );
    if( rc != VVR_OK ) { std::lock_guard<std::mutex> lk( c->mu ); c->setError( err ); return rc; }
    c->slotDim[p->hdr.out_slot] = std::make_pair( p->hdr.width, p->hdr.height );
  }
  Job* job;
  {
    std::unique_lock<std::mutex> lk( c->mu );
    retireLocked( c );
    if( !c->workers.empty() ) c->cv.wait( lk, [&]{ return c->queue.size() < 2 * c->workers.size(); } );       // back-pressure
    job = newJobLocked( c );
    job->pic = *p; job->ringSeq = c->nextRingSeq++;
    WD_STAMP( *job, tSubmit );
    if( !c->workers.empty() ) { c->queue.push_back( job ); c->cv.notify_all(); return job->id; }
    job->state = J_PREPARING;
  }
  // no worker threads: the submitting thread prepares and commits the picture itself
  prepareJob( c, *job, *c->inlineScratch );
  std::lock_guard<std::mutex> lk( c->mu );
  if( job->state == J_FAILED ) { const int rc = job->rc; c->setError( job->err ); job->waited = true; return rc; }
  return job->id;
}

VVR_API int vvr_inputs_done( vvr_context* c, int id )
{
  if( !c ) return VVR_ERR_PARAMETER;
  std::unique_lock<std::mutex> lk( c->mu );
  auto it = c->jobs.find( id );
  if( it == c->jobs.end() ) return VVR_OK;
  Job& j = *it->second;
  c->cv.wait( lk, [&]{ return j.state >= J_READY; } );
  if( j.ring && !j.ring->direct.empty() && j.ring->owner == &j )
  {
    // arrays in pinned memory are read by the copy engine: wait for the picture's upload
    c->cv.wait( lk, [&]{ return j.state == J_COMMITTED || j.completed; } );
    if( !j.completed )
    {
      hipEvent_t ev = j.ring->copied;
      lk.unlock();
      hipSetDevice( c->device );
      hipEventSynchronize( ev );
    }
  }
  return VVR_OK;
}

VVR_API void* vvr_host_alloc( vvr_context* c, size_t bytes )
{
  if( !c || !bytes ) return nullptr;
  hipSetDevice( c->device );
  void* p = nullptr;
  if( hipHostMalloc( &p, bytes, hipHostMallocDefault ) != hipSuccess ) return nullptr;
  std::lock_guard<std::mutex> lk( c->pinned.mu );
  c->pinned.r.emplace_back( (const char*) p, bytes );
  return p;
}

VVR_API void vvr_host_free( vvr_context* c, void* p )
{
  if( !c || !p ) return;
  { std::lock_guard<std::mutex> lk( c->pinned.mu ); for( auto it = c->pinned.r.begin(); it != c->pinned.r.end(); ++it ) if( it->first == (const char*) p ) { c->pinned.r.erase( it ); break; } }
  hipSetDevice( c->device );
  hipHostFree( p );
}

VVR_API void vvr_free_prepared( vvr_context* c, vvr_prepared* q )
{
  if( !q ) return;
  if( !c ) c = q->owner;
  if( c )
  {
    hipSetDevice( c->device );
    // pictures in flight may still read the handle: wait for them
    std::vector<int> ids;
    { std::lock_guard<std::mutex> lk( c->mu ); for( auto& kv : c->jobs ) if( kv.second->q == q && !kv.second->completed ) ids.push_back( kv.first ); }
    for( int id : ids ) finishJob( c, id );
  }
  if( q->blob ) hipFree( q->blob );
  if( q->dmvrHost ) hipHostFree( q->dmvrHost );
  if( q->colHost ) hipHostFree( q->colHost );
  delete q;
}

VVR_API int vvr_prepare( vvr_context* c, const vvr_picture* p, vvr_prepared** out )
{
  if( !c || !p || !out ) return VVR_ERR_PARAMETER;
  if( p->resident ) { c->setError( "vvr_prepare needs host arrays" ); return VVR_ERR_PARAMETER; }
  hipSetDevice( c->device );
  std::string err;
  int rc = vvr_host_validate( c->cfg, p, err );
  size_t total = 0;
  PrepScratch& S = *c->inlineScratch;           // (vvr_prepare and a vvr_submit without worker threads come from the one submitting thread)
  if( rc == VVR_OK ) rc = vvr_host_build( p, S, &total, err );
  if( rc != VVR_OK ) { std::lock_guard<std::mutex> lk( c->mu ); c->setError( err ); return rc; }
  vvr_prepared* q = new vvr_prepared();
  q->owner = c;
  // (pinned staging of the context, grown when a picture needs more: vvr_prepare comes from the one submitting thread)
  if( total > c->prepStageCap )
  {
    if( c->prepStage ) hipHostFree( c->prepStage );
    c->prepStage = nullptr; c->prepStageCap = 0;
    const size_t want = alignUp( total + total / 4, 1 << 16 );
    if( hipHostMalloc( (void**) &c->prepStage, want, hipHostMallocDefault ) == hipSuccess ) c->prepStageCap = want;
  }
  char* staging = c->prepStage;
  if( !staging || hipMalloc( &q->blob, total ) != hipSuccess )
  { vvr_free_prepared( c, q ); c->setError( "vvr_prepare: out of device or pinned memory" ); return VVR_ERR_DEVICE; }
  q->blobBytes = total;
  vvr_host_pack( S, staging );
  vvr_host_bind( S, *q, (char*) q->blob );
  std::vector<DirectCopy> none; size_t b0 = 0, b1 = 0;
  vvr_host_upload_plan( S, none, &b0, &b1 );
  const hipError_t e = hipMemcpy( q->blob, staging, b1, hipMemcpyHostToDevice );
  if( e == hipSuccess && ( p->hdr.tool_flags & VVR_TOOL_COL_MOTION ) )
  {
    q->numCol = vvr_host_num_col( p );
    if( hipHostMalloc( (void**) &q->colHost, sizeof( vvr_motion ) * q->numCol, hipHostMallocDefault ) != hipSuccess ) { vvr_free_prepared( c, q ); c->setError( "vvr_prepare: out of pinned memory" ); return VVR_ERR_DEVICE; }
    vvr_host_gather_col( p, q->colHost );          // (a handle submitted again starts from the motion its last run left: refinement is a function of the picture alone)
    q->pic.colMotion = q->colHost; q->pic.colStride = (int) ( ( ( ( p->hdr.width + 3 ) >> 2 ) + 1 ) >> 1 );
  }
  if( e == hipSuccess && q->numDmvr && hipHostMalloc( (void**) &q->dmvrHost, sizeof( int32_t ) * 2 * (size_t) q->numDmvr, hipHostMallocDefault ) != hipSuccess )
  { vvr_free_prepared( c, q ); c->setError( "vvr_prepare: out of pinned memory" ); return VVR_ERR_DEVICE; }
  if( e != hipSuccess ) { vvr_free_prepared( c, q ); c->setError( "H2D copy failed" ); return VVR_ERR_DEVICE; }
  *out = q;
  return VVR_OK;
}

VVR_API int vvr_submit_prepared( vvr_context* c, vvr_prepared* q )
{
  if( !c || !q ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  Job* job;
  {
    std::lock_guard<std::mutex> lk( c->mu );
    retireLocked( c );
    job = newJobLocked( c );
    job->q = q; job->state = J_READY;
  }
  commitReady( c );                   // (commits nothing yet if pictures queued by vvr_submit are still ahead of it; the launcher gets to it then)
  std::lock_guard<std::mutex> lk( c->mu );
  c->cv.notify_all();
  if( job->state == J_FAILED ) { const int rc = job->rc; c->setError( job->err ); job->waited = true; return rc; }
  return job->id;
}

VVR_API int vvr_wait( vvr_context* c, int job )
{
  if( !c ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  return finishJob( c, job );
}

VVR_API int vvr_test( vvr_context* c, int job )
{
  if( !c ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  std::lock_guard<std::mutex> lk( c->mu );
  auto it = c->jobs.find( job );
  if( it == c->jobs.end() ) return VVR_OK;          // retired: finished long ago
  Job& j = *it->second;
  if( !j.completed )
  {
    if( j.state == J_FAILED ) { c->setError( j.err ); return j.rc; }
    if( j.state != J_COMMITTED || !j.doneHost || hipEventQuery( j.doneHost ) != hipSuccess ) return VVR_NOT_READY;
    completeLocked( c, j );
  }
  if( j.state == J_FAILED ) { c->setError( j.err ); return j.rc; }
  return VVR_OK;
}

VVR_API int vvr_sync( vvr_context* c )
{
  if( !c ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  std::vector<int> ids;
  { std::lock_guard<std::mutex> lk( c->mu ); for( auto& kv : c->jobs ) if( !kv.second->waited ) ids.push_back( kv.first ); }
  int rc = VVR_OK;
  for( int id : ids ) { const int r = finishJob( c, id ); if( r != VVR_OK && rc == VVR_OK ) rc = r; }
  if( rc != VVR_OK ) return rc;
  for( auto s : c->streams ) HIPCHK( c, hipStreamSynchronize( s ) );
  // (a wait of the intra stage that gave up fails the picture's own job: completeLocked reads the job's error word - vvr_wait, vvr_test, vvr_read_* and this call all see it)
  // external events that are complete are forgotten here (the caller may destroy them after this call, vvr.h)
  { std::lock_guard<std::mutex> lk( c->mu ); for( int slot = 0; slot < (int) c->slotExt.size(); slot++ ) pruneExternalEventsLocked( c, slot ); }
  return VVR_OK;
}

VVR_API int vvr_stream_wait_job( vvr_context* c, int job, void* stream, int blocking )
{
  if( !c ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  std::unique_lock<std::mutex> lk( c->mu );
  auto it = c->jobs.find( job );
  if( it == c->jobs.end() ) return VVR_OK;          // retired: finished long ago
  Job* j = it->second.get();
  if( !( j->state == J_COMMITTED || j->completed ) )
  {
    if( !blocking ) return VVR_NOT_READY;
    c->cv.wait( lk, [&]{ auto q = c->jobs.find( job ); return q == c->jobs.end() || q->second->state == J_COMMITTED || q->second->completed; } );
    it = c->jobs.find( job );
    if( it == c->jobs.end() ) return VVR_OK;
    j = it->second.get();
  }
  if( j->state == J_FAILED ) { c->setError( j->err ); return j->rc; }
  if( !j->completed && j->done ) HIPCHK( c, hipStreamWaitEvent( (hipStream_t) stream, j->done, 0 ) );
  return VVR_OK;
}

VVR_API int vvr_stream_wait_slot( vvr_context* c, int slot, void* stream, int blocking )
{
  if( !c || slot < 0 || slot >= (int) c->slotUsers.size() ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  std::unique_lock<std::mutex> lk( c->mu );
  if( !c->bySeq.empty() )                            // pictures still with the workers: who uses the slot is only known once they are committed
  {
    if( !blocking ) return VVR_NOT_READY;
    c->cv.wait( lk, [&]{ return c->bySeq.empty(); } );
  }
  for( int id : c->slotUsers[slot] )
  {
    auto it = c->jobs.find( id );
    if( it == c->jobs.end() ) continue;
    Job& j = *it->second;
    if( !j.completed && j.state == J_COMMITTED && j.done ) HIPCHK( c, hipStreamWaitEvent( (hipStream_t) stream, j.done, 0 ) );
  }
  for( hipEvent_t ev : c->slotExt[slot] ) HIPCHK( c, hipStreamWaitEvent( (hipStream_t) stream, ev, 0 ) );
  return VVR_OK;
}

VVR_API int vvr_slot_external_event( vvr_context* c, int slot, void* event, int writes )
{
  if( !c || !event || slot < 0 || slot >= (int) c->slotUsers.size() ) return VVR_ERR_PARAMETER;
  std::lock_guard<std::mutex> lk( c->mu );
  if( writes ) { c->slotUsers[slot].clear(); c->slotExt[slot].clear(); }
  c->slotExt[slot].push_back( (hipEvent_t) event );
  return VVR_OK;
}

VVR_API void* vvr_job_stream( vvr_context* c, int job )
{
  if( !c ) return nullptr;
  std::unique_lock<std::mutex> lk( c->mu );
  auto it = c->jobs.find( job );
  if( it == c->jobs.end() ) return nullptr;
  Job& j = *it->second;
  c->cv.wait( lk, [&]{ return j.state == J_COMMITTED || j.completed; } );
  return j.lane >= 0 ? (void*) c->streams[j.lane] : nullptr;
}

VVR_API int vvr_read_dmvr( vvr_context* c, int job, int32_t* dst, size_t numEntries )
{
  if( !c || ( !dst && numEntries ) ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  { std::lock_guard<std::mutex> lk( c->mu ); if( c->jobs.find( job ) == c->jobs.end() ) { c->setError( "vvr_read_dmvr: job already retired" ); return VVR_ERR_PARAMETER; } }
  const int rc = finishJob( c, job );
  if( rc != VVR_OK ) return rc;
  std::lock_guard<std::mutex> lk( c->mu );
  auto it = c->jobs.find( job );
  if( it == c->jobs.end() ) { c->setError( "vvr_read_dmvr: job already retired" ); return VVR_ERR_PARAMETER; }
  Job& j = *it->second;
  const size_t n = std::min( numEntries, j.dmvr.size() / 2 );
  if( n ) memcpy( dst, j.dmvr.data(), sizeof( int32_t ) * 2 * n );
  return (int) ( j.dmvr.size() / 2 );
}

VVR_API int vvr_read_col_motion( vvr_context* c, int job, vvr_motion* dst, size_t numEntries )
{
  if( !c || ( !dst && numEntries ) ) return VVR_ERR_PARAMETER;
  hipSetDevice( c->device );
  { std::lock_guard<std::mutex> lk( c->mu ); if( c->jobs.find( job ) == c->jobs.end() ) { c->setError( "vvr_read_col_motion: job already retired" ); return VVR_ERR_PARAMETER; } }
  const int rc = finishJob( c, job );
  if( rc != VVR_OK ) return rc;
  std::lock_guard<std::mutex> lk( c->mu );
  auto it = c->jobs.find( job );
  if( it == c->jobs.end() ) { c->setError( "vvr_read_col_motion: job already retired" ); return VVR_ERR_PARAMETER; }
  Job& j = *it->second;
  const size_t n = std::min( numEntries, j.col.size() );
  if( n ) memcpy( dst, j.col.data(), sizeof( vvr_motion ) * n );
  return (int) j.col.size();
}

VVR_API int vvr_enable_stats( vvr_context* c, int on ) { if( !c ) return VVR_ERR_PARAMETER; vvr_sync( c ); std::lock_guard<std::mutex> lk( c->mu ); c->statsOn = on != 0; for( auto& s : c->stats ) s = Stat(); return VVR_OK; }

VVR_API int vvr_get_stats( vvr_context* c, vvr_kernel_stat* out, int maxEntries )
{
  if( !c ) return VVR_ERR_PARAMETER;
  vvr_sync( c );
  std::lock_guard<std::mutex> lk( c->mu );
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

// practical HBM ceiling ("measure the practical ceiling with a device copy kernel"): the library's copy kernel over as much of
// the DPB as the scratch planes hold (hundreds of MB for a 4K context: well beyond L2 and the Infinity Cache), HIP-event timed; returns bytes
// moved per second (read + written), averaged over `iters` launches.  The DPB is only read; the scratch planes hold nothing between pictures.
VVR_API double vvr_measure_copy_bandwidth( vvr_context* c, int iters )
{
  if( !c || iters <= 0 ) return 0.0;
  hipSetDevice( c->device );
  if( vvr_sync( c ) != VVR_OK ) return 0.0;
  hipEvent_t a, b;
  if( hipEventCreate( &a ) != hipSuccess || hipEventCreate( &b ) != hipSuccess ) return 0.0;
  hipStream_t s = c->streams[0];
  const size_t bytes = std::min( c->slotBytes * c->cfg.num_slots, c->slotBytes * 2 * c->streams.size() ) & ~(size_t) 4095;
  launch_copy_bytes( s, c->planeMem, c->scratchMem, bytes );       // warm-up
  hipEventRecord( a, s );
  for( int i = 0; i < iters; i++ ) launch_copy_bytes( s, c->planeMem, c->scratchMem, bytes );
  hipEventRecord( b, s );
  hipEventSynchronize( b );
  float ms = 0; hipEventElapsedTime( &ms, a, b );
  hipEventDestroy( a ); hipEventDestroy( b );
  return ms > 0 ? 2.0 * (double) bytes * iters / ( ms * 1e-3 ) : 0.0;
}

VVR_API uint8_t vvr_resolve_tr_type( const vvr_pic_header* hdr, const vvr_cu* cu, const vvr_tu* tu, int comp, int implicit_mts, int explicit_intra, int explicit_inter )
{
  // TrQuant::getTrTypes (TrQuant.cpp:330-407).  0 DCT2, 1 DCT8, 2 DST7; returns (ver << 2) | hor
  int hor = 0, ver = 0;
  const bool intra = cu->pred_mode == VVR_PRED_INTRA, luma = comp == 0;
  const bool isImplicit = intra && luma && implicit_mts && cu->lfnst_idx == 0 && !( cu->flags & VVR_CU_MIP );
  const bool isISP = intra && luma && cu->isp_mode;
  if( isISP && cu->lfnst_idx ) return 0;
  if( hdr && !( hdr->tool_flags & VVR_TOOL_MTS ) ) return 0;          // sps->getUseMTS() off: DCT-2 everywhere, also for ISP and SBT blocks (:346)
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

#include "vvr_output.inc"
