// integration/DecLibReconDropIn.cpp — link-time replacement of the reference's reconstruction stage: the member functions of the UNCHANGED class
// vvdec::DecLibRecon (DecoderLib/DecLibRecon.h:143-200), implemented on top of libvvdec_amd.so.  A decoder library linked from the reference's own
// objects minus DecoderLib/DecLibRecon.o plus this file (oracle/Makefile, target dropin: oracle/_ref/libvvdec.so) keeps the public vvdec_* C API
// (include/vvdec/vvdec.h.in) and everything above the seam - bitstream parsing, parameter sets, DPB management, output, SEI - byte for byte:
// DecLib::reconPicture (DecLib.cpp:612-636) calls create( ThreadPool*, unsigned, bool ) / decompressPicture / waitForPrevDecompressedPic exactly
// as before.
//
// What stays on the host, on the reference's own thread pool (row tasks, a submit task and a finish task per picture, ordered behind parseDone and the pictures
// it references; nobody sleeps in vvr_wait):
//   MIDER   DecCu::TaskDeriveCtuMotionInfo for every CTU (merge / AMVP / affine / HMVP derivation needs the finished motion of collocated pictures)
//   flatten vvr_extract.h: CodingStructure -> vvr_picture
// then vvr_submit, the planes back into the Picture's buffers (the application, the hash SEI check and film grain read them there), the DMVR-refined motion
// through DecCu::TaskFinishMotionInfo, reconDone.  LF_INIT (LoopFilter::calcFilterStrengthsCTU) does NOT run here since round 4: the description
// goes out with VVR_TOOL_LFP_ON_DEVICE and the back-end derives the edge parameters from the CU / TU records on the device (VVDEC_AMD_LF_INIT=1 runs the
// reference's own derivation instead, =2 runs both and compares them entry by entry).
// The context (DPB in HBM) is shared by the DecLibRecon instances of one decoder: they are keyed by the decoder's thread pool.
// A Picture object keeps the DPB slot it got when it was first reconstructed (PicListManager recycles Picture objects, so the number of slots is
// the number of Picture objects the decoder ever allocates: its DPB size + pictures in flight).
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>
#include <list>
#include <array>
#include <atomic>
#include <thread>
#include <condition_variable>
#include <functional>
#include <algorithm>
#include <chrono>
#include <exception>
#include <unordered_map>
#include <deque>
#include <set>
#include <iterator>
#include <numeric>
#include <limits>
#include <iostream>
#include <cstring>
#include <cmath>
// (nothing of the reference is redefined or patched: the one thing the extractor reads that is not public - the LMCS tables of Reshape, protected members - is
// read through pointers to members formed in a derived class, integration/vvr_extract.h::LmcsTables)
#include "DecLibRecon.h"
#include "CommonLib/UnitTools.h"
#include "CommonLib/TrQuant_EMT.h"
#include "../include/vvr.h"
#include "../vvdec_amd/csrc/vvr_lf_init.h"      // (self-check only, VVDEC_AMD_LF_INIT=2: the back-end's derivation of the edge parameters, compiled for the host)
#include "vvr_extract.h"

namespace vvdec
{

namespace
{
// The back-end of one coded video sequence: context + which picture every DPB slot holds.  Shared (shared_ptr) by the decoder's DecLibRecon
// instances and by every picture in flight on it: a sequence that activates an SPS the context cannot hold (larger pictures, another sample
// format or CTU size) gets a new one, and the old one goes when its last picture is finished.
struct AmdCtx
{
  vvr_context* ctx = nullptr;
  int numSlots = 0;
  // a Picture object keeps its slot (PicListManager recycles the objects); what the slot holds is a LIFE of the object: the POC it carried when the
  // back-end last wrote the slot, and whether the back-end reconstructed it itself or it was uploaded from the Picture's buffers
  // `held`: the picture in the slot is being reconstructed or its planes are still to be copied into the Picture's buffers (the slot is not given up);
  struct Slot { const Picture* pic = nullptr; int poc = 0; bool ours = false; bool held = false; uint64_t lastUse = 0; };
  std::vector<Slot> slots;
  std::map<const Picture*, int> slotOf;
  // pictures this back-end reconstructed whose slot was given up while the Picture's own buffers did NOT hold the samples (VVDEC_AMD_NO_READBACK): object -> POC
  std::map<const Picture*, int> lostWithoutHostCopy;
  int slotsGivenUp = 0, uploads = 0;
  uint64_t useCounter = 0;
  uint16_t maxW = 0, maxH = 0; uint8_t chroma = 0, bitDepth = 0, log2Ctu = 0;
  ~AmdCtx() { if( ctx ) vvr_destroy( ctx ); }
  bool holds( const SPS& sps ) const
  {
    return ctx && sps.getMaxPicWidthInLumaSamples() <= maxW && sps.getMaxPicHeightInLumaSamples() <= maxH && ( sps.getChromaFormatIdc() == CHROMA_400 ? 0 : 1 ) == chroma
        && sps.getBitDepth() == bitDepth && getLog2( sps.getMaxCUWidth() ) == log2Ctu;
  }
};
struct AmdShared                      // one per decoder instance (keyed by its thread pool)
{
  std::shared_ptr<AmdCtx> cur;
  int users = 0;
  std::mutex mu;                      // one submitting thread at a time (vvr.h), and the slot table
};
enum AmdTaskKind { AMD_ROW, AMD_SUBMIT, AMD_FINISH };
struct AmdTask { int kind; DecLibRecon* d; Picture* pic; int row; };
struct AmdInst                        // one per DecLibRecon instance: what the reference's class has no member for
{
  std::shared_ptr<AmdShared> sh;
  std::shared_ptr<AmdCtx> ctx;        // the context the picture in progress runs on
  vvr_glue::Extracted desc;
  std::vector<int32_t> dmvrOut;
  // the picture in progress
  std::vector<AmdTask> rowTasks; AmdTask submitTask, finishTask;
  std::unique_ptr<std::atomic<int>[]> rowProgress; int numRows = 0;      // CTUs of every CTU row that have their motion and edge parameters
  std::atomic<int> job{ -1 };         // -1: not submitted yet, -2: failed before it could be, else the back-end's job
  std::mutex jobMu; std::condition_variable jobCv;      // the thread that asks for the picture waits here for the hand-over
  std::exception_ptr error;
  int slot = -1;
  bool planesPending = false;         // the planes are still to be copied into the Picture's buffers (waitForPrevDecompressedPic)
  std::atomic<int64_t> usMider{ 0 }, usLfInit{ 0 };      // (summed over the row tasks, which run on several threads)
  double msMider = 0, msLfInit = 0, msFlatten = 0, msSubmit = 0, msDevice = 0, msReadBack = 0, tSubmitted = 0; int pictures = 0;
};
std::mutex g_mu;
std::map<const ThreadPool*, std::weak_ptr<AmdShared>> g_shared;
std::map<const DecLibRecon*, std::unique_ptr<AmdInst>> g_inst;

AmdInst& instOf( const DecLibRecon* d ) { std::lock_guard<std::mutex> lk( g_mu ); return *g_inst.at( d ); }
double nowMs() { return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count(); }
int envInt( const char* name, int def ) { const char* e = getenv( name ); return e ? atoi( e ) : def; }
// LF_INIT: 0 (default) = left to the back-end (VVR_TOOL_LFP_ON_DEVICE: no calcFilterStrengthsCTU here, no table copied or uploaded); 1 = the reference's own
// (the round-3 path); 2 = self-check: the reference's own runs, the back-end's derivation (vvdec_amd/csrc/vvr_lf_init.h, the source k_lf_init is compiled
// from) runs on the host beside it and every difference the deblocking filter would see is reported
int lfInitMode() { static const int m = envInt( "VVDEC_AMD_LF_INIT", 0 ); return m; }
bool noReadBack() { static const bool n = getenv( "VVDEC_AMD_NO_READBACK" ) != nullptr; return n; }
std::atomic<long> g_lfpCheckedCells{ 0 }, g_lfpDifferentCells{ 0 };

// self-check (VVDEC_AMD_LF_INIT=2): the description's tables (the reference's LF_INIT) against the back-end's derivation from the same description
void checkEdgeParameters( const vvr_glue::Extracted& E )
{
  const vvr_picture& p = E.pic;
  if( p.hdr.tool_flags & VVR_TOOL_DEBLOCK_OFF ) return;
  const int w4 = ( p.hdr.width + 3 ) >> 2, h4 = ( p.hdr.height + 3 ) >> 2, ctu = 1 << p.hdr.log2_ctu, ctusX = ( p.hdr.width + ctu - 1 ) / ctu, ctusY = ( p.hdr.height + ctu - 1 ) / ctu;
  std::vector<LfCell> cell( (size_t) w4 * h4 ), cellC( (size_t) w4 * h4 );
  std::vector<LfMv> mv( (size_t) w4 * h4 ); std::vector<uint32_t> ref( (size_t) w4 * h4 );
  lf_init_maps_host( p.hdr, p.cu, p.num_cu, p.tu, p.num_tu, cell.data(), cellC.data(), mv.data(), ref.data(), w4, h4 );
  // (the cells of CUs whose motion varies inside the CU: what the back-end's host stage lists for the device, from the description's motion field)
  for( uint32_t k = 0; k < p.num_cu; k++ )
  {
    const vvr_cu& c = p.cu[k]; vvr_motion m;
    if( lfi_cell_motion( p.hdr, c, c.x >> 2, c.y >> 2, m ) != 2 ) continue;
    for( int y = c.y >> 2; y < std::min( ( c.y + c.h + 3 ) >> 2, h4 ); y++ ) for( int x = c.x >> 2; x < std::min( ( c.x + c.w + 3 ) >> 2, w4 ); x++ ) { mv[(size_t) y * w4 + x] = lfi_pack_mv( p.motion[(size_t) y * w4 + x] ); ref[(size_t) y * w4 + x] = lfi_pack_refs( p.motion[(size_t) y * w4 + x] ); }
  }
  std::vector<uint16_t> ctuSubpic;
  if( p.subpics && p.num_subpics > 1 )
  {
    ctuSubpic.assign( (size_t) ctusX * ctusY, 0 );
    for( uint32_t k = 0; k < p.num_subpics; k++ ) for( int y = p.subpics[k].y0 >> p.hdr.log2_ctu; y <= p.subpics[k].y1 >> p.hdr.log2_ctu; y++ ) for( int x = p.subpics[k].x0 >> p.hdr.log2_ctu; x <= p.subpics[k].x1 >> p.hdr.log2_ctu; x++ ) ctuSubpic[(size_t) y * ctusX + x] = (uint16_t) k;
  }
  LfInitView V; V.hdr = &p.hdr; V.cell = cell.data(); V.cellC = cellC.data(); V.mv = mv.data(); V.ref = ref.data(); V.ctuSlice = p.ctu_slice; V.ctuTile = p.ctu_tile;
  V.ctuSubpic = ctuSubpic.empty() ? nullptr : ctuSubpic.data(); V.subpics = p.subpics; V.slices = p.slices; V.w4 = w4; V.h4 = h4; V.ctusX = ctusX;
  long bad = 0;
  for( int d = 0; d < 2; d++ ) for( int y = 0; y < h4; y++ ) for( int x = 0; x < w4; x++ )
  {
    const vvr_lfp a = p.lfp[d][(size_t) y * w4 + x], b = lf_init_cell( V, d, x, y, cell[(size_t) y * w4 + x] );
    const bool grid = p.hdr.chroma_format && ( ( ( d == 0 ? x : y ) << 2 ) & 15 ) == 0;
    bool differ = ( a.bs & 3 ) != ( b.bs & 3 );
    if( a.bs & 3 ) differ |= a.qp[0] != b.qp[0] || ( a.side_max_filt_length & 0x77 ) != ( b.side_max_filt_length & 0x77 );
    if( grid )
    {
      differ |= ( a.bs & 0x3c ) != ( b.bs & 0x3c );
      if( a.bs & 0x0c ) differ |= a.qp[1] != b.qp[1];
      if( a.bs & 0x30 ) differ |= a.qp[2] != b.qp[2];
      if( a.bs & 0x3c ) differ |= ( a.flags & 0x20 ) != ( b.flags & 0x20 );
    }
    if( differ && bad++ < 4 )
      fprintf( stderr, "[vvdec_amd] edge parameters differ: POC %d dir %d cell (%d, %d): LF_INIT bs %02x len %02x qp %d %d %d flags %02x, derived bs %02x len %02x qp %d %d %d flags %02x\n", p.hdr.poc, d, x, y,
               a.bs, a.side_max_filt_length, a.qp[0], a.qp[1], a.qp[2], a.flags, b.bs, b.side_max_filt_length, b.qp[0], b.qp[1], b.qp[2], b.flags );
  }
  g_lfpCheckedCells += 2L * w4 * h4; g_lfpDifferentCells += bad;
}
}   // namespace

DecLibRecon::DecLibRecon()
{
#if ENABLE_SIMD_OPT_BUFFER
#  if defined( TARGET_SIMD_X86 )
  g_pelBufOP.initPelBufOpsX86();
#  endif
#endif
#if ENABLE_SIMD_TCOEFF_OPS && defined( TARGET_SIMD_X86 )
  g_tCoeffOps.initTCoeffOpsX86();
#endif
}

void DecLibRecon::create( ThreadPool* threadPool, unsigned /*instanceId*/, bool upscaleOutputEnabled )
{
  this->~DecLibRecon();
  new( this ) DecLibRecon;
  m_decodeThreadPool     = threadPool;
  m_numDecThreads        = std::max( 1, threadPool ? threadPool->numThreads() : 1 );
  m_upscaleOutputEnabled = upscaleOutputEnabled;
  m_predBufSize = 0; m_dmvrMvCacheSize = 0; m_dmvrMvCache = nullptr; m_num4x4Elements = 0; m_loopFilterParam = nullptr; m_motionInfo = nullptr;
  // MIDER / TaskFinishMotionInfo run on pool threads: one DecCu per thread, as in the reference (DecLibRecon.cpp:163-168)
  m_pcThreadResource    = new PerThreadResource*[m_numDecThreads];
  m_pcThreadResource[0] = new PerThreadResource();
  for( int i = 1; i < m_numDecThreads; i++ ) m_pcThreadResource[i] = new PerThreadResource( m_pcThreadResource[0]->m_cTrQuant );
  std::lock_guard<std::mutex> lk( g_mu );
  std::unique_ptr<AmdInst> I( new AmdInst );
  I->sh = g_shared[threadPool].lock();
  if( !I->sh ) { I->sh = std::make_shared<AmdShared>(); g_shared[threadPool] = I->sh; }
  I->sh->users++;
  g_inst[this] = std::move( I );
}

void DecLibRecon::destroy()
{
  m_decodeThreadPool = nullptr;
  if( m_dmvrMvCache ) { free( m_dmvrMvCache ); m_dmvrMvCache = nullptr; m_dmvrMvCacheSize = 0; }
  if( m_loopFilterParam ) { free( m_loopFilterParam ); m_loopFilterParam = nullptr; }
  if( m_motionInfo ) { free( m_motionInfo ); m_motionInfo = nullptr; }
  m_num4x4Elements = 0;
  if( m_pcThreadResource ) { for( int i = 0; i < m_numDecThreads; i++ ) delete m_pcThreadResource[i]; delete[] m_pcThreadResource; m_pcThreadResource = nullptr; }
  std::lock_guard<std::mutex> lk( g_mu );
  auto it = g_inst.find( this );
  if( it != g_inst.end() )
  {
    if( getenv( "VVDEC_AMD_TIMES" ) && it->second->pictures )
      fprintf( stderr, "[vvdec_amd] %d pictures, host ms per picture: MIDER %.2f, LF_INIT %.2f, flatten %.2f, submit+device %.2f, planes back %.2f\n", it->second->pictures,
               it->second->msMider / it->second->pictures, it->second->msLfInit / it->second->pictures, it->second->msFlatten / it->second->pictures,
               ( it->second->msSubmit + it->second->msDevice ) / it->second->pictures, it->second->msReadBack / it->second->pictures );
    if( getenv( "VVDEC_AMD_TIMES" ) && it->second->pictures && it->second->ctx )
      fprintf( stderr, "[vvdec_amd] slots given up: %d, reference pictures uploaded from host memory: %d\n", it->second->ctx->slotsGivenUp, it->second->ctx->uploads );
    if( lfInitMode() == 2 && it->second->pictures )
      fprintf( stderr, "[vvdec_amd] edge parameters: %ld entries checked against the reference's LF_INIT, %ld differ\n", g_lfpCheckedCells.exchange( 0 ), g_lfpDifferentCells.exchange( 0 ) );
    g_inst.erase( it );        // (the last instance of a decoder takes the context, hence the DPB in HBM, with it)
  }
}

void DecLibRecon::swapBufs( CodingStructure& ) {}      // (ALF writes the DPB slot itself on the device: nothing to swap)

// A picture is a handful of tasks on the reference's own thread pool, all added here (the pool takes tasks from one thread only, ThreadPool.h:67), none of
// which waits for the device:
//   one task per CTU ROW   MIDER + LF_INIT of its CTUs, gated on the parser's progress in that row (ctuParsedBarrier, DecLibRecon.cpp:617-620) and on the
//                          pictures it references; a CTU waits for its above-right neighbour's motion like the reference's MIDER state (:762-805) - a
//                          row that cannot go on hands its thread back and is called again
//   the SUBMIT task        behind all rows: the flat description (vvr_extract.h), vvr_submit - which returns at once, the back-end's own workers build
//                          the device work lists
//   the FINISH task        ready when vvr_test says the picture is reconstructed: DMVR-refined motion through DecCu::TaskFinishMotionInfo, reconDone
// so the decoder's DecLibRecon instances (DecLib.h:70) keep as many pictures on the back-end as there are instances, parsing of a picture overlaps with
// the motion derivation of its upper rows, and no pool thread sleeps.  The planes are copied into the Picture's buffers (output, hash SEI, film grain
// read them there) by waitForPrevDecompressedPic, on the thread that asks for the picture.
// The class declares one private task function, ctuTask<onlyCheckReadyState> (DecLibRecon.h:196-197): its definition here is all three tasks and their
// ready checks, so they may use the class's members like the reference's does.
void DecLibRecon::decompressPicture( Picture* pcPic )
{
  m_currDecompPic = pcPic;
  CodingStructure& cs = *pcPic->cs;
  pcPic->progress = Picture::reconstructing;
  const SPS* sps = cs.sps.get();
  for( int i = 0; i < m_numDecThreads; i++ )
  {
    if( sps->getUseReshaper() )
    {
      m_pcThreadResource[i]->m_cReshaper.createDec( sps->getBitDepth() );
      m_pcThreadResource[i]->m_cReshaper.initSlice( pcPic->slices[0]->getNalUnitLayerId(), *pcPic->slices[0]->getPicHeader(), pcPic->slices[0]->getVPS_nothrow() );
    }
    m_pcThreadResource[i]->m_cIntraPred.init( sps->getChromaFormatIdc(), sps->getBitDepth() );
    m_pcThreadResource[i]->m_cInterPred.init( &m_cRdCost, sps->getChromaFormatIdc(), sps->getMaxCUHeight() );
    m_pcThreadResource[i]->m_cTrQuant.init( pcPic );
    m_pcThreadResource[i]->m_cCuDecoder.init( &m_pcThreadResource[i]->m_cIntraPred, &m_pcThreadResource[i]->m_cInterPred, &m_pcThreadResource[i]->m_cReshaper, &m_pcThreadResource[i]->m_cTrQuant );
  }
  const PreCalcValues* pcv = cs.pcv;
  const size_t maxDmvr = pcv->num8x8CtuBlks * pcv->sizeInCtus;                  // (DecLibRecon.cpp:497-505)
  if( maxDmvr != m_dmvrMvCacheSize ) { if( m_dmvrMvCache ) free( m_dmvrMvCache ); m_dmvrMvCacheSize = maxDmvr; m_dmvrMvCache = (Mv*) malloc( sizeof( Mv ) * maxDmvr ); }
  cs.m_dmvrMvCache = m_dmvrMvCache;
  cs.m_predBuf     = nullptr;                                                    // (prediction scratch of the CPU path: not needed)
  if( m_num4x4Elements != (ptrdiff_t) ( pcv->num4x4CtuBlks * pcv->sizeInCtus ) )
  {
    if( m_loopFilterParam ) free( m_loopFilterParam );
    if( m_motionInfo ) free( m_motionInfo );
    m_num4x4Elements  = pcv->num4x4CtuBlks * pcv->sizeInCtus;
    m_loopFilterParam = (LoopFilterParam*) malloc( sizeof( LoopFilterParam ) * m_num4x4Elements * 2 );
    m_motionInfo      = (MotionInfo*) malloc( sizeof( MotionInfo ) * m_num4x4Elements );
  }
  pcPic->startProcessingTimer();
  AmdInst& I = instOf( this );
  const int widthInCtus = (int) pcv->widthInCtus, heightInCtus = (int) pcv->heightInCtus;
  I.numRows = heightInCtus;
  I.rowProgress.reset( new std::atomic<int>[heightInCtus] );
  for( int r = 0; r < heightInCtus; r++ ) I.rowProgress[r].store( 0 );
  I.job.store( -1 ); I.error = nullptr; I.slot = -1; I.planesPending = false;
  // a caller that hands over pictures whose motion is already derived (the test harness builds its coding units with final motion vectors and binds
  // the per-CTU motion buffers itself) skips MIDER
  commonTaskParam.cs = &cs;
  commonTaskParam.perLineMiHist = std::vector<MotionHist>( heightInCtus );
  // ordered behind every picture it references: their samples live in the back-end's DPB, but their FINISHED MOTION (TaskFinishMotionInfo) is what
  // MIDER of this picture reads
  CBarrierVec refBarriers;
  for( Picture* ref : pcPic->buildAllRefPicsVec() ) if( std::find( refBarriers.cbegin(), refBarriers.cend(), &ref->reconDone ) == refBarriers.cend() ) refBarriers.push_back( &ref->reconDone );
  pcPic->reconDone.lock();
  I.rowTasks.assign( heightInCtus, AmdTask{ AMD_ROW, this, pcPic, 0 } );
  for( int r = 0; r < heightInCtus; r++ )
  {
    I.rowTasks[r].row = r;
    CBarrierVec barriers = refBarriers;
    if( pcPic->parseDone.isBlocked() )
    {
      // wait for the last CTU of the row to be parsed (DecLibRecon.cpp:617-620); a picture without per-CTU barriers: for the whole picture
      if( (int) pcPic->ctuParsedBarrier.size() >= ( r + 1 ) * widthInCtus ) barriers.push_back( &pcPic->ctuParsedBarrier[( r + 1 ) * widthInCtus - 1] );
      else barriers.push_back( &pcPic->parseDone );
    }
    m_decodeThreadPool->addBarrierTask( TP_TASK_NAME_ARG( "POC:" + std::to_string( pcPic->poc ) + " vvdec_amd row " + std::to_string( r ) )
                                        ctuTask<false>, &I.rowTasks[r], &pcPic->m_ctuTaskCounter, nullptr, std::move( barriers ), ctuTask<true> );
  }
  I.submitTask = AmdTask{ AMD_SUBMIT, this, pcPic, 0 };
  m_decodeThreadPool->addBarrierTask( TP_TASK_NAME_ARG( "POC:" + std::to_string( pcPic->poc ) + " vvdec_amd submit" )
                                      ctuTask<false>, &I.submitTask, &pcPic->m_divTasksCounter, nullptr, { pcPic->m_ctuTaskCounter.donePtr(), &pcPic->parseDone } );
  I.finishTask = AmdTask{ AMD_FINISH, this, pcPic, 0 };
  m_decodeThreadPool->addBarrierTask( TP_TASK_NAME_ARG( "POC:" + std::to_string( pcPic->poc ) + " vvdec_amd finish" )
                                      ctuTask<false>, &I.finishTask, &pcPic->m_divTasksCounter, &pcPic->reconDone, { pcPic->m_ctuTaskCounter.donePtr() }, ctuTask<true> );
}

template<bool onlyCheckReadyState>
bool DecLibRecon::ctuTask( int tid, void* task_param )
{
  AmdTask*            T     = static_cast<AmdTask*>( task_param );
  DecLibRecon&        d     = *T->d;
  Picture*            pic   = T->pic;
  CodingStructure&    cs    = *pic->cs;
  AmdInst&            I     = instOf( &d );
  const PreCalcValues& pcv  = *cs.pcv;
  const int numCtu = (int) pcv.sizeInCtus, wCtus = (int) pcv.widthInCtus;
  PerThreadResource&  R     = *d.m_pcThreadResource[std::max( 0, std::min( tid, d.m_numDecThreads - 1 ) )];

  if( T->kind == AMD_ROW )
  {
    // ---- MIDER (DecLibRecon.cpp:763-805) and LF_INIT (:807-829) of one CTU row.  CTU c needs the motion of its left neighbour (this task) and of its
    // above-right neighbour (the row above has got past c + 1); the edge parameters of a CTU read the motion of the CTUs left of and above it.
    const int r = T->row;
    std::atomic<int>& mine = I.rowProgress[r];
    int c = mine.load( std::memory_order_relaxed );
    auto aboveAllows = [&]( int col ) { return r == 0 || I.rowProgress[r - 1].load( std::memory_order_acquire ) >= std::min( col + 2, wCtus ); };
    if( onlyCheckReadyState ) return c >= wCtus || aboveAllows( c );
    for( ; c < wCtus; c++ )
    {
      if( !aboveAllows( c ) ) return false;                          // (called again when the pool gets round to it)
      const int a = r * wCtus + c;
      CtuData& cd = cs.getCtuData( a );
      const double t0 = nowMs();
      if( cd.motion == nullptr )
      {
        cd.motion = &d.m_motionInfo[pcv.num4x4CtuBlks * a];
        if( !cd.slice->isIntra() || cs.sps->getIBCFlag() )
        {
          const UnitArea ctuArea = getCtuArea( cs, c, r, true );
          R.m_cCuDecoder.TaskDeriveCtuMotionInfo( cs, a, ctuArea, d.commonTaskParam.perLineMiHist[r] );
        }
        else memset( NO_WARNING_class_memaccess( cd.motion ), MI_NOT_VALID, sizeof( MotionInfo ) * pcv.num4x4CtuBlks );
      }
      const double t1 = nowMs();
      if( lfInitMode() )
      {
        cd.lfParam[0] = &d.m_loopFilterParam[pcv.num4x4CtuBlks * ( 2 * a + 0 )];
        cd.lfParam[1] = &d.m_loopFilterParam[pcv.num4x4CtuBlks * ( 2 * a + 1 )];
        memset( cd.lfParam[0], 0, sizeof( LoopFilterParam ) * 2 * pcv.num4x4CtuBlks );
        d.m_cLoopFilter.calcFilterStrengthsCTU( cs, a );
      }
      const double t2 = nowMs();
      I.usMider += (int64_t) ( 1e3 * ( t1 - t0 ) ); I.usLfInit += (int64_t) ( 1e3 * ( t2 - t1 ) );
      mine.store( c + 1, std::memory_order_release );
    }
    return true;
  }

  if( T->kind == AMD_SUBMIT )
  {
    if( onlyCheckReadyState ) return true;
    // ---- the flat description and the hand-over to the back-end.  Whatever goes wrong is kept for the finish task (which carries reconDone): a task
    // that throws here would leave the finish task waiting for a job that never comes.
    try
    {
      AmdShared& S = *I.sh;
      const double t2 = nowMs();
      const int hostThreads = envInt( "VVDEC_AMD_HOST_THREADS", std::min( 16, std::max( 4, d.m_decodeThreadPool ? d.m_decodeThreadPool->numThreads() : 0 ) ) );      // threads of the flattening
      std::string why;
      if( vvr_glue::checkExpressible( cs, *pic, why ) != VVR_OK ) THROW_RECOVERABLE( "vvdec_amd: " << why );      // never flattened into something it is not
      Slice& slice = *pic->slices[0];
      Reshape* rsp = nullptr;
      bool lmcs = false;                                                                   // (LMCS is a switch of every slice header: the tables are the picture's)
      for( const Slice* sl : pic->slices ) lmcs |= sl->getLmcsEnabledFlag();
      if( cs.sps->getUseReshaper() && lmcs ) rsp = &R.m_cReshaper;                         // (initSlice was called in decompressPicture)
      if( cs.sps->getUseALF() ) for( Slice* sl : pic->slices ) AdaptiveLoopFilter::reconstructCoeffAPSs( *sl );      // (every slice names its own APSs)
      std::lock_guard<std::mutex> lk( S.mu );                                              // (one submitting thread at a time: vvr.h)
      if( !S.cur || !S.cur->holds( *cs.sps ) )
      {
        // the back-end of this coded video sequence: created with its first picture (size, sample format and CTU size are the sequence's).  A
        // sequence the current context cannot hold gets a new one; the pictures still in flight on the old one keep it alive until they are finished,
        // pictures of the old sequence that are still referenced are uploaded from their Picture's buffers like any picture this context has not seen.
        std::shared_ptr<AmdCtx> N = std::make_shared<AmdCtx>();
        vvr_config cfg; memset( &cfg, 0, sizeof( cfg ) );
        cfg.abi_version = VVR_ABI_VERSION;
        cfg.device = envInt( "VVDEC_AMD_DEVICE", 0 );
        cfg.max_width = N->maxW = (uint16_t) cs.sps->getMaxPicWidthInLumaSamples(); cfg.max_height = N->maxH = (uint16_t) cs.sps->getMaxPicHeightInLumaSamples();
        cfg.chroma_format = N->chroma = cs.sps->getChromaFormatIdc() == CHROMA_400 ? 0 : 1; cfg.bit_depth = N->bitDepth = (uint8_t) cs.sps->getBitDepth();
        cfg.log2_ctu = N->log2Ctu = (uint8_t) getLog2( cs.sps->getMaxCUWidth() );
        N->numSlots = envInt( "VVDEC_AMD_SLOTS", 48 );                                     // Picture objects the decoder keeps: DPB size + pictures in flight (more: least recently used slot is given up)
        cfg.num_slots = (uint8_t) N->numSlots; cfg.num_streams = 4; cfg.read_buffers = 2;
        cfg.host_threads = (uint8_t) envInt( "VVDEC_AMD_BACKEND_THREADS", 2 );             // the back-end's own workers build the device work lists: vvr_submit returns at once
        if( vvr_create( &cfg, &N->ctx ) != VVR_OK ) { N->ctx = nullptr; THROW_RECOVERABLE( "vvdec_amd: no MI355X back-end (vvr_create failed)" ); }
        N->slots.resize( N->numSlots );
        S.cur = N;
      }
      I.ctx = S.cur;
      AmdCtx& X = *I.ctx;
      // the slot of a Picture object; when every slot is taken, the least recently used one that this picture does not need is given up (whoever
      // references its picture later finds it gone and uploads it from the Picture's buffers)
      std::vector<const Picture*> needed; needed.push_back( pic );
      for( Picture* ref : pic->buildAllRefPicsVec() ) needed.push_back( ref );
      auto slotFor = [&]( const Picture* p, bool* isNew ) -> int
      {
        auto it = X.slotOf.find( p );
        if( isNew ) *isNew = it == X.slotOf.end();
        if( it == X.slotOf.end() )
        {
          int s = -1;
          for( int k = 0; k < X.numSlots && s < 0; k++ ) if( !X.slots[k].pic ) s = k;
          if( s < 0 )
          {
            for( int k = 0; k < X.numSlots; k++ )
              if( !X.slots[k].held && std::find( needed.begin(), needed.end(), X.slots[k].pic ) == needed.end() && ( s < 0 || X.slots[k].lastUse < X.slots[s].lastUse ) ) s = k;
            CHECK( s < 0, "vvdec_amd: a picture, its reference pictures and the pictures still in flight need more DPB slots than the back-end has (VVDEC_AMD_SLOTS)" );
            if( X.slots[s].ours && noReadBack() ) X.lostWithoutHostCopy[X.slots[s].pic] = X.slots[s].poc;
            X.slotOf.erase( X.slots[s].pic ); X.slotsGivenUp++;
          }
          X.slots[s] = AmdCtx::Slot(); X.slots[s].pic = p;
          it = X.slotOf.emplace( p, s ).first;
        }
        X.slots[it->second].lastUse = ++X.useCounter;
        return it->second;
      };
      I.slot = slotFor( pic, nullptr );
      // a reference picture whose slot does not hold what the Picture object holds now - a picture this back-end has not reconstructed: the grey
      // picture the decoder makes up for a missing reference (DecLibParser::prepareUnavailablePicture: a recycled Picture object, no picture header),
      // a picture of an earlier context, a picture handed in from outside - is uploaded from the Picture's own buffers, once per life of the object
      for( Picture* ref : pic->buildAllRefPicsVec() )
      {
        bool isNew = false;
        const int rs = slotFor( ref, &isNew );
        AmdCtx::Slot& sl = X.slots[rs];
        const bool madeUp = ref->slices.empty() || ref->slices[0]->getPicHeader() == nullptr;
        const bool current = !isNew && sl.poc == ref->poc && ( sl.ours ? !madeUp : true );
        if( current ) continue;
        {
          // (what is uploaded must be what the decoder reconstructed: a picture of ours that lost its slot before its samples reached the Picture's buffers is gone.
          // A conforming stream does not get here: a slot is only given up for a picture that is in no reference picture list of the current picture, and such a
          // picture is never referenced again)
          auto lost = X.lostWithoutHostCopy.find( ref );
          if( lost != X.lostWithoutHostCopy.end() && lost->second == ref->poc && !madeUp )
            THROW_RECOVERABLE( "vvdec_amd: reference picture POC " << ref->poc << " lost its slot without a copy in host memory (VVDEC_AMD_NO_READBACK with too few VVDEC_AMD_SLOTS)" );
          if( lost != X.lostWithoutHostCopy.end() ) X.lostWithoutHostCopy.erase( lost );
        }
        vvr_slot_picture_size( X.ctx, rs, (int) ref->lwidth(), (int) ref->lheight() );      // (a coded video sequence may change its picture size)
        CPelUnitBuf rb = const_cast<const Picture*>( ref )->getRecoBuf();
        for( size_t c = 0; c < rb.bufs.size(); c++ )
          if( vvr_write_plane( X.ctx, rs, (int) c, reinterpret_cast<const uint16_t*>( rb.bufs[c].buf ), (size_t) rb.bufs[c].stride ) != VVR_OK ) THROW_RECOVERABLE( "vvdec_amd: " << vvr_last_error( X.ctx ) );
        sl.poc = ref->poc; sl.ours = false; X.uploads++;
      }
      { AmdCtx::Slot& sl = X.slots[I.slot]; sl.poc = pic->poc; sl.ours = true; sl.held = true; X.lostWithoutHostCopy.erase( pic ); }
      const double t2b = nowMs();
      vvr_glue::extractPicture( cs, slice, *pic, rsp, R.m_cTrQuant, [&X]( const Picture* p ) { auto q = X.slotOf.find( p ); return q == X.slotOf.end() ? -1 : q->second; }, I.slot, I.desc, hostThreads, /* the motion field only where the back-end reads it */ lfInitMode() != 2, /* edge parameters: the back-end's */ lfInitMode() == 0 );
      if( lfInitMode() == 2 ) checkEdgeParameters( I.desc );
      const double t3 = nowMs(); I.msFlatten += t3 - t2b;
      const int job = vvr_submit( X.ctx, &I.desc.pic );
      if( job < 0 ) THROW_RECOVERABLE( "vvdec_amd: " << vvr_last_error( X.ctx ) );
      I.tSubmitted = nowMs(); I.msSubmit += I.tSubmitted - t3 + ( t2b - t2 );
      { std::lock_guard<std::mutex> jl( I.jobMu ); I.job.store( job, std::memory_order_release ); }
    }
    catch( ... )
    {
      I.error = std::current_exception();
      { std::lock_guard<std::mutex> jl( I.jobMu ); I.job.store( -2, std::memory_order_release ); }
    }
    I.jobCv.notify_all();
    return true;
  }

  // ---- AMD_FINISH: ready when the back-end says the picture is reconstructed (nobody sleeps in vvr_wait)
  {
    const int job = I.job.load( std::memory_order_acquire );
    if( onlyCheckReadyState )
    {
      if( job == -1 ) return false;
      if( job == -2 ) return true;
      return vvr_test( I.ctx->ctx, job ) != VVR_NOT_READY;
    }
    if( job == -1 ) return false;
    if( job == -2 ) std::rethrow_exception( I.error );
    if( vvr_wait( I.ctx->ctx, job ) < 0 ) THROW_RECOVERABLE( "vvdec_amd: " << vvr_last_error( I.ctx->ctx ) );        // (returns at once: the status)
    I.msDevice += nowMs() - I.tSubmitted;
    I.msMider += 1e-3 * (double) I.usMider.exchange( 0 ); I.msLfInit += 1e-3 * (double) I.usLfInit.exchange( 0 );
    I.planesPending = true; I.pictures++;
    // the DMVR-refined MVs feed the temporal MV prediction of later pictures: through the reference's own finish step (DecCu.cpp:161)
    if( pic->stillReferenced && I.desc.numDmvr )
    {
      I.dmvrOut.resize( 2 * (size_t) I.desc.numDmvr );
      // (a short or failed read would leave zeros / stale vectors in the cache that feeds the temporal MV prediction of later pictures: stop here instead)
      const int got = vvr_read_dmvr( I.ctx->ctx, job, I.dmvrOut.data(), I.desc.numDmvr );
      if( got < (int) I.desc.numDmvr ) THROW_RECOVERABLE( "vvdec_amd: vvr_read_dmvr returned " << got << " of " << I.desc.numDmvr << " refined motion vectors: " << vvr_last_error( I.ctx->ctx ) );
      for( auto& e : I.desc.dmvrCus )
      {
        CodingUnit& cu = *e.first;
        const int n = std::max( 1, (int) cu.lwidth() >> 4 ) * std::max( 1, (int) cu.lheight() >> 4 );
        for( int k = 0; k < n; k++ ) cs.m_dmvrMvCache[cu.mvdL0SubPuOff + k] = Mv( I.dmvrOut[2 * ( e.second + k )], I.dmvrOut[2 * ( e.second + k ) + 1] );
        cu.setDmvrCondition( true );
      }
    }
    if( pic->stillReferenced ) for( int a = 0; a < numCtu; a++ ) R.m_cCuDecoder.TaskFinishMotionInfo( cs, a, a % wCtus, a / wCtus );
    pic->stopProcessingTimer();
    pic->progress = Picture::reconstructed;
    return true;                                                  // (the pool unlocks reconDone)
  }
}
template bool DecLibRecon::ctuTask<false>( int, void* );
template bool DecLibRecon::ctuTask<true>( int, void* );

Picture* DecLibRecon::waitForPrevDecompressedPic()
{
  if( !m_currDecompPic ) return nullptr;
  AmdInst& I = instOf( this );
  // (the slot may be given up again once this picture has left the back-end, with or without an error)
  auto release = [&]{ if( I.ctx && I.slot >= 0 ) { std::lock_guard<std::mutex> lk( I.sh->mu ); if( I.slot < (int) I.ctx->slots.size() && I.ctx->slots[I.slot].pic == m_currDecompPic ) I.ctx->slots[I.slot].held = false; } };
  try
  {
    if( m_decodeThreadPool->numThreads() == 0 )
    {
      // everything on the calling thread: the rows and the hand-over, the one wait for the device there is, the finish task
      m_decodeThreadPool->processTasksOnMainThread();
      const int job = I.job.load();
      if( job >= 0 ) vvr_wait( I.ctx->ctx, job );
      m_decodeThreadPool->processTasksOnMainThread();
      CHECK_FATAL( m_currDecompPic->reconDone.isBlocked(), "can't make progress. some dependecy has not been finished" );
    }
    else if( m_currDecompPic->reconDone.isBlocked() )
    {
      // The finish task is polled by the pool's threads while they look for work (its ready check is vvr_test) - but a pool that has been idle for a few
      // milliseconds goes to sleep until a task is added (ThreadPool.cpp:241-256), and nothing tells it that the device is done.  The thread that
      // asks for the picture is the one that may wait: for the hand-over, then for the device (vvr_wait), then it wakes the pool with an empty task.
      {
        // (the hand-over is announced on jobCv; a picture whose PARSING failed never gets that far - the pool drops its row, submit and finish tasks as their
        // barriers carry the parser's exception, which ends up on reconDone with nobody to announce it: looked for every few milliseconds, isBlocked() rethrows it.
        // Found by tools/fuzz_dropin_on_the_oracle.py: a stream with a broken first picture, which the reference decoder skips, left this thread waiting for ever)
        std::unique_lock<std::mutex> jl( I.jobMu );
        while( !( I.job.load( std::memory_order_acquire ) != -1 || !m_currDecompPic->reconDone.isBlocked() ) ) I.jobCv.wait_for( jl, std::chrono::milliseconds( 2 ) );
      }
      const int job = I.job.load( std::memory_order_acquire );
      if( job >= 0 ) vvr_wait( I.ctx->ctx, job );
      static auto wakeUp = []( int, void* ) { return true; };
      m_decodeThreadPool->addBarrierTask( TP_TASK_NAME_ARG( "vvdec_amd wake-up" ) wakeUp, nullptr );
    }
    m_currDecompPic->reconDone.wait();
    // ---- the picture as the rest of the decoder expects it: planes in the Picture's own buffers (output, hash SEI, film grain) - this picture only, the
    // others in flight are not waited for; through pinned staging, rows laid out by a few threads of this call.  VVDEC_AMD_NO_READBACK=1 (throughput
    // experiments only: output and hash checks then see stale buffers) leaves it out.
    if( I.planesPending && !noReadBack() )
    {
      const double t4 = nowMs();
      PelUnitBuf reco = m_currDecompPic->getRecoBuf();
      uint16_t* dst[3] = { nullptr, nullptr, nullptr }; size_t stride[3] = { 0, 0, 0 };
      for( size_t c = 0; c < reco.bufs.size(); c++ ) { dst[c] = reinterpret_cast<uint16_t*>( reco.bufs[c].buf ); stride[c] = (size_t) reco.bufs[c].stride; }
      const int hostThreads = envInt( "VVDEC_AMD_HOST_THREADS", std::min( 16, std::max( 4, m_decodeThreadPool->numThreads() ) ) );
      if( vvr_read_picture( I.ctx->ctx, I.slot, dst, stride, hostThreads ) != VVR_OK ) THROW_RECOVERABLE( "vvdec_amd: " << vvr_last_error( I.ctx->ctx ) );
      I.msReadBack += nowMs() - t4;
    }
    I.planesPending = false;
    release();
    m_currDecompPic->cs->deallocTempInternals();
  }
  catch( ... )
  {
    m_currDecompPic->error = true;
    m_currDecompPic->reconDone.setException( std::current_exception() );
    release();
  }
  if( m_currDecompPic->error || m_currDecompPic->reconDone.hasException() ) cleanupOnException();
  return std::exchange( m_currDecompPic, nullptr );
}

void DecLibRecon::cleanupOnException() { m_currDecompPic->waitForAllTasks(); }

}   // namespace vvdec
