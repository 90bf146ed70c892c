// integration/DecLibReconDropIn.cpp — link-time replacement of the reference's reconstruction stage: the member functions of the UNCHANGED class
// vvdec::DecLibRecon (DecoderLib/DecLibRecon.h:143-200), implemented on top of libvvdec_amd.so.  A decoder library linked from the reference's own
// objects minus DecoderLib/DecLibRecon.o plus this file (oracle/Makefile, target dropin: oracle/_ref/libvvdec.so) keeps the public vvdec_* C API
// (include/vvdec/vvdec.h.in) and everything above the seam - bitstream parsing, parameter sets, DPB management, output, SEI - byte for byte:
// DecLib::reconPicture (DecLib.cpp:612-636) calls create( ThreadPool*, unsigned, bool ) / decompressPicture / waitForPrevDecompressedPic exactly
// as before.
//
// What stays on the host, on the reference's own thread pool (one barrier task per picture, ordered behind parseDone and the pictures it references):
//   MIDER   DecCu::TaskDeriveCtuMotionInfo for every CTU (merge / AMVP / affine / HMVP derivation needs the finished motion of collocated pictures)
//   LF_INIT LoopFilter::calcFilterStrengthsCTU (the edge-parameter tables are an input of the back-end)
//   flatten vvr_extract.h: CodingStructure -> vvr_picture
// then vvr_submit / vvr_wait, the planes back into the Picture's buffers (the application, the hash SEI check and film grain read them there),
// the DMVR-refined motion through DecCu::TaskFinishMotionInfo, reconDone.
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
// The extractor reads two things that are not public in the reference (the LMCS tables of Reshape, TrQuant::getTrTypes); a maintainer who compiles
// this file into the reference tree adds two friend declarations instead of the next two lines (the layout of the classes does not change).
#define private public
#define protected public
#include "DecLibRecon.h"
#include "CommonLib/UnitTools.h"
#include "CommonLib/TrQuant_EMT.h"
#undef private
#undef protected
#include "../include/vvr.h"
#include "vvr_extract.h"

namespace vvdec
{

namespace
{
struct AmdShared                      // one per decoder instance
{
  vvr_context* ctx = nullptr;
  std::map<const Picture*, int> slotOf;
  int nextSlot = 0, numSlots = 0, users = 0;
  std::mutex mu;
  ~AmdShared() { if( ctx ) vvr_destroy( ctx ); }
};
struct AmdInst                        // one per DecLibRecon instance: what the reference's class has no member for
{
  std::shared_ptr<AmdShared> sh;
  vvr_glue::Extracted desc;
  std::vector<int32_t> dmvrOut;
  double msMider = 0, msLfInit = 0, msFlatten = 0, msDevice = 0, msReadBack = 0; int pictures = 0;
};
std::mutex g_mu;
std::map<const ThreadPool*, std::weak_ptr<AmdShared>> g_shared;
std::map<const DecLibRecon*, std::unique_ptr<AmdInst>> g_inst;

AmdInst& instOf( const DecLibRecon* d ) { std::lock_guard<std::mutex> lk( g_mu ); return *g_inst.at( d ); }
double nowMs() { return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count(); }
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
               it->second->msDevice / it->second->pictures, it->second->msReadBack / it->second->pictures );
    g_inst.erase( it );        // (the last instance of a decoder takes the context, hence the DPB in HBM, with it)
  }
}

void DecLibRecon::swapBufs( CodingStructure& ) {}      // (ALF writes the DPB slot itself on the device: nothing to swap)

// the whole picture as ONE task of the reference's thread pool (or of the main thread when the pool has no threads).  The class declares a private
// task function, ctuTask (DecLibRecon.h:196-197): its definition here is that task, so it may use the class's members like the reference's does.
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
  // ordered behind: the parser (the whole picture: the simplest correct gate) and every picture it references - their samples live in the
  // back-end's DPB, but their FINISHED MOTION (TaskFinishMotionInfo) is what MIDER of this picture reads
  CBarrierVec barriers;
  barriers.push_back( &pcPic->parseDone );
  for( Picture* ref : pcPic->buildAllRefPicsVec() ) if( std::find( barriers.cbegin(), barriers.cend(), &ref->reconDone ) == barriers.cend() ) barriers.push_back( &ref->reconDone );
  commonTaskParam.cs = &cs;
  commonTaskParam.perLineMiHist = std::vector<MotionHist>( pcv->heightInCtus );
  pcPic->reconDone.lock();
  taskFinishPic = FinishPicTaskParam( this, pcPic );
  m_decodeThreadPool->addBarrierTask( TP_TASK_NAME_ARG( "POC:" + std::to_string( pcPic->poc ) + " vvdec_amd picture" )
                                      ctuTask<false>, &taskFinishPic, &pcPic->m_divTasksCounter, &pcPic->reconDone, std::move( barriers ) );
}

template<bool onlyCheckReadyState>
bool DecLibRecon::ctuTask( int tid, void* task_param )
{
  FinishPicTaskParam* param = static_cast<FinishPicTaskParam*>( task_param );
  DecLibRecon&        d     = *param->decLib;
  Picture*            pic   = param->pic;
  CodingStructure&    cs    = *pic->cs;
  AmdInst&            I     = instOf( &d );
  AmdShared&          S     = *I.sh;
  const PreCalcValues& pcv  = *cs.pcv;
  const int numCtu = (int) pcv.sizeInCtus, wCtus = (int) pcv.widthInCtus;
  PerThreadResource&  R     = *d.m_pcThreadResource[std::max( 0, std::min( tid, d.m_numDecThreads - 1 ) )];
  double t0 = nowMs();
  // ---- MIDER (DecLibRecon.cpp:763-805).  A caller that hands over pictures whose motion is already derived (the test harness builds its
  // coding units with final motion vectors and binds the per-CTU motion buffers itself) skips it.
  bool haveMotion = true;
  for( int a = 0; a < numCtu && haveMotion; a++ ) haveMotion = cs.getCtuData( a ).motion != nullptr;
  if( !haveMotion )
    for( int a = 0; a < numCtu; a++ )
    {
      CtuData& cd = cs.getCtuData( a );
      cd.motion = &d.m_motionInfo[pcv.num4x4CtuBlks * a];
      if( !cd.slice->isIntra() || cs.sps->getIBCFlag() )
      {
        const UnitArea ctuArea = getCtuArea( cs, a % wCtus, a / wCtus, true );
        R.m_cCuDecoder.TaskDeriveCtuMotionInfo( cs, a, ctuArea, d.commonTaskParam.perLineMiHist[a / wCtus] );
      }
      else memset( NO_WARNING_class_memaccess( cd.motion ), MI_NOT_VALID, sizeof( MotionInfo ) * pcv.num4x4CtuBlks );
    }
  double t1 = nowMs(); I.msMider += t1 - t0;
  // ---- LF_INIT (DecLibRecon.cpp:807-829): the edge parameters are an input of the back-end
  // (the CTUs are independent - the reference runs one task per CTU; here the picture's task fans out over a few threads of its own.  VVDEC_AMD_HOST_THREADS,
  // default 4: with several pictures in flight the decoder's pool is busy with their tasks)
  // threads of this picture's host work (LF_INIT, the flattening): as many as the decoder's pool has, at least 4 (VVDEC_AMD_HOST_THREADS overrides)
  const int hostThreads = getenv( "VVDEC_AMD_HOST_THREADS" ) ? atoi( getenv( "VVDEC_AMD_HOST_THREADS" ) ) : std::min( 16, std::max( 4, d.m_decodeThreadPool ? d.m_decodeThreadPool->numThreads() : 0 ) );
  vvr_glue::parallelFor( numCtu, hostThreads, [&]( int a )
  {
    CtuData& cd = cs.getCtuData( a );
    cd.lfParam[0] = &d.m_loopFilterParam[pcv.num4x4CtuBlks * ( 2 * a + 0 )];
    cd.lfParam[1] = &d.m_loopFilterParam[pcv.num4x4CtuBlks * ( 2 * a + 1 )];
    memset( cd.lfParam[0], 0, sizeof( LoopFilterParam ) * 2 * pcv.num4x4CtuBlks );
    d.m_cLoopFilter.calcFilterStrengthsCTU( cs, a );
  } );
  double t2 = nowMs(); I.msLfInit += t2 - t1;
  // ---- the back-end of this decoder: created with the first picture (its size, sample format and CTU size are the sequence's)
  int slot = -1, job = -1;
  {
    std::string why;
    if( vvr_glue::checkExpressible( cs, *pic, why ) != VVR_OK ) THROW_RECOVERABLE( "vvdec_amd: " << why );      // never flattened into something it is not
    Slice& slice = *pic->slices[0];
    Reshape* rsp = nullptr;
    bool lmcs = false;                                                                   // (LMCS is a switch of every slice header: the tables are the picture's)
    for( const Slice* sl : pic->slices ) lmcs |= sl->getLmcsEnabledFlag();
    if( cs.sps->getUseReshaper() && lmcs ) rsp = &R.m_cReshaper;                         // (initSlice was called in decompressPicture)
    if( cs.sps->getUseALF() ) for( Slice* sl : pic->slices ) AdaptiveLoopFilter::reconstructCoeffAPSs( *sl );      // (every slice names its own APSs)
    std::lock_guard<std::mutex> lk( S.mu );                                              // (one submitting thread at a time: vvr.h)
    if( !S.ctx )
    {
      vvr_config cfg; memset( &cfg, 0, sizeof( cfg ) );
      cfg.abi_version = VVR_ABI_VERSION;
      cfg.device = getenv( "VVDEC_AMD_DEVICE" ) ? atoi( getenv( "VVDEC_AMD_DEVICE" ) ) : 0;
      cfg.max_width = (uint16_t) cs.sps->getMaxPicWidthInLumaSamples(); cfg.max_height = (uint16_t) cs.sps->getMaxPicHeightInLumaSamples();
      cfg.chroma_format = cs.sps->getChromaFormatIdc() == CHROMA_400 ? 0 : 1; cfg.bit_depth = (uint8_t) cs.sps->getBitDepth();
      cfg.log2_ctu = (uint8_t) getLog2( cs.sps->getMaxCUWidth() );
      S.numSlots = getenv( "VVDEC_AMD_SLOTS" ) ? atoi( getenv( "VVDEC_AMD_SLOTS" ) ) : 48;       // Picture objects the decoder allocates: DPB size + pictures in flight
      cfg.num_slots = (uint8_t) S.numSlots; cfg.num_streams = 4; cfg.host_threads = 0; cfg.read_buffers = 2;           // (this task IS the worker thread of its picture)
      if( vvr_create( &cfg, &S.ctx ) != VVR_OK ) { S.ctx = nullptr; THROW_RECOVERABLE( "vvdec_amd: no MI355X back-end (vvr_create failed)" ); }
    }
    auto slotFor = [&S]( const Picture* p ) -> int
    {
      auto it = S.slotOf.find( p );
      if( it == S.slotOf.end() ) { CHECK( S.nextSlot >= S.numSlots, "vvdec_amd: more Picture objects than DPB slots (VVDEC_AMD_SLOTS)" ); it = S.slotOf.emplace( p, S.nextSlot++ ).first; }
      return it->second;
    };
    slot = slotFor( pic );
    // a reference picture this back-end has not reconstructed - the grey picture the decoder makes up for a missing reference
    // (DecLibParser::prepareUnavailablePicture), a picture handed in from outside - is uploaded from the Picture's own buffers once
    for( Picture* ref : pic->buildAllRefPicsVec() )
      if( S.slotOf.find( ref ) == S.slotOf.end() )
      {
        const int rs = slotFor( ref );
        vvr_slot_picture_size( S.ctx, rs, (int) ref->lwidth(), (int) ref->lheight() );      // (a coded video sequence may change its picture size)
        CPelUnitBuf rb = const_cast<const Picture*>( ref )->getRecoBuf();
        for( size_t c = 0; c < rb.bufs.size(); c++ )
          if( vvr_write_plane( S.ctx, rs, (int) c, reinterpret_cast<const uint16_t*>( rb.bufs[c].buf ), (size_t) rb.bufs[c].stride ) != VVR_OK ) THROW_RECOVERABLE( "vvdec_amd: " << vvr_last_error( S.ctx ) );
      }
    double t2b = nowMs();
    vvr_glue::extractPicture( cs, slice, *pic, rsp, R.m_cTrQuant, [&S]( const Picture* p ) { auto q = S.slotOf.find( p ); return q == S.slotOf.end() ? -1 : q->second; }, slot, I.desc, hostThreads, /* the motion field only where the back-end reads it */ true );
    double t3 = nowMs(); I.msFlatten += t3 - t2b; t2 = t3;
    job = vvr_submit( S.ctx, &I.desc.pic );
    if( job < 0 ) THROW_RECOVERABLE( "vvdec_amd: " << vvr_last_error( S.ctx ) );
  }
  if( vvr_wait( S.ctx, job ) < 0 ) THROW_RECOVERABLE( "vvdec_amd: " << vvr_last_error( S.ctx ) );
  double t4 = nowMs(); I.msDevice += t4 - t2;
  // ---- the picture as the rest of the decoder expects it: planes in the Picture's own buffers (output, hash SEI, film grain)
  {
    PelUnitBuf reco = pic->getRecoBuf();
    uint16_t* dst[3] = { nullptr, nullptr, nullptr }; size_t stride[3] = { 0, 0, 0 };
    for( size_t c = 0; c < reco.bufs.size(); c++ ) { dst[c] = reinterpret_cast<uint16_t*>( reco.bufs[c].buf ); stride[c] = (size_t) reco.bufs[c].stride; }
    // (this picture only - the others in flight are not waited for -, through pinned staging, rows laid out by the picture's host threads)
    if( vvr_read_picture( S.ctx, slot, dst, stride, hostThreads ) != VVR_OK ) THROW_RECOVERABLE( "vvdec_amd: " << vvr_last_error( S.ctx ) );
  }
  double t5 = nowMs(); I.msReadBack += t5 - t4; I.pictures++;
  // ---- DMVR-refined MVs feed the temporal MV prediction of later pictures: through the reference's own finish step (DecCu.cpp:161)
  if( pic->stillReferenced && I.desc.numDmvr )
  {
    I.dmvrOut.resize( 2 * (size_t) I.desc.numDmvr );
    vvr_read_dmvr( S.ctx, job, I.dmvrOut.data(), I.desc.numDmvr );
    for( auto& e : I.desc.dmvrCus )
    {
      CodingUnit& cu = *e.first;
      const int n = std::max( 1, (int) cu.lwidth() >> 4 ) * std::max( 1, (int) cu.lheight() >> 4 );
      for( int k = 0; k < n; k++ ) cs.m_dmvrMvCache[cu.mvdL0SubPuOff + k] = Mv( I.dmvrOut[2 * ( e.second + k )], I.dmvrOut[2 * ( e.second + k ) + 1] );
      cu.setDmvrCondition( true );
    }
  }
  if( pic->stillReferenced ) for( int a = 0; a < numCtu; a++ ) R.m_cCuDecoder.TaskFinishMotionInfo( cs, a, a % wCtus, a / wCtus );
  cs.deallocTempInternals();
  pic->stopProcessingTimer();
  pic->progress = Picture::reconstructed;
  return true;                                                  // (the pool unlocks reconDone)
}
template bool DecLibRecon::ctuTask<false>( int, void* );
template bool DecLibRecon::ctuTask<true>( int, void* );

Picture* DecLibRecon::waitForPrevDecompressedPic()
{
  if( !m_currDecompPic ) return nullptr;
  try
  {
    if( m_decodeThreadPool->numThreads() == 0 )
    {
      m_decodeThreadPool->processTasksOnMainThread();
      CHECK_FATAL( m_currDecompPic->reconDone.isBlocked(), "can't make progress. some dependecy has not been finished" );
    }
    m_currDecompPic->reconDone.wait();
  }
  catch( ... )
  {
    m_currDecompPic->error = true;
    m_currDecompPic->reconDone.setException( std::current_exception() );
  }
  if( m_currDecompPic->error || m_currDecompPic->reconDone.hasException() ) cleanupOnException();
  return std::exchange( m_currDecompPic, nullptr );
}

void DecLibRecon::cleanupOnException() { m_currDecompPic->waitForAllTasks(); }

}   // namespace vvdec
