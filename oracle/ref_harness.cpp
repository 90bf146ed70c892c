// oracle/ref_harness.cpp — TEST INFRASTRUCTURE.  Drives the *unmodified reference decoder classes* (compiled from
// /root/reference by oracle/Makefile into oracle/_ref/) with a synthetic, pre-parsed picture description
// (include/vvr.h).  It is the strongest parity pin available offline: no conformance bitstreams exist in this
// container, but every reconstruction stage of the reference is callable on a hand-built
// CodingStructure, exactly like DecLibRecon::ctuTask does (source/Lib/DecoderLib/DecLibRecon.cpp:732-1110):
//
//   TaskTrafoCtu -> TaskInterCtu -> TaskCriticalIntraKernel -> rspCtuBcw -> loopFilterCTU(VER) -> loopFilterCTU(HOR)
//   -> SAOPrepareCTULine/SAOProcessCTU -> ALF prepareCTU/processCTU -> buffer swap
//
// run here as whole-picture passes in CTU raster order, which satisfies all neighbour-state preconditions of that
// state machine.  The host-only stages (MIDER = MV derivation, and optionally LF_INIT = boundary-strength derivation)
// are replaced by the motion field / edge-parameter tables carried in the description, because they are *inputs* of
// the reconstruction stage.
//
// Nothing in the product (vvdec_amd/) links or loads this file.
#include <sstream>
#include <string>
#include <vector>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <atomic>
#include <array>
#include <list>
#include <map>
#include <functional>
#include <iostream>
#include <algorithm>
#include <chrono>
#include <exception>
#include <unordered_map>
#include <deque>
#include <set>
#include <iterator>
#include <cstring>
#include <cmath>
#include <numeric>
#include <limits>
#define private public
#define protected public
#include "CommonLib/CommonDef.h"
#include "CommonLib/Rom.h"
#include "CommonLib/Slice.h"
#include "CommonLib/Picture.h"
#include "CommonLib/CodingStructure.h"
#include "CommonLib/Unit.h"
#include "CommonLib/UnitTools.h"
#include "CommonLib/TrQuant.h"
#include "CommonLib/InterPrediction.h"
#include "CommonLib/IntraPrediction.h"
#include "CommonLib/Reshape.h"
#include "CommonLib/LoopFilter.h"
#include "CommonLib/SampleAdaptiveOffset.h"
#include "CommonLib/AdaptiveLoopFilter.h"
#include "CommonLib/RdCost.h"
#include "DecoderLib/DecCu.h"
#include "DecoderLib/DecLibRecon.h"
#include "CommonLib/TrQuant_EMT.h"
#include "CommonLib/x86/CommonDefX86.h"
#undef private
#undef protected

#include "CommonLib/SEI_internal.h"
namespace vvdec   // (defined in CommonLib/PicYuvMD5.cpp, declared in no header)
{
uint32_t calcMD5( const CPelUnitBuf& pic, PictureHash& digest, const BitDepths& bitDepths );
uint32_t calcCRC( const CPelUnitBuf& pic, PictureHash& digest, const BitDepths& bitDepths );
uint32_t calcChecksum( const CPelUnitBuf& pic, PictureHash& digest, const BitDepths& bitDepths );
}
#include "../include/vvr.h"
#include "../integration/vvr_extract.h"      // the reference-side glue under test (round trip, see vvref_extract below)
#include "../integration/DecLibReconAmd.h"   // compiled only: keeps the binding of INTEGRATION.md in step with both sides

using namespace vvdec;


#ifndef VVREF_WITH_DROPIN
// The reference's own multithreaded reconstruction of one picture - the CPU path the GPU back-end is measured against (bench.py cpu_baseline).  What
// DecLibRecon::decompressPicture (DecLibRecon.cpp:429-682) does, restated with ONE difference: the CTU tasks start in state LF_INIT instead of MIDER
// (the harness builds coding units that carry their final motion - there is nothing for the motion derivation to derive from - and binds the per-CTU
// motion buffers itself).  Everything else is the reference's: its per-thread tool sets, its SAO / ALF objects and filter buffer, its 15-state CTU
// task DecLibRecon::ctuTask (:732-1110) with the wavefront dependencies between the states, scheduled in its zig-zag order on its ThreadPool
// (Utilities/ThreadPool.cpp), its finish task (swapBufs, reconDone).  Reference pictures come with their borders extended (buildRefPic).
static void decompressFromLfInit( DecLibRecon& r, Picture* pcPic )
{
  r.m_currDecompPic = pcPic;
  CodingStructure& cs = *pcPic->cs;
  pcPic->progress = Picture::reconstructing;
  const SPS* sps = cs.sps.get();
  const PPS* pps = cs.pps.get();
  const PreCalcValues* pcv0 = cs.pcv;
  for( int i = 0; i < r.m_numDecThreads; i++ )
  {
    auto& R = *r.m_pcThreadResource[i];
    if( sps->getUseReshaper() )
    {
      R.m_cReshaper.createDec( sps->getBitDepth() );
      R.m_cReshaper.initSlice( pcPic->slices[0]->getNalUnitLayerId(), *pcPic->slices[0]->getPicHeader(), pcPic->slices[0]->getVPS_nothrow() );
    }
    R.m_cIntraPred.init( sps->getChromaFormatIdc(), sps->getBitDepth() );
    R.m_cInterPred.init( &r.m_cRdCost, sps->getChromaFormatIdc(), sps->getMaxCUHeight() );
    R.m_cTrQuant.init( pcPic );
    R.m_cCuDecoder.init( &R.m_cIntraPred, &R.m_cInterPred, &R.m_cReshaper, &R.m_cTrQuant );
  }
  // (getCompatibleBuffer, :207-234, for a fresh instance: the filter buffer has the picture's geometry)
  if( r.m_fltBuf.bufs.empty() ) r.m_fltBuf.create( cs.picture->chromaFormat, cs.picture->lumaSize(), pcv0->maxCUWidth, cs.picture->margin, MEMORY_ALIGN_DEF_SIZE, true, pcPic->getUserAllocator() );
  const PreCalcValues* pcv = cs.pcv;
  r.m_cSAO.create( pps->getPicWidthInLumaSamples(), pps->getPicHeightInLumaSamples(), sps->getChromaFormatIdc(), sps->getMaxCUWidth(), sps->getMaxCUHeight(),
                   getLog2( sps->getMaxCUWidth() ) - pcv->minCUWidthLog2, (uint32_t) std::max( 0, sps->getBitDepth() - MAX_SAO_TRUNCATED_BITDEPTH ), r.m_fltBuf );
  if( sps->getUseALF() ) r.m_cALF.create( cs.picHeader.get(), sps, pps, r.m_numDecThreads, r.m_fltBuf );
  const ptrdiff_t lumaCtu = (ptrdiff_t) pcv->maxCUHeight * pcv->maxCUWidth;
  const size_t predSize = (size_t) ( lumaCtu + ( isChromaEnabled( pcv->chrFormat ) ? 2 * ( lumaCtu >> 2 ) : 0 ) ) * pcv->sizeInCtus;
  if( predSize != r.m_predBufSize ) { r.m_predBuf.reset( (Pel*) xMalloc( Pel, predSize ) ); r.m_predBufSize = predSize; }
  cs.m_predBuf = r.m_predBuf.get();
  const size_t maxDmvr = (size_t) pcv->num8x8CtuBlks * pcv->sizeInCtus;
  if( maxDmvr != r.m_dmvrMvCacheSize ) { if( r.m_dmvrMvCache ) free( r.m_dmvrMvCache ); r.m_dmvrMvCacheSize = maxDmvr; r.m_dmvrMvCache = (Mv*) malloc( sizeof( Mv ) * maxDmvr ); }
  cs.m_dmvrMvCache = r.m_dmvrMvCache;
  if( r.m_num4x4Elements != (ptrdiff_t) ( pcv->num4x4CtuBlks * pcv->sizeInCtus ) )
  {
    if( r.m_loopFilterParam ) free( r.m_loopFilterParam );
    if( r.m_motionInfo ) free( r.m_motionInfo );
    r.m_num4x4Elements  = pcv->num4x4CtuBlks * pcv->sizeInCtus;
    r.m_loopFilterParam = (LoopFilterParam*) malloc( sizeof( LoopFilterParam ) * r.m_num4x4Elements * 2 );
    r.m_motionInfo      = (MotionInfo*) malloc( sizeof( MotionInfo ) * r.m_num4x4Elements );
  }
  const int wCtus = (int) pcv->widthInCtus, hCtus = (int) pcv->heightInCtus;
  pcPic->startProcessingTimer();
  r.picBarriers.clear();
  const bool allIntra = std::all_of( pcPic->slices.begin(), pcPic->slices.end(), []( const Slice* sl ) { return sl->isIntra(); } );
  const int colsPerTask  = std::max( std::min( wCtus, ( wCtus / std::max( r.m_numDecThreads * ( allIntra ? 2 : 1 ), 1 ) ) + ( allIntra ? 0 : 1 ) ), 1 );      // (:588)
  const int tasksPerLine = wCtus / colsPerTask + !!( wCtus % colsPerTask );
  pcPic->refPicExtDepBarriers.clear();
  const bool doALF = sps->getUseALF() && !AdaptiveLoopFilter::getAlfSkipPic( cs );
  r.commonTaskParam.reset( cs, LF_INIT, tasksPerLine, doALF );                                   // <- MIDER in the reference
  r.tasksFinishMotion = std::vector<LineTaskParam>( hCtus, LineTaskParam{ r.commonTaskParam, -1 } );
  r.tasksCtu          = std::vector<CtuTaskParam >( (size_t) hCtus * tasksPerLine, CtuTaskParam{ r.commonTaskParam, -1, -1, {} } );
  pcPic->reconDone.lock();
  for( int diag = 0; diag < tasksPerLine + hCtus; ++diag )                                       // zig-zag order (:611-650)
  {
    int line = 0;
    for( int col = diag; col >= 0; --col, ++line )
    {
      if( line >= hCtus || col >= tasksPerLine ) continue;
      CtuTaskParam* param    = &r.tasksCtu[(size_t) line * tasksPerLine + col];
      param->taskLine        = line;
      param->taskCol         = col;
      param->ctuStart        = col * colsPerTask;
      param->ctuEnd          = std::min( param->ctuStart + colsPerTask, wCtus );
      param->numColPerTask   = colsPerTask;
      param->numTasksPerLine = tasksPerLine;
      r.m_decodeThreadPool->addBarrierTask( TP_TASK_NAME_ARG( "ctuTask" ) DecLibRecon::ctuTask<false>, param, &pcPic->m_ctuTaskCounter, nullptr, CBarrierVec( r.picBarriers ), DecLibRecon::ctuTask<true> );
    }
  }
  static auto finishTask = []( int, void* p )                                                    // (:653-676)
  {
    FinishPicTaskParam* param = static_cast<FinishPicTaskParam*>( p );
    CodingStructure& pcs = *param->pic->cs;
    if( pcs.sps->getUseALF() && !AdaptiveLoopFilter::getAlfSkipPic( pcs ) ) param->decLib->swapBufs( pcs );
    param->pic->stopProcessingTimer();
    param->pic->progress = Picture::reconstructed;
    return true;
  };
  r.taskFinishPic = FinishPicTaskParam( &r, pcPic );
  r.m_decodeThreadPool->addBarrierTask( TP_TASK_NAME_ARG( "finishPicTask" ) finishTask, &r.taskFinishPic, &pcPic->m_divTasksCounter, &pcPic->reconDone, { pcPic->m_ctuTaskCounter.donePtr() } );
}
#endif

extern "C" {

// flags for vvref_reconstruct
enum {
  VVREF_SIMD            = 1,   // let the reference use its SSE4.1/AVX2 kernels (default: scalar "Core" = the specification)
  VVREF_DERIVE_LFP      = 2,   // run LoopFilter::calcFilterStrengthsCTU and export its table instead of consuming pic->lfp
  VVREF_STOP_AFTER_RECO = 4,   // output after INTRA stage (no in-loop filters)
  VVREF_STOP_AFTER_DBK  = 8,
  VVREF_STOP_AFTER_SAO  = 16,
  VVREF_SPAN_AFFINE     = 32,  // the motion of affine CUs is not taken from the description: PU::setAllAffineMv spans it from the control-point MVs
  VVREF_ROTATE_REF_LISTS = 64, // slices with headers of their own: slice k holds the description's (union) reference lists rotated by k entries, its CUs, motion
                               // and weights renumbered to match - pictures as a decoder sees them, for the extractor to merge again
};

static std::string g_err;
// round trip of the reference-side glue: when set, vvref_reconstruct stops after it has built the reference's objects from the
// description and lets integration/vvr_extract.h turn them back into a description
static vvr_glue::Extracted* g_extractTo = nullptr;
// test hook of the extractor's refusals (vvref_check_expressible): a feature the flat description cannot express is switched on in the
// reference's objects before they are handed to the glue; g_expressible receives what the glue says
static int g_feature = 0, g_expressible = 0; static std::string g_why;
// motion field of the picture after DecCu::TaskFinishMotionInfo (DMVR-refined MVs written back), picture raster 4x4 grid: set by
// vvref_reconstruct_with_motion (the reference's own stages) and by vvref_run_binding (the DecLibRecon replacement on the GPU back-end)
static vvr_motion* g_motionOut = nullptr;
#ifdef VVREF_WITH_BINDING
struct BindingRun { uint16_t* const* out_planes; int numSlots; };
static BindingRun* g_binding = nullptr;
#endif
#ifdef VVREF_WITH_DROPIN
struct DropInRun { uint16_t* const* out_planes; int threads; };
static DropInRun* g_dropin = nullptr;
#endif
static std::function<int( int, int, int, int )> g_unionIdx;      // ( x, y, list, index in the slice's list ) -> index in the description's lists (VVREF_ROTATE_REF_LISTS)
static void dumpMotion( CodingStructure& cs, int W, int Hh, vvr_motion* out )
{
  const int w4 = ( W + 3 ) >> 2, h4 = ( Hh + 3 ) >> 2;
  for( int y = 0; y < h4; y++ ) for( int x = 0; x < w4; x++ )
  {
    const MotionInfo& mi = cs.getMotionInfo( Position( x << 2, y << 2 ) );
    vvr_motion& o = out[(size_t) y * w4 + x]; memset( &o, 0, sizeof( o ) );
    for( int l = 0; l < 2; l++ ) { o.mv[l][0] = mi.mv[l].hor; o.mv[l][1] = mi.mv[l].ver; o.ref_idx[l] = (int8_t) ( g_unionIdx && mi.miRefIdx[l] >= 0 && mi.miRefIdx[l] < MAX_NUM_REF ? g_unionIdx( x << 2, y << 2, l, mi.miRefIdx[l] ) : mi.miRefIdx[l] ); }
  }
}
static bool g_trace = getenv("VVREF_TRACE") != nullptr;
#define TR(...) do { if( g_trace ) { fprintf( stderr, __VA_ARGS__ ); fflush( stderr ); } } while(0)
__attribute__((visibility("default"))) const char* vvref_last_error() { return g_err.c_str(); }

struct RefPicHolder { std::unique_ptr<Picture> pic; };

static void fillPlane( PelBuf dst, const uint16_t* src, int w, int h )
{
  for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ ) dst.at( x, y ) = Pel( src[(size_t)y * w + x] );
}

// Builds everything and reconstructs one picture.
//   ref_planes[slot*3+c]  : tightly packed (stride = plane width) 16-bit planes of the DPB slots the picture references
//   out_planes[c]         : tightly packed output
//   lfp_out[2]            : optional, receives the reference's own LoopFilterParam tables (picture raster 4x4 grid) when VVREF_DERIVE_LFP
//   dmvr_out              : optional, receives m_dmvrMvCache entries (hor,ver) in CU order (cu.dmvr_off)
//   stage_ms[8]           : optional wall time per stage {trafo+inter, intra, rsp, lf_v, lf_h, sao, alf, total}
__attribute__((visibility("default")))
// vvref_reconstruct_threaded: the picture through the reference's own scheduler (ThreadPool + DecLibRecon::ctuTask) on `threads` threads
struct ThreadedRun { int threads; double ms; };
static ThreadedRun* g_threaded = nullptr;
static int g_extraFlags = 0;      // VVREF_* flags for the entry points that have no flags argument (vvref_run_binding / vvref_run_dropin)
extern "C" __attribute__((visibility("default"))) void vvref_set_extra_flags( int f ) { g_extraFlags = f; }

int vvref_reconstruct( const vvr_picture* vp, const uint16_t* const* ref_planes, uint16_t* const* out_planes,
                       vvr_lfp* const* lfp_out, int32_t* dmvr_out, int flags, double* stage_ms )
{
  flags |= g_extraFlags;
  try
  {
    static bool romInit = false;
    if( !romInit ) { initROM(); romInit = true; }

    // picture-level switches (SPS / PPS / PH) are on when any slice uses the tool; the slices then say which of them do (vvr_slice_header)
    vvr_pic_header Hunion = vp->hdr;
    if( vp->slices ) { uint32_t any = 0; for( uint32_t i = 0; i < vp->num_slices; i++ ) any |= vp->slices[i].tool_flags & VVR_SLICE_TOOL_MASK; Hunion.tool_flags = ( Hunion.tool_flags & ~(uint32_t) VVR_SLICE_TOOL_MASK ) | any; }
    const vvr_pic_header& H = Hunion;
    const bool enableOpt  = !!( flags & VVREF_SIMD );
    // same switch as vvdecParams::simd (vvdecimpl.cpp:92-100): SCALAR keeps every function pointer on its "Core" version
    read_x86_extension_flags( enableOpt ? x86_simd::UNDEFINED : x86_simd::SCALAR );
    g_tCoeffOps = TCoeffOps();
    const int  W = H.width, Hh = H.height;
    const ChromaFormat cf = H.chroma_format == 0 ? CHROMA_400 : CHROMA_420;
    const int  ctuSize = 1 << H.log2_ctu;
    const int  bd = H.bit_depth;

    // ------------------------------------------------------------------ parameter sets
    auto spsP = std::make_shared<SPS>(); SPS& sps = *spsP;   // BasePS: shared_from_this needs shared ownership
    sps.setSPSId( 0 );
    sps.setChromaFormatIdc( cf );
    {   // (pic_width_max_in_luma_samples: a sequence with reference picture resampling holds pictures of several sizes)
      int maxW = W, maxH = Hh;
      for( int l = 0; l < 2 && vp->rpr; l++ ) for( int i = 0; i < H.num_ref[l]; i++ ) { maxW = std::max<int>( maxW, vp->rpr->ref[l][i].width ); maxH = std::max<int>( maxH, vp->rpr->ref[l][i].height ); }
      sps.setMaxPicWidthInLumaSamples( maxW );
      sps.setMaxPicHeightInLumaSamples( maxH );
    }
    sps.setBitDepth( bd );
    sps.setQpBDOffset( 6 * ( bd - 8 ) );
    sps.setInternalMinusInputBitDepth( ( H.min_qp_ts - 4 ) / 6 );
    sps.setCTUSize( ctuSize );
    sps.setMaxCUWidth( ctuSize );
    sps.setMaxCUHeight( ctuSize );
    sps.setLog2MinCodingBlockSize( 2 );
    sps.setLog2MaxTbSize( 6 );
    sps.setUseSAO( !!( H.tool_flags & ( VVR_TOOL_SAO_LUMA | VVR_TOOL_SAO_CHROMA ) ) );
    sps.setUseALF( !!( H.tool_flags & VVR_TOOL_ALF ) );
    sps.setUseCCALF( !!( H.tool_flags & VVR_TOOL_CCALF ) );
    sps.setUseReshaper( !!( H.tool_flags & VVR_TOOL_LMCS ) );
    sps.setUseLFNST( !!( H.tool_flags & VVR_TOOL_LFNST ) );
    sps.setUseMTS( true );            // the per-TU transform types are forced below through mtsIdx/implicit rules, see tu setup
    sps.setUseIntraMTS( !( H.tool_flags & VVR_TOOL_IMPLICIT_MTS ) );       // off: implicit MTS for intra luma blocks (SPS::getUseImplicitMTS)
    sps.setUseInterMTS( true );
    sps.setUseSBT( true );
    sps.setUseISP( true );
    sps.setUseMIP( true );
    sps.setUseLMChroma( true );
    sps.setVerCollocatedChromaFlag( !!( H.tool_flags & VVR_TOOL_CCLM_COLLOC ) );
    sps.setUseMRL( true );
    sps.setBDPCMEnabledFlag( true );
    sps.setTransformSkipEnabledFlag( true );
    sps.setLog2MaxTransformSkipBlockSize( 5 );
    sps.setJointCbCrEnabledFlag( true );
    sps.setUseBIO( !!( H.tool_flags & VVR_TOOL_BDOF ) );
    sps.setUseDMVR( !!( H.tool_flags & VVR_TOOL_DMVR ) );
    sps.setUsePROF( !!( H.tool_flags & VVR_TOOL_PROF ) );
    sps.setUseAffine( true );
    sps.setUseAffineType( true );
    sps.setUseBcw( true );
    sps.setUseCiip( true );
    sps.setUseGeo( true );
    sps.setUseMMVD( true );
    sps.setUseSMVD( true );
    sps.setAMVREnabledFlag( true );
    sps.setSBTMVPEnabledFlag( true );
    sps.setScalingListFlag( !!( H.tool_flags & VVR_TOOL_SCALING_LIST ) );
    sps.setDisableScalingMatrixForLfnstBlks( !!( H.tool_flags & VVR_TOOL_SCALING_LIST_NO_LFNST ) );
    sps.setDepQuantEnabledFlag( !!( H.tool_flags & VVR_TOOL_DEP_QUANT ) );
    sps.setIBCFlag( !!( H.tool_flags & VVR_TOOL_IBC ) );              // Picture::finalInit creates the IBC virtual buffers (Picture.cpp:294)
    sps.setMaxTLayers( 1 );
    if( H.ladf_num_intervals )
    {
      sps.setLadfEnabled( true ); sps.setLadfNumIntervals( H.ladf_num_intervals );
      for( int k = 0; k < H.ladf_num_intervals; k++ ) { sps.setLadfQpOffset( H.ladf_qp_offset[k], k ); sps.setLadfIntervalLowerBound( H.ladf_lower_bound[k], k ); }
    }
    {
      ChromaQpMappingTable& t = sps.m_chromaQpMappingTable;
      t.m_qpBdOffset = sps.getQpBDOffset();
      t.m_sameCQPTableForAllChromaFlag = true;
      for( int i = 0; i < MAX_NUM_CQP_MAPPING_TABLES; i++ )
      {
        t.m_chromaQpMappingTables[i].resize( MAX_QP + t.m_qpBdOffset + 1 );
        for( int q = -t.m_qpBdOffset; q <= MAX_QP; q++ ) t.m_chromaQpMappingTables[i][q + t.m_qpBdOffset] = q;   // identity
      }
    }

    auto ppsP = std::make_shared<PPS>(); PPS& pps = *ppsP;
    pps.setPPSId( 0 );
    pps.setSPSId( 0 );
    pps.setPicWidthInLumaSamples( W );
    pps.setPicHeightInLumaSamples( Hh );
    pps.setLog2CtuSize( H.log2_ctu );
    {
      // tile grid as the description's CTU -> tile map says (uniform grids only differ in where the columns / rows are cut)
      const int cX = ( W + ctuSize - 1 ) / ctuSize, cY = ( Hh + ctuSize - 1 ) / ctuSize;
      std::vector<int> colW, rowH;
      if( vp->ctu_tile )
      {
        int run = 1; for( int x = 1; x <= cX; x++ ) { if( x == cX || vp->ctu_tile[x] != vp->ctu_tile[x - 1] ) { colW.push_back( run ); run = 1; } else run++; }
        run = 1; for( int y = 1; y <= cY; y++ ) { if( y == cY || vp->ctu_tile[(size_t) y * cX] != vp->ctu_tile[(size_t) ( y - 1 ) * cX] ) { rowH.push_back( run ); run = 1; } else run++; }
      }
      else { colW.push_back( cX ); rowH.push_back( cY ); }
      pps.setNumExpTileColumns( (uint32_t) colW.size() );
      pps.setNumExpTileRows( (uint32_t) rowH.size() );
      for( int w : colW ) pps.addTileColumnWidth( w );
      for( int h : rowH ) pps.addTileRowHeight( h );
      pps.initTiles();
    }
    // horizontal reference wrap-around (sps_ref_wraparound_enabled_flag / pps_ref_wraparound_enabled_flag + offset)
    sps.setUseWrapAround( H.wrap_offset != 0 ); pps.setUseWrapAround( H.wrap_offset != 0 ); pps.setWrapAroundOffset( H.wrap_offset );
    pps.setLoopFilterAcrossTilesEnabledFlag( !( H.tool_flags & VVR_TOOL_NO_LF_ACROSS_TILES ) );
    pps.setLoopFilterAcrossSlicesEnabledFlag( !( H.tool_flags & VVR_TOOL_NO_LF_ACROSS_SLICES ) );
    pps.setNumSubPics( 1 );
    if( vp->subpics && vp->num_subpics > 1 )
    {
      // sub-picture layout as the SPS signals it (sps_subpic_info_present_flag), one rectangular slice per sub-picture
      // (pps_single_slice_per_subpic_flag): PPS::initRectSliceMap builds the slice map from it and PPS::initSubPic the SubPic objects
      sps.setSubPicInfoPresentFlag( true ); sps.setNumSubPics( (uint8_t) vp->num_subpics );
      for( uint32_t k = 0; k < vp->num_subpics; k++ )
      {
        const vvr_subpic& sp = vp->subpics[k];
        sps.setSubPicCtuTopLeftX( k, sp.x0 >> H.log2_ctu ); sps.setSubPicCtuTopLeftY( k, sp.y0 >> H.log2_ctu );
        sps.setSubPicWidth( k, ( sp.x1 >> H.log2_ctu ) - ( sp.x0 >> H.log2_ctu ) + 1 ); sps.setSubPicHeight( k, ( sp.y1 >> H.log2_ctu ) - ( sp.y0 >> H.log2_ctu ) + 1 );
        sps.setSubPicTreatedAsPicFlag( k, sp.treated_as_pic != 0 ); sps.setLoopFilterAcrossSubpicEnabledFlag( k, sp.lf_across != 0 );
      }
      pps.setRectSliceFlag( true ); pps.setSingleSlicePerSubPicFlag( true );
      pps.setNumSubPics( (uint8_t) vp->num_subpics );
      pps.initRectSliceMap( &sps );
      pps.initSubPic( sps );
    }
    {
      bool wpP = ( H.tool_flags & VVR_TOOL_WP ) && H.slice_type == 1, wpB = ( H.tool_flags & VVR_TOOL_WP ) && H.slice_type == 0;
      if( vp->slices )
      {
        wpP = wpB = false;
        for( uint32_t i = 0; i < vp->num_slices; i++ ) if( vp->slices[i].tool_flags & VVR_TOOL_WP ) { wpP |= vp->slices[i].slice_type == 1; wpB |= vp->slices[i].slice_type == 0; }
      }
      pps.setUseWP( wpP );        // pps_weighted_pred_flag (P slices)
      pps.setWPBiPred( wpB );     // pps_weighted_bipred_flag (B slices)
    }
    // reference picture resampling (vvr_picture.rpr): the scaling window of the current picture's PPS; the reference pictures get PPSs of their own below.
    // The description carries what the prediction reads (ratios, left / top offsets, sizes); the ratios are given to the slices as they are (the parser
    // derives them from the complete windows, Slice::scaleRefPicList / CU::getRprScaling - host glue outside this path)
    const int winUnitX = SPS::getWinUnitX( cf ), winUnitY = SPS::getWinUnitY( cf );
    if( vp->rpr )
    {
      CHECK( vp->rpr->win_left % winUnitX || vp->rpr->win_top % winUnitY, "scaling window offsets are multiples of the chroma sub-sampling" );
      sps.setRprEnabledFlag( true );
      pps.getScalingWindow().setWindow( vp->rpr->win_left / winUnitX, 0, vp->rpr->win_top / winUnitY, 0 );
    }
    pps.pcv = std::make_unique<PreCalcValues>( sps, pps );

    auto ph = std::make_shared<PicHeader>();
    ph->setValid();
    ph->setSPSId( 0 );
    ph->setPPSId( 0 );
    ph->setDisBdofFlag( !( H.tool_flags & VVR_TOOL_BDOF ) );
    ph->setDisDmvrFlag( !( H.tool_flags & VVR_TOOL_DMVR ) );
    ph->setDisProfFlag( !( H.tool_flags & VVR_TOOL_PROF ) );
    ph->setJointCbCrSignFlag( !!( H.tool_flags & VVR_TOOL_JCCR_SIGN ) );
    ph->setLmcsEnabledFlag( !!( H.tool_flags & VVR_TOOL_LMCS ) );
    ph->setLmcsChromaResidualScaleFlag( !!( H.tool_flags & VVR_TOOL_LMCS_CSCALE ) );
    std::shared_ptr<APS> slAps;
    if( ( H.tool_flags & VVR_TOOL_SCALING_LIST ) && vp->scaling )
    {   // scaling list APS as the parser leaves it (APS::m_scalingListApsInfo); Quant::init (Quant.cpp:622) expands it
      slAps = std::make_shared<APS>();
      slAps->setAPSId( 0 ); slAps->setAPSType( SCALING_LIST_APS );
      ScalingList& sl = slAps->getScalingList();
      for( int id = 0; id < 28; id++ )
      {
        const int n = ScalingList::matrixSize( id );
        int* dst = sl.getScalingListAddress( id );
        for( int k = 0; k < n * n; k++ ) dst[k] = vp->scaling->coef[id][k];
        sl.setScalingListDC( id, vp->scaling->dc[id] );
      }
      ph->setExplicitScalingListEnabledFlag( true );
      ph->setScalingListAPS( slAps );
    }
    std::shared_ptr<APS> lmcsAps;
    if( ( H.tool_flags & VVR_TOOL_LMCS ) && vp->lmcs )
    {
      // LMCS APS with the syntax-level model (HLSyntaxReader.cpp:1030-1056); Reshape::initSlice builds its tables from it
      lmcsAps = std::make_shared<APS>();
      lmcsAps->setAPSId( 0 ); lmcsAps->setAPSType( LMCS_APS );
      SliceReshapeInfo& ri = lmcsAps->getReshaperAPSInfo();
      ri.sliceReshaperEnableFlag = true; ri.sliceReshaperModelPresentFlag = true;
      ri.enableChromaAdj = !!( H.tool_flags & VVR_TOOL_LMCS_CSCALE );
      ri.reshaperModelMinBinIdx = vp->lmcs->min_bin; ri.reshaperModelMaxBinIdx = vp->lmcs->max_bin;
      for( int i = 0; i < 16; i++ ) ri.reshaperModelBinCWDelta[i] = vp->lmcs->model_delta_cw[i];
      ri.chrResScalingOffset = vp->lmcs->model_delta_crs;
      ph->setLmcsAPS( lmcsAps );
    }
    // virtual boundaries of the in-loop filters (ph_virtual_boundaries_present_flag, or the SPS's copied into the picture header, HLSyntaxReader.cpp:2924-2970)
    ph->setVirtualBoundariesPresentFlag( ( H.num_ver_vb | H.num_hor_vb ) != 0 );
    ph->setNumVerVirtualBoundaries( H.num_ver_vb ); ph->setNumHorVirtualBoundaries( H.num_hor_vb );
    for( int i = 0; i < H.num_ver_vb; i++ ) ph->setVirtualBoundariesPosX( H.vb_pos_x[i], i );
    for( int i = 0; i < H.num_hor_vb; i++ ) ph->setVirtualBoundariesPosY( H.vb_pos_y[i], i );

    TR("ps done\n");
    // ------------------------------------------------------------------ pictures
    CUChunkCache cuCache; TUChunkCache tuCache;
    const unsigned margin = 16 + ctuSize;

    std::map<int, std::unique_ptr<Picture>> refPics;
    std::vector<std::shared_ptr<PPS>> refPpsStore;
    std::map<int, std::shared_ptr<SPS>> refSpsStore;
    auto getRefPic = [&]( int slot, int poc, const vvr_rpr_ref* rr ) -> Picture*
    {
      auto it = refPics.find( slot );
      if( it != refPics.end() ) return it->second.get();
      std::unique_ptr<Picture> p( new Picture( enableOpt ) );
      const int rW = rr ? rr->width : W, rH = rr ? rr->height : Hh;
      p->create( cf, Size( rW, rH ), ctuSize, margin, 0, nullptr );
      p->poc = poc;
      const int nc = cf == CHROMA_400 ? 1 : 3;
      for( int c = 0; c < nc; c++ )
      {
        const int cw = c ? rW >> 1 : rW, chh = c ? rH >> 1 : rH;
        CHECK( !ref_planes || !ref_planes[slot * 3 + c], "missing reference plane" );
        fillPlane( p->getRecoBuf( ComponentID( c ) ), ref_planes[slot * 3 + c], cw, chh );
      }
      const PPS* refPps = &pps; const SPS* refSps = &sps;
      if( rr )
      {   // the PPS (size, scaling window) and the SPS (chroma sample location) this reference picture was coded with
        auto rp = std::make_shared<PPS>();
        rp->setPPSId( 1 + (int) refPpsStore.size() ); rp->setSPSId( 0 );
        rp->setPicWidthInLumaSamples( rW ); rp->setPicHeightInLumaSamples( rH ); rp->setLog2CtuSize( H.log2_ctu );
        CHECK( rr->win_left % winUnitX || rr->win_top % winUnitY, "scaling window offsets are multiples of the chroma sub-sampling" );
        const bool sameGeometry = rW == W && rH == Hh && rr->win_left == vp->rpr->win_left && rr->win_top == vp->rpr->win_top;
        CHECK( !rr->scaled && ( !sameGeometry || rr->ratio[0] != SCALE_1X.first || rr->ratio[1] != SCALE_1X.second ), "a reference picture of another size / window / ratio is a scaled one" );
        // Picture::isRefScaled compares the complete windows: a scaled reference of the same size and left / top offsets differs in the right offset
        rp->getScalingWindow().setWindow( rr->win_left / winUnitX, rr->scaled && sameGeometry ? 1 : 0, rr->win_top / winUnitY, 0 );
        const int key = ( rr->hor_collocated_chroma ? 1 : 0 ) | ( rr->ver_collocated_chroma ? 2 : 0 );
        if( !refSpsStore.count( key ) )
        {
          auto rs = std::make_shared<SPS>( sps );
          rs->setHorCollocatedChromaFlag( !!rr->hor_collocated_chroma ); rs->setVerCollocatedChromaFlag( !!rr->ver_collocated_chroma );
          refSpsStore[key] = rs;
        }
        refSps = refSpsStore[key].get();
        rp->pcv = std::make_unique<PreCalcValues>( *refSps, *rp );
        refPpsStore.push_back( rp );
        refPps = rp.get();
      }
      {
        const APS* noAps[ALF_CTB_MAX_NUM_APS] = { nullptr };
        p->finalInit( &cuCache, &tuCache, refSps, refPps, ph, noAps, nullptr, nullptr, false );   // gives the picture a CodingStructure (pps/sps/pcv) like a decoded one has
      }
      { Slice* rs = p->allocateNewSlice(); rs->setPOC( poc ); rs->setPicHeader( ph.get() ); rs->setSliceType( I_SLICE ); }
      if( H.wrap_offset )
      {   // the wrap-around copy of a reference picture (DecLibRecon::borderExtPic, DecLibRecon.cpp:262-283; its margins are filled by extendPicBorder below)
        p->createWrapAroundBuf( true, ctuSize );
        p->getRecoBuf( true ).copyFrom( p->getRecoBuf() );
      }
      if( vp->subpics && vp->num_subpics > 1 )
      {   // what the parser and DecLibRecon::createSubPicRefBufs (DecLibRecon.cpp:388-421) do for a reference picture with sub-pictures: the layout, and a
          // copy of every sub-picture with its own replicated border
        p->subPictures.clear();
        for( uint32_t k = 0; k < vp->num_subpics; k++ ) p->subPictures.push_back( pps.getSubPic( k ) );
        p->m_subPicRefBufs.resize( vp->num_subpics );
        for( uint32_t k = 0; k < vp->num_subpics; k++ )
        {
          const SubPic& sp = pps.getSubPic( k );
          const Area area( sp.getSubPicLeft(), sp.getSubPicTop(), sp.getSubPicWidthInLumaSample(), sp.getSubPicHeightInLumaSample() );
          p->m_subPicRefBufs[k].create( p->chromaFormat, Size( area ), ctuSize, p->margin, MEMORY_ALIGN_DEF_SIZE );
          p->m_subPicRefBufs[k].copyFrom( p->getRecoBuf().subBuf( area ) );
          p->extendPicBorderBuf( p->m_subPicRefBufs[k] );
        }
        p->subPicExtStarted = true;
      }
      p->extendPicBorder( true, true, true, true );   // DecLibRecon::borderExtPic (DecLibRecon.cpp:236) does the same through 6 tasks
      p->borderExtStarted = true;
      p->progress = Picture::reconstructed;
      p->dpbReferenceMark = Picture::ShortTerm;
      p->reconDone.unlock();
      Picture* r = p.get();
      refPics[slot] = std::move( p );
      return r;
    };

    Picture pic( enableOpt );
    pic.create( cf, Size( W, Hh ), ctuSize, margin, 0, nullptr );
    pic.poc = H.poc;
    const APS* nullAps[ALF_CTB_MAX_NUM_APS] = { nullptr };

    // ALF APSs: one array of APS objects per parameter set of the description (slices name their set, vvr_slice_header::alf_set;
    // every Slice holds its own APS pointers, Slice.h:2570)
    std::vector<std::shared_ptr<APS>> alfApsStore;
    const int numAlfSets = vp->slices && vp->num_alf_sets > 1 ? (int) vp->num_alf_sets : 1;
    std::vector<std::array<const APS*, ALF_CTB_MAX_NUM_APS>> alfSetApss( numAlfSets );
    for( auto& s : alfSetApss ) s.fill( nullptr );
    const APS** alfApss = alfSetApss[0].data();
    const bool useAlf = !!( H.tool_flags & VVR_TOOL_ALF ) && vp->alf && vp->alf_params;
    for( int set = 0; useAlf && set < numAlfSets; set++ )
    {
      const vvr_alf_params& A = vp->alf_params[set];
      auto clipIdx = [&]( int v ) { for( int i = 0; i < 4; i++ ) if( AdaptiveLoopFilter::m_alfClippVls[bd - 8][i] == v ) return i; THROW_FATAL( "bad ALF clip value " << v ); return 0; };
      for( int a = 0; a < ALF_CTB_MAX_NUM_APS; a++ )
      {
        auto aps = std::make_shared<APS>();
        aps->setAPSId( a );
        aps->setAPSType( ALF_APS );
        AlfSliceParam& p = aps->getMutableAlfAPSParam();
        p.reset();
        // syntax-level parameters chosen such that the reference's own host step reconstructCoeff (AdaptiveLoopFilter.cpp:888)
        // reproduces exactly the final filters of the description
        p.nonLinearFlagLuma = p.nonLinearFlagChroma = true;
        p.numLumaFilters = MAX_NUM_ALF_CLASSES;
        for( int c = 0; c < MAX_NUM_ALF_CLASSES; c++ )
        {
          p.filterCoeffDeltaIdx[c] = c;
          for( int k = 0; k < MAX_NUM_ALF_LUMA_COEFF - 1; k++ )
          {
            p.lumaCoeff[c * MAX_NUM_ALF_LUMA_COEFF + k] = A.luma_coeff[a][c][k];
            p.lumaClipp[c * MAX_NUM_ALF_LUMA_COEFF + k] = clipIdx( A.luma_clip[a][c][k] );
          }
        }
        p.numAlternativesChroma = VVR_ALF_MAX_CHR_ALT;
        for( int alt = 0; alt < VVR_ALF_MAX_CHR_ALT; alt++ ) for( int k = 0; k < MAX_NUM_ALF_CHROMA_COEFF - 1; k++ )
        {
          p.chromaCoeff[alt * MAX_NUM_ALF_CHROMA_COEFF + k] = A.chroma_coeff[alt][k];
          p.chromaClipp[alt * MAX_NUM_ALF_CHROMA_COEFF + k] = clipIdx( A.chroma_clip[alt][k] );
        }
        aps->releaseMutableAlfAPSParam( p );
        CcAlfFilterParam& cc = aps->getCcAlfAPSParam();
        for( int c = 0; c < 2; c++ )
        {
          cc.ccAlfFilterEnabled[c] = true; cc.ccAlfFilterCount[c] = MAX_NUM_CC_ALF_FILTERS;
          for( int f = 0; f < MAX_NUM_CC_ALF_FILTERS; f++ ) { cc.ccAlfFilterIdxEnabled[c][f] = true; for( int k = 0; k < MAX_NUM_CC_ALF_CHROMA_COEFF; k++ ) cc.ccAlfCoeff[c][f][k] = A.ccalf_coeff[c][f][k]; }
        }
        alfApsStore.push_back( aps );
        alfSetApss[set][a] = aps.get();
      }
    }
    pic.finalInit( &cuCache, &tuCache, &sps, &pps, ph, useAlf ? alfApss : nullAps, lmcsAps.get(), slAps.get() );
    CodingStructure& cs = *pic.cs;
    TR("finalInit done\n");

    // one Slice object per slice of the description; with vvr_picture::slices every one carries its own header values, else all those of `hdr`
    int numSlices = 1;
    const int numCtuAll = ( ( W + ctuSize - 1 ) / ctuSize ) * ( ( Hh + ctuSize - 1 ) / ctuSize );
    if( vp->ctu_slice ) for( int a = 0; a < numCtuAll; a++ ) numSlices = std::max( numSlices, vp->ctu_slice[a] + 1 );
    if( vp->slices && (int) vp->num_slices < numSlices ) THROW_FATAL( "ctu_slice names a slice without a header" );
    std::vector<Slice*> slices;
    // VVREF_ROTATE_REF_LISTS: entry i of slice si's list l is entry ( i + rot ) % n of the description's list
    std::vector<std::array<int, 2>> rot( numSlices, std::array<int, 2>{ { 0, 0 } } );
    if( ( flags & VVREF_ROTATE_REF_LISTS ) && vp->slices ) for( int si = 0; si < numSlices; si++ ) for( int l = 0; l < 2; l++ ) rot[si][l] = H.num_ref[l] > 1 ? si % H.num_ref[l] : 0;
    const int ctusXAll = ( W + ctuSize - 1 ) / ctuSize;
    auto sliceIdxAt = [&]( int x, int y ) { return vp->ctu_slice ? (int) vp->ctu_slice[( y / ctuSize ) * ctusXAll + x / ctuSize] : 0; };
    auto toSlice = [&]( int si, int l, int u ) { const int n = H.num_ref[l]; return u < 0 || u >= n ? u : ( u - rot[si][l] + n ) % n; };
    auto toUnion = [&]( int si, int l, int i ) { const int n = H.num_ref[l]; return i < 0 || i >= n ? i : ( i + rot[si][l] ) % n; };
    g_unionIdx = nullptr;
    if( flags & VVREF_ROTATE_REF_LISTS ) g_unionIdx = [=]( int x, int y, int l, int i ) { return toUnion( sliceIdxAt( x, y ), l, i ); };
    for( int si = 0; si < numSlices; si++ )
    {
    vvr_slice_header SH;
    if( vp->slices ) SH = vp->slices[si];
    else
    {
      memset( &SH, 0, sizeof( SH ) );
      SH.tool_flags = H.tool_flags; SH.slice_type = H.slice_type;
      for( int c = 0; c < 3; c++ ) { SH.deblock_beta_offset_div2[c] = H.deblock_beta_offset_div2[c]; SH.deblock_tc_offset_div2[c] = H.deblock_tc_offset_div2[c]; }
    }
    const bool sliceWp = ( SH.tool_flags & VVR_TOOL_WP ) && vp->wp && SH.slice_type != 2;
    if( SH.slice_type != 2 && !sliceWp && ( ( SH.slice_type == 1 && pps.getUseWP() ) || ( SH.slice_type == 0 && pps.getWPBiPred() ) ) )
      THROW_FATAL( "slices of the same type disagree about weighted prediction: the PPS flag holds for all of them" );
    Slice* slice = pic.allocateNewSlice();
    slices.push_back( slice );
    slice->setPicHeader( ph.get() );
    slice->setSliceType( SliceType( SH.slice_type ) );
    slice->setPOC( H.poc );
    slice->setSliceQp( 32 );
    slice->setDefaultClpRng( sps );
    slice->setDepQuantEnabledFlag( !!( SH.tool_flags & VVR_TOOL_DEP_QUANT ) );
    slice->setDeblockingFilterDisable( !!( H.tool_flags & VVR_TOOL_DEBLOCK_OFF ) );
    slice->setDeblockingFilterBetaOffsetDiv2( SH.deblock_beta_offset_div2[0] );
    slice->setDeblockingFilterTcOffsetDiv2( SH.deblock_tc_offset_div2[0] );
    slice->setDeblockingFilterCbBetaOffsetDiv2( SH.deblock_beta_offset_div2[1] );
    slice->setDeblockingFilterCbTcOffsetDiv2( SH.deblock_tc_offset_div2[1] );
    slice->setDeblockingFilterCrBetaOffsetDiv2( SH.deblock_beta_offset_div2[2] );
    slice->setDeblockingFilterCrTcOffsetDiv2( SH.deblock_tc_offset_div2[2] );
    slice->setSaoEnabledFlag( CHANNEL_TYPE_LUMA, !!( H.tool_flags & VVR_TOOL_SAO_LUMA ) );
    slice->setSaoEnabledFlag( CHANNEL_TYPE_CHROMA, !!( H.tool_flags & VVR_TOOL_SAO_CHROMA ) );
    slice->setLmcsEnabledFlag( !!( SH.tool_flags & VVR_TOOL_LMCS ) );
    if( vp->slices && ( SH.tool_flags & VVR_TOOL_LMCS ) && !!( SH.tool_flags & VVR_TOOL_LMCS_CSCALE ) != !!( vp->hdr.tool_flags & VVR_TOOL_LMCS_CSCALE ) )
      THROW_FATAL( "chroma residual scaling is a picture header flag: a slice with LMCS has it when the picture does" );
    slice->setExplicitScalingListUsed( ( SH.tool_flags & VVR_TOOL_SCALING_LIST ) && vp->scaling );
    slice->setIndependentSliceIdx( si );
    slice->resetSliceMap();
    if( !vp->ctu_slice ) slice->addCtusToSlice( 0, pps.pcv->widthInCtus, 0, pps.pcv->heightInCtus, pps.pcv->widthInCtus );
    else for( int a = 0; a < numCtuAll; a++ ) if( vp->ctu_slice[a] == si ) { const int cx = a % (int) pps.pcv->widthInCtus, cy = a / (int) pps.pcv->widthInCtus; slice->addCtusToSlice( cx, cx + 1, cy, cy + 1, pps.pcv->widthInCtus ); }
    // reference picture lists: the description holds the union of the slices' lists and its CUs index that union, so every non-I slice is
    // given the union (a decoder's slices hold their own lists; whoever flattens a picture renumbers the indices, integration/vvr_extract.h)
    for( int l = 0; l < 2; l++ )
    {
      slice->setNumRefIdx( RefPicList( l ), SH.slice_type == 2 ? 0 : H.num_ref[l] );
      for( int i = 0; i < H.num_ref[l] && SH.slice_type != 2; i++ )
      {
        const int u = toUnion( si, l, i );
        Picture* rp = getRefPic( H.ref_slot[l][u], H.ref_poc[l][u], vp->rpr ? &vp->rpr->ref[l][u] : nullptr );
        slice->m_apcRefPicList[l][i]     = rp;
        slice->m_scalingRatio[l][i]      = vp->rpr ? std::pair<int, int>( vp->rpr->ref[l][u].ratio[0], vp->rpr->ref[l][u].ratio[1] ) : SCALE_1X;
        CHECK( vp->rpr && rp->isRefScaled( &pps ) != !!vp->rpr->ref[l][u].scaled, "reference picture: scaled or not, the description and Picture::isRefScaled disagree" );
        slice->m_aiRefPOCList[l][i]      = H.ref_poc[l][u];
        slice->m_bIsUsedAsLongTerm[l][i] = false;
      }
    }
    slice->resetWpScaling();
    if( sliceWp )
    {   // pred_weight_table() as the parser leaves it (Slice::m_weightPredTable)
      const vvr_wp_params& WP = vp->wp[vp->slices && vp->num_wp_sets > 1 ? SH.wp_set : 0];
      for( int l = 0; l < 2; l++ ) for( int i = 0; i < H.num_ref[l]; i++ )
      {
        WPScalingParam* wp = nullptr;
        slice->getWpScaling( RefPicList( l ), i, wp );
        for( int c = 0; c < 3; c++ )
        {
          const vvr_wp_entry& e = WP.e[l][toUnion( si, l, i )][c];
          wp[c].bPresentFlag = e.present != 0; wp[c].uiLog2WeightDenom = WP.log2_denom[c ? 1 : 0]; wp[c].iWeight = e.weight; wp[c].iOffset = e.offset;
        }
      }
    }
    pic.stillReferenced = !!( H.tool_flags & VVR_TOOL_STILL_REF );
    if( useAlf )
    {
      const int set = vp->slices && numAlfSets > 1 ? SH.alf_set : 0;
      if( set >= numAlfSets ) THROW_FATAL( "slice names an ALF parameter set the picture does not have" );
      const vvr_alf_params& A = vp->alf_params[set];
      slice->setAlfApss( alfSetApss[set].data() );
      slice->setAlfEnabledFlag( COMPONENT_Y, true ); slice->setAlfEnabledFlag( COMPONENT_Cb, cf != CHROMA_400 ); slice->setAlfEnabledFlag( COMPONENT_Cr, cf != CHROMA_400 );
      slice->setNumAlfAps( A.num_luma_aps );
      AlfApsIdVec ids; for( int a = 0; a < A.num_luma_aps; a++ ) ids.push_back( a );
      slice->setAlfApsIdsLuma( ids );
      slice->setAlfApsIdChroma( 0 );
      const bool cc = !!( H.tool_flags & VVR_TOOL_CCALF );
      slice->setCcAlfCbEnabledFlag( cc ); slice->setCcAlfCrEnabledFlag( cc ); slice->setCcAlfCbApsId( 0 ); slice->setCcAlfCrApsId( 0 );
      AdaptiveLoopFilter::reconstructCoeffAPSs( *slice );
    }
    }   // slices
    Slice* slice = slices[0];
    auto sliceOfCtu = [&]( int a ) { return vp->ctu_slice ? slices[vp->ctu_slice[a]] : slices[0]; };
    auto tileOfCtu  = [&]( int a ) { return vp->ctu_tile ? (int) vp->ctu_tile[a] : 0; };

    TR("slice done\n");
    // ------------------------------------------------------------------ CTU data, CUs, TUs
    const PreCalcValues& pcv = *cs.pcv;
    const int numCtu = pcv.sizeInCtus;
    const int w4 = ( W + 3 ) >> 2, h4 = ( Hh + 3 ) >> 2;
    const int ctu4 = ctuSize >> 2;
    std::vector<LoopFilterParam> lfpStore( (size_t) pcv.num4x4CtuBlks * numCtu * 2 );
    std::vector<MotionInfo>      miStore ( (size_t) pcv.num4x4CtuBlks * numCtu );
    memset( (void*) lfpStore.data(), 0, lfpStore.size() * sizeof( LoopFilterParam ) );
    memset( (void*) miStore.data(), MI_NOT_VALID, miStore.size() * sizeof( MotionInfo ) );

    for( int a = 0; a < numCtu; a++ )
    {
      CtuData& cd = cs.getCtuData( a );
      cd.slice = sliceOfCtu( a ); cd.pps = &pps; cd.sps = &sps; cd.ph = ph.get();
      cd.motion     = &miStore[(size_t) pcv.num4x4CtuBlks * a];
      cd.lfParam[0] = &lfpStore[(size_t) pcv.num4x4CtuBlks * ( 2 * a + 0 )];
      cd.lfParam[1] = &lfpStore[(size_t) pcv.num4x4CtuBlks * ( 2 * a + 1 )];
      cd.saoParam.reset();
      if( vp->sao )
      {
        const vvr_sao_ctu& s = vp->sao[a];
        for( int c = 0; c < 3; c++ )
        {
          SAOOffset& o = cd.saoParam[c];
          o.reset();
          if( !s.mode[c] ) continue;
          o.modeIdc = SAO_MODE_NEW;
          o.typeIdc = s.type[c];
          o.typeAuxInfo = s.type[c] == SAO_TYPE_BO ? s.band_pos[c] : 0;
          // SAOPrepareCTULine -> reconstructBlkSAOParam -> invertQuantOffsets re-scales "coded" offsets; we give it coded offsets
          // = final offsets >> log2OffsetScale so that its output equals the description (scale is 0 for bitDepth <= 10).
          const int sc = H.log2_sao_offset_scale[c ? 1 : 0];
          if( s.type[c] == SAO_TYPE_BO )
            for( int i = 0; i < 4; i++ ) o.offset[( s.band_pos[c] + i ) % NUM_SAO_BO_CLASSES] = s.offset[c][i] >> sc;
          else
          {
            o.offset[SAO_CLASS_EO_FULL_VALLEY] = s.offset[c][0] >> sc;
            o.offset[SAO_CLASS_EO_HALF_VALLEY] = s.offset[c][1] >> sc;
            o.offset[SAO_CLASS_EO_PLAIN]       = 0;
            o.offset[SAO_CLASS_EO_HALF_PEAK]   = s.offset[c][2] >> sc;
            o.offset[SAO_CLASS_EO_FULL_PEAK]   = s.offset[c][3] >> sc;
          }
        }
      }
      if( vp->alf )
      {
        const vvr_alf_ctu& f = vp->alf[a];
        for( int c = 0; c < 3; c++ ) cd.alfParam.alfCtuEnableFlag[c] = f.enable[c];
        for( int c = 0; c < 2; c++ ) { cd.alfParam.alfCtuAlternative[c] = f.alt[c]; cd.alfParam.ccAlfFilterControl[c] = f.cc_idc[c]; }
        cd.alfParam.alfCtbFilterIndex = f.luma_filter_idx;
      }
    }

    // motion field (the output of MIDER)
    if( vp->motion )
    {
      for( int y = 0; y < h4; y++ ) for( int x = 0; x < w4; x++ )
      {
        const vvr_motion& m = vp->motion[(size_t) y * w4 + x];
        const int ctuA = ( y / ctu4 ) * pcv.widthInCtus + ( x / ctu4 );
        MotionInfo& mi = miStore[(size_t) pcv.num4x4CtuBlks * ctuA + ( y % ctu4 ) * ctu4 + ( x % ctu4 )];
        mi.mv[0] = Mv( m.mv[0][0], m.mv[0][1] );
        mi.mv[1] = Mv( m.mv[1][0], m.mv[1][1] );
        const int si = sliceIdxAt( x << 2, y << 2 );
        mi.miRefIdx[0] = m.ref_idx[0] < 0 ? MI_NOT_VALID : toSlice( si, 0, m.ref_idx[0] );
        mi.miRefIdx[1] = m.ref_idx[1] < 0 ? MI_NOT_VALID : toSlice( si, 1, m.ref_idx[1] );
      }
    }

    TR("ctu data done\n");
    PelUnitBuf reco = cs.getRecoBuf();
    std::vector<CodingUnit*> cuPtrs( vp->num_cu );
    for( uint32_t i = 0; i < vp->num_cu; i++ )
    {
      const vvr_cu& c = vp->cu[i];
      UnitArea ua( cf, Area( c.x, c.y, c.w, c.h ) );
      ChannelType chType = CHANNEL_TYPE_LUMA; TreeType tt = TREE_D;
      if( c.tree == VVR_TREE_LUMA )   { ua.blocks[1] = CompArea(); ua.blocks[2] = CompArea(); tt = TREE_L; if( cf == CHROMA_400 ) tt = TREE_D; }
      if( c.tree == VVR_TREE_CHROMA ) { ua.blocks[0] = CompArea(); tt = TREE_C; chType = CHANNEL_TYPE_CHROMA; }
      if( cf == CHROMA_400 ) ua.blocks.resize( 1 );
      const Position p0 = ua.blocks[chType].pos();
      const int ctuOfCu = ( c.y >> H.log2_ctu ) * (int) pcv.widthInCtus + ( c.x >> H.log2_ctu );
      Slice* cuSlice = sliceOfCtu( ctuOfCu );
      const CodingUnit* cuL = cs.getCURestricted( p0.offset( -1, 0 ), p0, cuSlice->getIndependentSliceIdx(), tileOfCtu( ctuOfCu ), chType );
      const CodingUnit* cuA = cs.getCURestricted( p0.offset( 0, -1 ), p0, cuSlice->getIndependentSliceIdx(), tileOfCtu( ctuOfCu ), chType );
      CodingUnit& cu = cs.addCU( ua, chType, tt, MODE_TYPE_ALL, cuL, cuA );
      cuPtrs[i] = &cu;
      cu.slice = cuSlice; cu.pps = &pps; cu.sps = &sps; cu.tileIdx = tileOfCtu( ctuOfCu );
      cu.qp = c.qp; cu.chromaQpAdj = 0;
      cu.setPredMode( c.pred_mode == VVR_PRED_INTRA ? MODE_INTRA : c.pred_mode == VVR_PRED_IBC ? MODE_IBC : MODE_INTER );
      cu.setRootCbf( !!( c.flags & VVR_CU_ROOT_CBF ) );
      cu.setSkip( !!( c.flags & VVR_CU_SKIP ) );
      cu.setMergeFlag( !!( c.flags & VVR_CU_MERGE ) );
      // (a CU in sub-block merge mode carries the affine flag whether its candidate was an affine one or the SbTMVP one: CABACReader::subblock_merge_flag sets it,
      // DecCu.cpp:746-767 only changes the merge type - and LoopFilter.cpp:920 reads the flag of the CU on the P side.  Found with the parser-fed streams, round 4)
      cu.setAffineFlag( !!( c.flags & ( VVR_CU_AFFINE | VVR_CU_SBTMVP ) ) );
      cu.setAffineType( ( c.flags & VVR_CU_AFFINE_6P ) ? AFFINEMODEL_6PARAM : AFFINEMODEL_4PARAM );
      cu.setCiipFlag( !!( c.flags & VVR_CU_CIIP ) );
      cu.setGeoFlag( !!( c.flags & VVR_CU_GEO ) );
      cu.setMergeType( ( c.flags & VVR_CU_SBTMVP ) ? MRG_TYPE_SUBPU_ATMVP : MRG_TYPE_DEFAULT_N );
      cu.setMipFlag( !!( c.flags & VVR_CU_MIP ) );
      cu.setMipTransposedFlag( !!( c.flags & VVR_CU_MIP_TRANSP ) );
      cu.setSmvdMode( ( c.flags & VVR_CU_SMVD ) ? 1 : 0 );
      cu.setMmvdFlag( !!( c.flags & VVR_CU_MMVD ) );
      cu.intraDir[0] = c.intra_dir[0]; cu.intraDir[1] = c.intra_dir[1];
      cu.setMultiRefIdx( c.multi_ref_idx );
      cu.setIspMode( c.isp_mode );
      cu.setBdpcmMode( c.bdpcm[0] ); cu.setBdpcmModeChroma( c.bdpcm[1] );
      cu.setLfnstIdx( c.lfnst_idx );
      cu.setSbtInfo( c.sbt_info );
      cu.setInterDir( c.inter_dir );
      const int cuSl = sliceIdxAt( c.x, c.y );
      cu.refIdx[0] = toSlice( cuSl, 0, c.ref_idx[0] ); cu.refIdx[1] = toSlice( cuSl, 1, c.ref_idx[1] );
      if( c.pred_mode == VVR_PRED_IBC )
      {   // what the parser leaves for an IBC CU (CABACReader.cpp:930-968, DecCu.cpp:850-870): list 0, no reference index, the CTU row marked
        cu.setInterDir( 1 ); cu.refIdx[0] = MAX_NUM_REF; cu.refIdx[1] = -1;
        cs.hasIbcBlock[c.y >> H.log2_ctu] = 1;
      }
      cu.setBcwIdx( c.pred_mode == VVR_PRED_INTER ? g_BcwInternFwd[c.bcw_idx] : BCW_DEFAULT );   // description uses the weight-table index (2 = default), the reference its "internal domain"
      cu.setImv( c.imv );
      cu.geoSplitDir = c.geo_split_dir;
      {
        uint8_t g[2];
        for( int k = 0; k < 2; k++ ) g[k] = (uint8_t) ( ( c.geo_dir_ref[k] & 0xf0 ) | ( ( c.flags & VVR_CU_GEO ) ? toSlice( cuSl, ( c.geo_dir_ref[k] >> 4 ) == 1 ? 0 : 1, c.geo_dir_ref[k] & 15 ) : ( c.geo_dir_ref[k] & 15 ) ) );
        cu.setInterDirrefIdxGeo0( g[0] ); cu.setInterDirrefIdxGeo1( g[1] );
      }
      for( int l = 0; l < 2; l++ ) for( int k = 0; k < 3; k++ ) cu.mv[l][k] = Mv( c.mv[l][k][0], c.mv[l][k][1] );
      // GPM keeps its two uni-prediction MVs in mv[0][1] / mv[1][1] (InterPrediction::motionCompensationGeo, InterPrediction.cpp:1478,1489)
      if( c.flags & VVR_CU_GEO ) { cu.mv[0][1] = Mv( c.geo_mv[0][0], c.geo_mv[0][1] ); cu.mv[1][1] = Mv( c.geo_mv[1][0], c.geo_mv[1][1] ); }
      if( ( flags & VVREF_SPAN_AFFINE ) && ( c.flags & VVR_CU_AFFINE ) )
      {
        MotionBuf mb = cu.getMotionBuf();
        for( unsigned yy = 0; yy < mb.height; yy++ ) for( unsigned xx = 0; xx < mb.width; xx++ ) { mb.at( xx, yy ).mv[0] = Mv(); mb.at( xx, yy ).mv[1] = Mv(); }
        for( int l = 0; l < 2; l++ ) if( c.ref_idx[l] >= 0 ) PU::setAllAffineMv( cu, cu.mv[l][0], cu.mv[l][1], cu.mv[l][2], RefPicList( l ), false );
      }
      cu.setPlaneCbf( 0, false ); cu.setPlaneCbf( 1, false ); cu.setPlaneCbf( 2, false );

      for( uint32_t t = c.first_tu; t < c.first_tu + c.num_tu; t++ )
      {
        const vvr_tu& vt = vp->tu[t];
        UnitArea ta( cf, Area( vt.x, vt.y, vt.w, vt.h ) );
        if( cf == CHROMA_400 ) ta.blocks.resize( 1 );
        for( int k = 0; k < (int) ta.blocks.size(); k++ ) if( !( vt.comp_mask & ( 1 << k ) ) ) ta.blocks[k] = CompArea();
        // ISP luma sub-partitions narrower than 4 share one chroma block placed in the last TU: comp_mask carries that
        if( ( vt.comp_mask & 6 ) && ( c.isp_mode ) ) { ta.blocks[1] = cu.blocks[1]; ta.blocks[2] = cu.blocks[2]; }
        TransformUnit& tu = cs.addTU( ta, chType, cu );
        tu.cbf = vt.cbf;
        tu.jointCbCr = vt.joint_cbcr;
        tu.chromaQp[0] = vt.qp[1]; tu.chromaQp[1] = vt.qp[2];   // CABACReader.cpp:612-618: QpParam( tu, Cb/Cr ).Qp( false )
        for( int k = 0; k < 3; k++ )
        {
          tu.setMtsIdx( k, vt.mts_idx[k] );
          tu.maxScanPosX[k] = vt.max_scan_x[k];
          tu.maxScanPosY[k] = vt.max_scan_y[k];
          if( vt.cbf & ( 1 << k ) ) cu.setPlaneCbf( k, true );
          if( k && vt.joint_cbcr ) cu.setPlaneCbf( k, true );
        }
        // levels: scatter the packed corner into the reco plane at the TU position (CABACReader.cpp:2457-2478)
        for( int k = 0; k < (int) ta.blocks.size(); k++ )
        {
          if( !ta.blocks[k].valid() ) continue;
          const bool coded = ( vt.cbf >> k ) & 1;
          if( !coded ) continue;
          const CompArea& blk = tu.blocks[k];
          PelBuf dst = reco.bufs[k].subBuf( blk.pos(), blk.size() );
          const int16_t* src = vp->coef + vt.coef_off[k];
          const bool fullBlk = ( k == 0 ? c.bdpcm[0] : c.bdpcm[1] ) != 0;
          const int cw = fullBlk ? blk.width : vt.max_scan_x[k] + 1, chh = fullBlk ? blk.height : vt.max_scan_y[k] + 1;
          for( int y = 0; y < chh; y++ ) for( int x = 0; x < cw; x++ ) dst.at( x, y ) = src[y * cw + x];
        }
      }
    }

    TR("cus done\n");
    // LoopFilterParam tables from the description (unless we let the reference derive them)
    if( !( flags & VVREF_DERIVE_LFP ) && vp->lfp[0] && vp->lfp[1] )
    {
      for( int d = 0; d < 2; d++ ) for( int y = 0; y < h4; y++ ) for( int x = 0; x < w4; x++ )
      {
        const vvr_lfp& s = vp->lfp[d][(size_t) y * w4 + x];
        const int ctuA = ( y / ctu4 ) * pcv.widthInCtus + ( x / ctu4 );
        LoopFilterParam& l = lfpStore[(size_t) pcv.num4x4CtuBlks * ( 2 * ctuA + d ) + ( y % ctu4 ) * ctu4 + ( x % ctu4 )];
        l.qp[0] = s.qp[0]; l.qp[1] = s.qp[1]; l.qp[2] = s.qp[2];
        l.bs = s.bs; l.sideMaxFiltLength = s.side_max_filt_length; l.flags = s.flags;
      }
    }

    // ------------------------------------------------------------------ tools (DecLibRecon::create / decompressPicture set-up)
    RdCost rdCost( enableOpt );
    LoopFilter lf( enableOpt );
    SampleAdaptiveOffset sao( enableOpt );
    AdaptiveLoopFilter alf( enableOpt );
    std::unique_ptr<IntraPrediction> intraPred( new IntraPrediction() );
    std::unique_ptr<InterPrediction> interPred( new InterPrediction() );
    std::unique_ptr<Reshape>         reshaper( new Reshape() );
    std::unique_ptr<TrQuant>         trQuant( new TrQuant( interPred.get() ) );
    DecCu decCu;
    intraPred->init( cf, bd );
    interPred->init( &rdCost, cf, ctuSize, enableOpt );
    trQuant->init( &pic );
    if( sps.getUseReshaper() )
    {
      TR("reshaper create\n");
      reshaper->createDec( sps.getBitDepth() );
      TR("reshaper initSlice aps=%p lmcsflag=%d vp->lmcs=%p\n", (const void*) ph->getLmcsAPS().get(), (int) ph->getLmcsEnabledFlag(), (const void*) vp->lmcs);
      reshaper->initSlice( 0, *ph, nullptr );          // DecLibRecon.cpp:449-453
      TR("reshaper done\n");
    }
    decCu.init( intraPred.get(), interPred.get(), reshaper.get(), trQuant.get() );
    if( g_feature )
    {
      switch( g_feature )
      {
      case 1: sps.setLadfEnabled( true ); sps.setLadfNumIntervals( 6 ); break;                             // (more LADF intervals than the header holds)
      case 2: sps.setUseWrapAround( true ); pps.setUseWrapAround( true ); pps.setWrapAroundOffset( 0 ); break;            // (a period of zero: no conforming stream has it)
      case 3: ph->setVirtualBoundariesPresentFlag( true ); ph->setNumVerVirtualBoundaries( 1 ); ph->setVirtualBoundariesPosX( 12, 0 ); break;     // (not on the 8-sample grid: no conforming stream has it)
      case 4: { Slice* s2 = pic.allocateNewSlice(); s2->setPicHeader( ph.get() ); s2->setSliceType( B_SLICE ); s2->setPOC( H.poc ); s2->setIndependentSliceIdx( numSlices ); s2->setNumRefIdx( REF_PIC_LIST_0, 1 ); s2->m_apcRefPicList[0][0] = nullptr; } break;     // (a further slice that refers to a picture the DPB does not hold)
      case 5: pps.setNumSubPics( 2 ); sps.setUseWrapAround( true ); pps.setUseWrapAround( true ); pps.setWrapAroundOffset( W ); break;      // (sub-pictures together with wrap-around)
      case 6: sps.setUseColorTrans( true ); break;
      case 7: sps.setBitDepth( 12 ); break;
      case 8: pps.setNumTileColumns( 70000 ); break;                                                       // (more tiles than an index holds)
      case 9: pps.setPicWidthInLumaSamples( W / 2 ); sps.setUseWrapAround( true ); pps.setUseWrapAround( true ); pps.setWrapAroundOffset( 128 ); break;      // (the references keep their size - Picture::isRefScaled - and the picture uses wrap-around)
      default: break;
      }
      g_expressible = vvr_glue::checkExpressible( cs, pic, g_why );
      if( g_feature == 9 ) { pps.setPicWidthInLumaSamples( W ); sps.setUseWrapAround( false ); pps.setUseWrapAround( false ); }
      if( g_feature == 8 ) pps.setNumTileColumns( 1 );
      for( int a = 0; a < numCtu; a++ ) { CtuData& cd = cs.getCtuData( a ); cd.motion = nullptr; cd.lfParam[0] = cd.lfParam[1] = nullptr; }
      return 0;
    }
#ifdef VVREF_WITH_BINDING
    if( g_binding )
    {
      // The DecLibRecon replacement of integration/DecLibReconAmd.h, executed: the reference's objects of this picture go through the extractor
      // into vvr_submit of whichever back-end library the process has loaded, vvr_wait, the DMVR delta MVs come back into the reference's motion
      // field (DecCu::TaskFinishMotionInfo).  The reference pictures are uploaded into the DPB slots the binding's SlotPool assigns to them.
      vvr_config cfg; memset( &cfg, 0, sizeof( cfg ) );
      cfg.abi_version = VVR_ABI_VERSION; cfg.device = 0; cfg.max_width = (uint16_t) W; cfg.max_height = (uint16_t) Hh;
      cfg.chroma_format = H.chroma_format; cfg.bit_depth = H.bit_depth; cfg.log2_ctu = H.log2_ctu; cfg.num_slots = (uint8_t) g_binding->numSlots; cfg.num_streams = 2;
      vvr_context* ctx = nullptr;
      const int crc = vvr_create( &cfg, &ctx );
      CHECK( crc != VVR_OK, "vvr_create failed with " << crc );
      try
      {
        vvr_glue::SlotPool pool( cfg.num_slots );
        const int nc = cf == CHROMA_400 ? 1 : 3;
        std::vector<uint16_t> tmp;
        for( auto& kv : refPics )
        {
          const int s = pool.acquire( kv.second.get() );
          for( int c = 0; c < nc; c++ ) CHECK( vvr_write_plane( ctx, s, c, ref_planes[kv.first * 3 + c], (size_t) ( c ? W >> 1 : W ) ) != VVR_OK, vvr_last_error( ctx ) );
        }
        pic.parseDone.unlock();
        vvr_glue::DecLibReconAmd binding;
        binding.create( ctx, &pool );
        binding.decompressPicture( &pic );
        Picture* done = binding.waitForPrevDecompressedPic();
        CHECK( done != &pic || binding.getCurrPic() != nullptr, "the binding did not hand the picture back" );
        for( int c = 0; c < nc; c++ ) CHECK( vvr_read_plane( ctx, pool.slotOf( &pic ), c, g_binding->out_planes[c], (size_t) ( c ? W >> 1 : W ) ) != VVR_OK, vvr_last_error( ctx ) );
        if( g_motionOut ) dumpMotion( cs, W, Hh, g_motionOut );
        binding.destroy();
      }
      catch( ... ) { vvr_destroy( ctx ); for( int a = 0; a < numCtu; a++ ) { CtuData& cd = cs.getCtuData( a ); cd.motion = nullptr; cd.lfParam[0] = cd.lfParam[1] = nullptr; } throw; }
      vvr_destroy( ctx );
      for( int a = 0; a < numCtu; a++ ) { CtuData& cd = cs.getCtuData( a ); cd.motion = nullptr; cd.lfParam[0] = cd.lfParam[1] = nullptr; }
      return 0;
    }
#endif
#ifdef VVREF_WITH_DROPIN
    if( g_dropin )
    {
      // The DROP-IN: the reference's own class vvdec::DecLibRecon, whose member functions this library takes from integration/DecLibReconDropIn.cpp
      // instead of DecoderLib/DecLibRecon.cpp (oracle/Makefile, libvvrefdropin.so), driven the way DecLib::reconPicture drives it (DecLib.cpp:612-636):
      // create( ThreadPool*, instance, upscale ) / decompressPicture / waitForPrevDecompressedPic / destroy, on a pool of g_dropin->threads threads
      // (0: everything on the calling thread, like vvdec_params.threads = 0).  The result is read from the Picture's own buffers - where the rest of
      // the decoder looks for it.
      const int nc = cf == CHROMA_400 ? 1 : 3;
      for( auto& kv : refPics ) { kv.second->reconDone.unlock(); }          // (reference pictures handed in from outside: finished)
      pic.parseDone.unlock();
      {
        ThreadPool pool( g_dropin->threads, "dropin" );
        std::list<DecLibRecon> recon( 2 );                                   // (two instances share one back-end, as in DecLib.h:70)
        unsigned id = 0;
        for( auto& r : recon ) r.create( &pool, id++, false );
        DecLibRecon& rec = recon.front();
        CHECK( rec.waitForPrevDecompressedPic() != nullptr, "an idle instance handed a picture back" );
        rec.decompressPicture( &pic );
        CHECK( rec.getCurrPic() != &pic, "getCurrPic() is not the picture in progress" );
        Picture* done = rec.waitForPrevDecompressedPic();
        CHECK( done != &pic || rec.getCurrPic() != nullptr, "the drop-in did not hand the picture back" );
        if( pic.reconDone.hasException() ) std::rethrow_exception( pic.reconDone.getException() );
        CHECK( pic.progress != Picture::reconstructed || pic.reconDone.isBlocked(), "picture not marked reconstructed" );
        for( int c = 0; c < nc; c++ )
        {
          const CPelBuf b = const_cast<const Picture&>( pic ).getRecoBuf( ComponentID( c ) );
          for( int y = 0; y < (int) b.height; y++ ) for( int x = 0; x < (int) b.width; x++ ) g_dropin->out_planes[c][(size_t) y * b.width + x] = (uint16_t) b.at( x, y );
        }
        if( g_motionOut ) dumpMotion( cs, W, Hh, g_motionOut );
        for( auto& r : recon ) r.destroy();
      }
      for( int a = 0; a < numCtu; a++ ) { CtuData& cd = cs.getCtuData( a ); cd.motion = nullptr; cd.lfParam[0] = cd.lfParam[1] = nullptr; }
      return 0;
    }
#endif
#ifndef VVREF_WITH_DROPIN
    if( g_threaded )
    {
      // the reference's own scheduler on g_threaded->threads threads (0: everything on the calling thread); timed: set-up of the picture's tools,
      // LF_INIT, all CTU tasks, the finish task - what DecLibRecon::decompressPicture + waitForPrevDecompressedPic take for a parsed picture
      for( auto& kv : refPics ) kv.second->reconDone.unlock();
      pic.parseDone.unlock();
      const int nc = cf == CHROMA_400 ? 1 : 3;
      {
        ThreadPool pool( g_threaded->threads, "ref" );
        DecLibRecon rec;
        rec.create( &pool, 0, false );
        const auto t0 = std::chrono::steady_clock::now();
        decompressFromLfInit( rec, &pic );
        Picture* done = rec.waitForPrevDecompressedPic();
        g_threaded->ms = std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count();
        CHECK( done != &pic, "the reference's scheduler did not hand the picture back" );
        if( pic.reconDone.hasException() ) std::rethrow_exception( pic.reconDone.getException() );
        for( int c = 0; c < nc; c++ )
        {
          if( !out_planes[c] ) continue;
          const CPelBuf b = const_cast<const Picture&>( pic ).getRecoBuf( ComponentID( c ) );
          for( int y = 0; y < (int) b.height; y++ ) for( int x = 0; x < (int) b.width; x++ ) out_planes[c][(size_t) y * b.width + x] = (uint16_t) b.at( x, y );
        }
        cs.m_predBuf = nullptr; cs.m_dmvrMvCache = nullptr;
        for( int a = 0; a < numCtu; a++ ) { CtuData& cd = cs.getCtuData( a ); cd.motion = nullptr; cd.lfParam[0] = cd.lfParam[1] = nullptr; }
        rec.destroy();
      }
      return 0;
    }
#endif
    if( g_extractTo )
    {
      { std::string why; CHECK( vvr_glue::checkExpressible( cs, pic, why ) != VVR_OK, why ); }
      auto slotOf = [&]( const Picture* p ) { for( auto& kv : refPics ) if( kv.second.get() == p ) return kv.first; return -1; };
      if( flags & VVREF_DERIVE_LFP ) for( int a = 0; a < numCtu; a++ ) lf.calcFilterStrengthsCTU( cs, a );      // LF_INIT by the reference itself
      vvr_glue::extractPicture( cs, *slice, pic, sps.getUseReshaper() ? reshaper.get() : nullptr, *trQuant, slotOf, H.out_slot, *g_extractTo );
      for( int a = 0; a < numCtu; a++ ) { CtuData& cd = cs.getCtuData( a ); cd.motion = nullptr; cd.lfParam[0] = cd.lfParam[1] = nullptr; }
      return 0;
    }

    PelStorage fltBuf;
    fltBuf.create( cf, Size( W, Hh ), ctuSize, margin, MEMORY_ALIGN_DEF_SIZE, true, nullptr );
    const uint32_t log2SaoOffsetScale = (uint32_t) std::max( 0, bd - MAX_SAO_TRUNCATED_BITDEPTH );
    sao.create( W, Hh, cf, ctuSize, ctuSize, H.log2_ctu - pcv.minCUWidthLog2, log2SaoOffsetScale, fltBuf );
    if( useAlf ) alf.create( ph.get(), &sps, &pps, 1, fltBuf );

    const ptrdiff_t ctuSampleSizeL = ctuSize * ctuSize;
    const ptrdiff_t ctuSampleSize  = ctuSampleSizeL + ( cf == CHROMA_400 ? 0 : 2 * ( ctuSampleSizeL >> 2 ) );
    std::vector<Pel> predBuf( (size_t) ctuSampleSize * numCtu + 64 );
    cs.m_predBuf = predBuf.data();
    std::vector<Mv> dmvrCache( (size_t) pcv.num8x8CtuBlks * numCtu );
    cs.m_dmvrMvCache = dmvrCache.data();

    TR("tools done\n");
    auto now = []{ return std::chrono::steady_clock::now(); };
    auto ms  = []( std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b ){ return std::chrono::duration<double, std::milli>( b - a ).count(); };
    double st[8] = { 0 };
    auto t0 = now();

    // ------------------------------------------------------------------ stages (DecLibRecon::ctuTask)
    if( flags & VVREF_DERIVE_LFP )
      for( int a = 0; a < numCtu; a++ ) lf.calcFilterStrengthsCTU( cs, a );

    auto tA = now();
    for( int a = 0; a < numCtu; a++ )                                    // INTER
    {
      const int col = a % pcv.widthInCtus, line = a / pcv.widthInCtus;
      const UnitArea ctuArea = getCtuArea( cs, col, line, true );
      if( !cs.getCtuData( a ).firstCU ) continue;
      TR("trafo ctu %d\n", a);
      decCu.TaskTrafoCtu( cs, a, ctuArea );
      TR("inter ctu %d\n", a);
      if( !sliceOfCtu( a )->isIntra() ) decCu.TaskInterCtu( cs, a, ctuArea );
    }
    auto tB = now(); st[0] = ms( tA, tB );
    for( int a = 0; a < numCtu; a++ )                                    // INTRA (raster order satisfies the wavefront dependencies)
    {
      const int col = a % pcv.widthInCtus, line = a / pcv.widthInCtus;
      if( !cs.getCtuData( a ).firstCU ) continue;
      decCu.TaskCriticalIntraKernel( cs, a, getCtuArea( cs, col, line, true ) );
    }
    if( sps.getUseReshaper() )                                           // RSP: inverse luma mapping of every CTU (DecLibRecon.cpp:935)
      for( int line = 0; line < (int) pcv.heightInCtus; line++ ) for( int col = 0; col < (int) pcv.widthInCtus; col++ )
        if( cs.getCtuData( col, line ).firstCU ) reshaper->rspCtuBcw( cs, col, line );
    auto tC = now(); st[1] = ms( tB, tC );
    TR("intra done\n");

    bool stop = !!( flags & VVREF_STOP_AFTER_RECO );
    if( !stop )
    {
      for( int line = 0; line < (int) pcv.heightInCtus; line++ ) for( int col = 0; col < (int) pcv.widthInCtus; col++ )
        lf.loopFilterCTU( cs, MAX_NUM_CHANNEL_TYPE, col, line, EDGE_VER );
      auto tD = now(); st[3] = ms( tC, tD );
      for( int line = 0; line < (int) pcv.heightInCtus; line++ ) for( int col = 0; col < (int) pcv.widthInCtus; col++ )
        lf.loopFilterCTU( cs, MAX_NUM_CHANNEL_TYPE, col, line, EDGE_HOR );
      auto tE = now(); st[4] = ms( tD, tE );
      stop = !!( flags & VVREF_STOP_AFTER_DBK );
      if( !stop )
      {
        if( sps.getUseSAO() )
        {
          for( int line = 0; line < (int) pcv.heightInCtus; line++ ) sao.SAOPrepareCTULine( cs, getLineArea( cs, line, true ) );
          for( int line = 0; line < (int) pcv.heightInCtus; line++ ) for( int col = 0; col < (int) pcv.widthInCtus; col++ )
            sao.SAOProcessCTU( cs, getCtuArea( cs, col, line, true ) );
        }
        auto tF = now(); st[5] = ms( tE, tF );
        stop = !!( flags & VVREF_STOP_AFTER_SAO );
        if( !stop && useAlf )
        {
          for( int line = 0; line < (int) pcv.heightInCtus; line++ ) for( int col = 0; col < (int) pcv.widthInCtus; col++ )
            AdaptiveLoopFilter::prepareCTU( cs, col, line );
          for( int line = 0; line < (int) pcv.heightInCtus; line++ ) for( int col = 0; col < (int) pcv.widthInCtus; col++ )
            alf.processCTU( cs, col, line, 0 );
          // DecLibRecon::swapBufs (DecLibRecon.cpp:423)
          pic.m_bufs[PIC_RECONSTRUCTION].swap( fltBuf );
          cs.rebindPicBufs();
          st[6] = ms( tF, now() );
        }
      }
    }
    st[7] = ms( t0, now() );
    if( stage_ms ) memcpy( stage_ms, st, sizeof( st ) );

    // ------------------------------------------------------------------ outputs
    {
      PelUnitBuf out = pic.getRecoBuf();
      const int nc = cf == CHROMA_400 ? 1 : 3;
      for( int c = 0; c < nc; c++ )
      {
        const int cw = c ? W >> 1 : W, chh = c ? Hh >> 1 : Hh;
        if( !out_planes[c] ) continue;
        for( int y = 0; y < chh; y++ ) for( int x = 0; x < cw; x++ ) out_planes[c][(size_t) y * cw + x] = (uint16_t) out.bufs[c].at( x, y );
      }
    }
    if( lfp_out && ( flags & VVREF_DERIVE_LFP ) )
    {
      for( int d = 0; d < 2; d++ ) if( lfp_out[d] ) for( int y = 0; y < h4; y++ ) for( int x = 0; x < w4; x++ )
      {
        const int ctuA = ( y / ctu4 ) * pcv.widthInCtus + ( x / ctu4 );
        const LoopFilterParam& l = lfpStore[(size_t) pcv.num4x4CtuBlks * ( 2 * ctuA + d ) + ( y % ctu4 ) * ctu4 + ( x % ctu4 )];
        vvr_lfp& o = lfp_out[d][(size_t) y * w4 + x];
        o.qp[0] = l.qp[0]; o.qp[1] = l.qp[1]; o.qp[2] = l.qp[2]; o.bs = l.bs; o.side_max_filt_length = l.sideMaxFiltLength; o.flags = l.flags; o.pad[0] = o.pad[1] = 0;
      }
    }
    if( dmvr_out )
    {
      for( uint32_t i = 0; i < vp->num_cu; i++ )
      {
        const vvr_cu& c = vp->cu[i];
        if( c.mc_mode != VVR_MC_DMVR && c.mc_mode != VVR_MC_DMVR_BDOF ) continue;
        const int n = std::max( 1, c.w >> 4 ) * std::max( 1, c.h >> 4 );
        for( int k = 0; k < n; k++ )
        {
          const Mv& m = cs.m_dmvrMvCache[cuPtrs[i]->mvdL0SubPuOff + k];
          dmvr_out[2 * ( c.dmvr_off + k ) + 0] = m.hor; dmvr_out[2 * ( c.dmvr_off + k ) + 1] = m.ver;
        }
      }
    }
    if( g_motionOut )
    {
      // what DecLibRecon does for a picture that is still referenced (MIDER_cont / TaskFinishMotionInfo, DecLibRecon.cpp:1080-1100): the
      // DMVR-refined MVs go into the motion field
      if( pic.stillReferenced ) for( int a = 0; a < numCtu; a++ ) if( cs.getCtuData( a ).firstCU ) decCu.TaskFinishMotionInfo( cs, a, a % pcv.widthInCtus, a / pcv.widthInCtus );
      dumpMotion( cs, W, Hh, g_motionOut );
    }
    cs.m_predBuf = nullptr; cs.m_dmvrMvCache = nullptr;
    for( int a = 0; a < numCtu; a++ ) { CtuData& cd = cs.getCtuData( a ); cd.motion = nullptr; cd.lfParam[0] = cd.lfParam[1] = nullptr; }
    fltBuf.destroy();
    return 0;
  }
  catch( std::exception& e )
  {
    g_err = e.what();
    return -1;
  }
}

// (forces the compiler to check the member functions of the binding; never called)
__attribute__((unused)) static void vvref_compile_check_binding( vvr_glue::DecLibReconAmd* b, Picture* pic ) { b->decompressPicture( pic ); b->waitForPrevDecompressedPic(); }

// decoded picture hash by the reference's own functions (PicYuvMD5.cpp): planes tightly packed, digest = per-component digests back to back
__attribute__((visibility("default")))
int vvref_picture_hash( const uint16_t* const* planes, int W, int Hh, int chroma_format, int bit_depth, int method, uint8_t* digest )
{
  try
  {
    const ChromaFormat cf = chroma_format == 0 ? CHROMA_400 : CHROMA_420;
    PelStorage st; st.create( cf, Area( 0, 0, W, Hh ) );
    const int nc = cf == CHROMA_400 ? 1 : 3;
    for( int c = 0; c < nc; c++ ) fillPlane( st.bufs[c], planes[c], c ? W >> 1 : W, c ? Hh >> 1 : Hh );
    BitDepths bds; bds.recon = bit_depth;
    PictureHash h;
    const CPelUnitBuf buf = st;
    const uint32_t len = method == 0 ? calcMD5( buf, h, bds ) : method == 1 ? calcCRC( buf, h, bds ) : calcChecksum( buf, h, bds );
    for( size_t i = 0; i < h.hash.size(); i++ ) digest[i] = h.hash[i];
    return (int) len;
  }
  catch( std::exception& e ) { g_err = e.what(); return -1; }
}

// description -> the reference's objects -> description (integration/vvr_extract.h).  The result stays valid until the next call.
__attribute__((visibility("default")))
const vvr_picture* vvref_extract( const vvr_picture* vp, const uint16_t* const* ref_planes, uint32_t* num_dmvr, int flags )
{
  static vvr_glue::Extracted E;
  g_extractTo = &E;
  uint16_t* none[3] = { nullptr, nullptr, nullptr };
  const int rc = vvref_reconstruct( vp, ref_planes, none, nullptr, nullptr, flags & ( VVREF_DERIVE_LFP | VVREF_ROTATE_REF_LISTS ), nullptr );
  g_extractTo = nullptr;
  if( rc != 0 ) return nullptr;
  if( num_dmvr ) *num_dmvr = E.numDmvr;
  return &E.pic;
}

// the reference's own stages, plus the motion field after DecCu::TaskFinishMotionInfo (w4 * h4 entries)
__attribute__((visibility("default")))
int vvref_reconstruct_with_motion( const vvr_picture* vp, const uint16_t* const* ref_planes, uint16_t* const* out_planes, vvr_motion* motion_out, int flags )
{
  g_motionOut = motion_out;
  const int rc = vvref_reconstruct( vp, ref_planes, out_planes, nullptr, nullptr, flags, nullptr );
  g_motionOut = nullptr;
  return rc;
}

#ifdef VVREF_WITH_BINDING
// f1 executed: the picture through integration/DecLibReconAmd.h (extractor -> vvr_submit -> vvr_wait -> TaskFinishMotionInfo) on the back-end
// library loaded in this process (libvvdec_amd.so on a GPU box; the stand-in build in the CPU tests).  out_planes / motion_out as above.
__attribute__((visibility("default")))
int vvref_run_binding( const vvr_picture* vp, const uint16_t* const* ref_planes, uint16_t* const* out_planes, vvr_motion* motion_out, int num_slots )
{
  BindingRun run{ out_planes, num_slots };
  g_binding = &run; g_motionOut = motion_out;
  uint16_t* none[3] = { nullptr, nullptr, nullptr };
  const int rc = vvref_reconstruct( vp, ref_planes, none, nullptr, nullptr, VVREF_DERIVE_LFP, nullptr );
  g_binding = nullptr; g_motionOut = nullptr;
  return rc;
}
#endif

#ifdef VVREF_WITH_DROPIN
// one picture through the drop-in implementation of vvdec::DecLibRecon (see above); `threads` = threads of the decoder's pool
__attribute__((visibility("default")))
int vvref_run_dropin( const vvr_picture* vp, const uint16_t* const* ref_planes, uint16_t* const* out_planes, vvr_motion* motion_out, int threads )
{
  DropInRun run{ out_planes, threads };
  g_dropin = &run; g_motionOut = motion_out;
  uint16_t* none[3] = { nullptr, nullptr, nullptr };
  const int rc = vvref_reconstruct( vp, ref_planes, none, nullptr, nullptr, 0, nullptr );
  g_dropin = nullptr; g_motionOut = nullptr;
  return rc;
}
#endif

#ifndef VVREF_WITH_DROPIN
// one picture through the reference's own multithreaded reconstruction (decompressFromLfInit above) on `threads` pool threads; *ms = wall clock of it
__attribute__((visibility("default")))
int vvref_reconstruct_threaded( const vvr_picture* vp, const uint16_t* const* ref_planes, uint16_t* const* out_planes, int threads, double* ms )
{
  ThreadedRun run{ threads, 0.0 };
  g_threaded = &run;
  const int rc = vvref_reconstruct( vp, ref_planes, out_planes, nullptr, nullptr, 0, nullptr );
  g_threaded = nullptr;
  if( ms ) *ms = run.ms;
  return rc;
}
#endif

// the extractor's refusals: the description is turned into reference objects, `feature` switches on something a flat description cannot express
// (0 nothing, 1 LADF, 2 wrap-around, 3 virtual boundaries, 4 a second slice, 5 sub-pictures, 6 ACT, 7 12-bit samples, 8 a second tile column,
// 9 a reference picture of another size); returns what vvr_glue::checkExpressible says (VVR_OK / VVR_ERR_UNSUPPORTED), the reason in `why`
__attribute__((visibility("default")))
int vvref_check_expressible( const vvr_picture* vp, const uint16_t* const* ref_planes, int feature, char* why, int whyLen )
{
  g_feature = feature ? feature : 100; g_expressible = -1; g_why.clear();
  uint16_t* none[3] = { nullptr, nullptr, nullptr };
  const int rc = vvref_reconstruct( vp, ref_planes, none, nullptr, nullptr, 0, nullptr );
  g_feature = 0;
  if( rc != 0 ) { snprintf( why, whyLen, "harness: %s", g_err.c_str() ); return -100; }
  snprintf( why, whyLen, "%s", g_why.c_str() );
  return g_expressible;
}

} // extern "C"
