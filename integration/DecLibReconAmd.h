// integration/DecLibReconAmd.h — the class of INTEGRATION.md §2 as code: same public interface as the reference's DecLibRecon
// (DecoderLib/DecLibRecon.h:185-192), reconstruction done by libvvdec_amd.so.  Like vvr_extract.h it belongs INTO the reference tree; here
// it is compiled by the test harness against the reference's headers and include/vvr.h AND executed by it (oracle/ref_harness.cpp,
// vvref_run_binding): reference-built objects of a picture go through decompressPicture / waitForPrevDecompressedPic, on the stand-in runtime
// in the CPU tests and on the GPU in tests/test_gpu_parity.py, where planes and motion field are compared with the reference's own DecLibRecon
// stages.  What it has not seen is a picture parsed from a bitstream (none are available offline).
#pragma once
#include <unordered_map>
#include <memory>
#include <thread>
#include <exception>
#include "vvr_extract.h"

namespace vvr_glue
{

// DPB slot of every Picture the decoder holds: acquired when PicListManager hands the picture out, released with it
class SlotPool
{
  std::vector<int> m_free; std::unordered_map<const Picture*, int> m_slot;
public:
  explicit SlotPool( int numSlots ) { for( int s = numSlots - 1; s >= 0; s-- ) m_free.push_back( s ); }
  int  acquire( const Picture* p ) { auto it = m_slot.find( p ); if( it != m_slot.end() ) return it->second; CHECK( m_free.empty(), "no free DPB slot" ); const int s = m_free.back(); m_free.pop_back(); m_slot[p] = s; return s; }
  void release( const Picture* p ) { auto it = m_slot.find( p ); if( it == m_slot.end() ) return; m_free.push_back( it->second ); m_slot.erase( it ); }
  int  slotOf( const Picture* p ) const { auto it = m_slot.find( p ); return it == m_slot.end() ? -1 : it->second; }
};

class DecLibReconAmd
{
  vvr_context* m_ctx       = nullptr;      // shared by all instances of one decoder (the DPB lives in it)
  SlotPool*    m_slots     = nullptr;
  Picture*     m_currDecompPic = nullptr;
  int          m_job       = -1;
  Extracted    m_desc;                      // reusable host staging of the flat description
  LoopFilter   m_loopFilter;                // host-side LF_INIT only (calcFilterStrengthsCTU)
  Reshape      m_reshaper;
  TrQuant      m_trQuant;                   // TrQuant::getTrTypes
  DecCu        m_decCu;                     // TaskFinishMotionInfo
  std::vector<Mv>      m_dmvrMvCache;
  std::vector<int32_t> m_dmvrOut;

public:
  DecLibReconAmd() : m_loopFilter( false ), m_trQuant( nullptr ) {}

  void create( vvr_context* ctx, SlotPool* slots ) { m_ctx = ctx; m_slots = slots; }
  void destroy() { m_ctx = nullptr; }
  Picture* getCurrPic() const { return m_currDecompPic; }

  // LF_INIT: the CTUs are independent (the reference runs one task per CTU on its thread pool); here a few threads take every n-th CTU
  void deriveEdgeParameters( CodingStructure& cs, int numCtu )
  {
    const int nt = std::max( 1, std::min<int>( 8, std::min<int>( (int) std::thread::hardware_concurrency(), numCtu / 16 ) ) );
    if( nt == 1 ) { for( int a = 0; a < numCtu; a++ ) m_loopFilter.calcFilterStrengthsCTU( cs, a ); return; }
    std::vector<std::thread> th; std::vector<std::exception_ptr> err( nt );
    for( int t = 0; t < nt; t++ ) th.emplace_back( [&, t]{ try { for( int a = t; a < numCtu; a += nt ) m_loopFilter.calcFilterStrengthsCTU( cs, a ); } catch( ... ) { err[t] = std::current_exception(); } } );
    for( auto& x : th ) x.join();
    for( auto& e : err ) if( e ) std::rethrow_exception( e );
  }

  // DecLibRecon::decompressPicture (DecLibRecon.cpp:429): host-only stages, flatten, submit
  void decompressPicture( Picture* pic )
  {
    CodingStructure& cs = *pic->cs;
    Slice& slice = *pic->slices[0];
    const int numCtu = cs.pcv->sizeInCtus;
    pic->parseDone.wait();                                                             // simplest correct integration
    { std::string why; if( checkExpressible( cs, *pic, why ) != VVR_OK ) THROW_RECOVERABLE( "vvdec_amd: " << why ); }    // never flattened into something it is not
    deriveEdgeParameters( cs, numCtu );                                                // LF_INIT (DecLibRecon.cpp:912-941)
    Reshape* rsp = nullptr;
    bool lmcs = false;                                                                 // (LMCS is a switch of every slice header: the tables are the picture's)
    for( const Slice* sl : pic->slices ) lmcs |= sl->getLmcsEnabledFlag();
    if( cs.sps->getUseReshaper() && lmcs )
    {
      m_reshaper.createDec( cs.sps->getBitDepth() );
      m_reshaper.initSlice( slice.getNalUnitLayerId(), *slice.getPicHeader(), slice.getVPS_nothrow() );   // DecLibRecon.cpp:449-453
      rsp = &m_reshaper;
    }
    if( cs.sps->getUseALF() ) for( Slice* sl : pic->slices ) AdaptiveLoopFilter::reconstructCoeffAPSs( *sl );      // (every slice names its own APSs)
    extractPicture( cs, slice, *pic, rsp, m_trQuant, [this]( const Picture* p ) { return m_slots->slotOf( p ); }, m_slots->acquire( pic ), m_desc );
    m_job = vvr_submit( m_ctx, &m_desc.pic );                                          // asynchronous: the arrays may be reused when it returns
    if( m_job < 0 ) THROW_RECOVERABLE( vvr_last_error( m_ctx ) );
    m_currDecompPic = pic;
  }

  // DecLibRecon::waitForPrevDecompressedPic (DecLibRecon.cpp:694-720)
  Picture* waitForPrevDecompressedPic()
  {
    if( !m_currDecompPic ) return nullptr;
    Picture* pic = m_currDecompPic;
    CodingStructure& cs = *pic->cs;
    if( vvr_wait( m_ctx, m_job ) < 0 ) THROW_RECOVERABLE( vvr_last_error( m_ctx ) );
    if( pic->stillReferenced && m_desc.numDmvr )
    {
      // DMVR-refined MVs feed the temporal MV prediction of later pictures: hand the delta MVs to the reference's own finish step
      // (DecCu::TaskFinishMotionInfo, DecCu.cpp:161), which also builds the co-located motion field
      m_dmvrOut.resize( 2 * (size_t) m_desc.numDmvr );
      vvr_read_dmvr( m_ctx, m_job, m_dmvrOut.data(), m_desc.numDmvr );
      m_dmvrMvCache.assign( (size_t) cs.pcv->num8x8CtuBlks * cs.pcv->sizeInCtus, Mv() );
      cs.m_dmvrMvCache = m_dmvrMvCache.data();
      for( auto& e : m_desc.dmvrCus )
      {
        CodingUnit& cu = *e.first;
        const int n = std::max( 1, (int) cu.lwidth() >> 4 ) * std::max( 1, (int) cu.lheight() >> 4 );
        for( int k = 0; k < n; k++ ) cs.m_dmvrMvCache[cu.mvdL0SubPuOff + k] = Mv( m_dmvrOut[2 * ( e.second + k )], m_dmvrOut[2 * ( e.second + k ) + 1] );
        cu.setDmvrCondition( true );
      }
    }
    if( pic->stillReferenced )
      for( int a = 0; a < (int) cs.pcv->sizeInCtus; a++ ) m_decCu.TaskFinishMotionInfo( cs, a, a % cs.pcv->widthInCtus, a / cs.pcv->widthInCtus );
    cs.m_dmvrMvCache = nullptr;
    pic->progress = Picture::reconstructed;
    pic->reconDone.unlock();
    return std::exchange( m_currDecompPic, nullptr );
  }
};

}   // namespace vvr_glue
