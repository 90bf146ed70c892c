// integration/vvr_extract.h — reference-side glue: walks a picture of the reference decoder after
// parsing, motion derivation (MIDER) and edge-parameter derivation (LF_INIT) and writes the flat description of include/vvr.h that
// vvr_submit() consumes.  It is what a maintainer compiles INTO the reference (next to DecLibRecon): it includes the reference's own
// headers and uses its own helpers for everything that is "derived state" (final intra modes, transform types, the branch
// InterPrediction::motionCompensation takes, CIIP neighbour flags, LMCS tables, final ALF filters, resolved SAO merges).
//
// It is not part of the product library (that one never sees reference types).  In this repository it is compiled only by the test
// harness (oracle/ref_harness.cpp, which needs /root/reference), where a round trip  description -> reference objects -> extractor ->
// description  is checked field by field (tests/test_extractor_roundtrip.py).  Members that are not public in the reference
// (Reshape tables, TrQuant::getTrTypes) are reached the way a member function of those classes would reach them.
#pragma once
#include <vector>
#include <memory>
#include <array>
#include <functional>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <thread>
#include <exception>

namespace vvr_glue
{

using namespace vvdec;

// The ONE place that reads non-public members of a reference class: the LMCS tables of Reshape, which are PROTECTED there.  No patch of the reference and no
// `#define private public`: a class derived from Reshape may name the protected members, and a pointer to member formed through the derived class
// ( &LmcsTables::m_invLUT has the type  Pel* Reshape::* ) applies to any Reshape object - plain C++ ([class.protected]); LmcsTables is never instantiated.
struct LmcsTables : Reshape
{
  template<class T> static const T& get( const Reshape& r, T Reshape::* m ) { return r.*m; }
  static void read( const Reshape& r, int bd, vvr_lmcs_params& L )
  {
    const int lutSize = 1 << bd, orgCW = lutSize / PIC_CODE_CW_BINS, l2cw = getLog2( orgCW );
    const SliceReshapeInfo& ri = const_cast<Reshape&>( r ).getSliceReshaperInfo();
    const Pel* invLUT = get( r, &LmcsTables::m_invLUT );
    const auto& reshapePivot = get( r, &LmcsTables::m_reshapePivot );
    const auto& fwdScaleCoef = get( r, &LmcsTables::m_fwdScaleCoef );
    const auto& inputPivot = get( r, &LmcsTables::m_inputPivot );
    const auto& chromaAdjHelpLUT = get( r, &LmcsTables::m_chromaAdjHelpLUT );
    for( int v = 0; v < lutSize; v++ )
    {
      L.inv_lut[v] = invLUT[v];
      const int i = v >> l2cw;
      L.fwd_lut[v] = (int16_t) Clip3( 0, lutSize - 1, (int) reshapePivot[i] + ( ( (int) fwdScaleCoef[i] * ( v - (int) inputPivot[i] ) + ( 1 << ( FP_PREC - 1 ) ) ) >> FP_PREC ) );
    }
    for( int i = 0; i < 16; i++ ) { L.chroma_scale[i] = (int16_t) chromaAdjHelpLUT[i]; L.model_delta_cw[i] = (int16_t) ri.reshaperModelBinCWDelta[i]; }
    for( int i = 0; i < 17; i++ ) L.pivot[i] = reshapePivot[i];
    L.min_bin = (int16_t) ri.reshaperModelMinBinIdx; L.max_bin = (int16_t) ri.reshaperModelMaxBinIdx; L.model_delta_crs = (int16_t) ri.chrResScalingOffset;
  }
};

// [0, n) over up to `threads` threads (contiguous chunks); the first exception is rethrown.  The per-picture host steps of an integration (LF_INIT,
// the 4x4 tables of the flat description) are independent per CTU / row; a decoder that keeps its own pool busy with other pictures passes 1.
template<class F> static inline void parallelFor( int n, int threads, F&& fn )
{
  threads = std::max( 1, std::min( threads, n ) );
  if( threads == 1 ) { for( int i = 0; i < n; i++ ) fn( i ); return; }
  std::vector<std::thread> th; std::vector<std::exception_ptr> err( threads );
  for( int t = 0; t < threads; t++ ) th.emplace_back( [&, t]{ try { for( int i = (int) ( (int64_t) n * t / threads ); i < (int) ( (int64_t) n * ( t + 1 ) / threads ); i++ ) fn( i ); } catch( ... ) { err[t] = std::current_exception(); } } );
  for( auto& x : th ) x.join();
  for( auto& e : err ) if( e ) std::rethrow_exception( e );
}

// storage of records that are all written by the flattening: grows, is never value-initialised (a std::vector would clear megabytes per picture on
// one thread before the walk's threads write them) and does not keep its contents over a resize
template<class T> struct RawArray
{
  std::unique_ptr<T[]> p; size_t n = 0, cap = 0;
  void resize( size_t k ) { if( k > cap ) { p.reset( new T[k] ); cap = k; } n = k; }
  T* data() { return p.get(); } const T* data() const { return p.get(); }
  size_t size() const { return n; } bool empty() const { return n == 0; }
  T& operator[]( size_t i ) { return p[i]; } const T& operator[]( size_t i ) const { return p[i]; }
  T* begin() { return p.get(); } T* end() { return p.get() + n; } const T* begin() const { return p.get(); } const T* end() const { return p.get() + n; }
};

struct Extracted
{
  vvr_picture               pic;          // pointers into the members below
  RawArray<vvr_cu>          cu;
  RawArray<vvr_tu>          tu;
  RawArray<int16_t>         coef;
  std::vector<uint32_t>     ctuFirstCu;
  std::vector<vvr_motion>   motion;
  std::unique_ptr<vvr_motion[]> motionSparse; size_t motionSparseCells = 0;      // (subBlockMotionOnly) the cells under affine / SbTMVP CUs only: never cleared, pages nobody writes are never touched
  RawArray<vvr_lfp>         lfp[2];
  std::vector<vvr_sao_ctu>  sao;
  std::vector<vvr_alf_ctu>  alf;
  std::vector<vvr_alf_params> alfSets;            // final filters of the APSs the slices refer to: one table per distinct choice (vvr_slice_header::alf_set)
  vvr_lmcs_params           lmcs;
  std::vector<vvr_wp_params> wpSets;              // pred_weight_table() of the slices over the union of their reference lists (vvr_slice_header::wp_set)
  std::vector<vvr_slice_header> slices;           // filled (and pointed to) when the picture has more than one slice
  vvr_scaling_list          scaling;
  vvr_rpr_params            rpr;                  // filled (and pointed to) when a reference picture of the picture is a scaled one
  std::vector<uint16_t>     ctuSlice, ctuTile;    // filled (and pointed to) when the picture has more than one slice / tile
  std::vector<vvr_subpic>   subpics;              // filled (and pointed to) when the picture has more than one sub-picture
  // the CU / TU walk in parts (bands of CTUs, one per thread), merged into cu / tu / coef afterwards
  struct Walk { std::vector<vvr_cu> cu; std::vector<vvr_tu> tu; std::vector<int16_t> coef; std::vector<uint32_t> first; uint32_t numDmvr = 0; std::vector<std::pair<CodingUnit*, uint32_t>> dmvrCus; };
  std::vector<Walk>         walk;
  uint32_t                  numDmvr = 0;
  std::vector<std::pair<CodingUnit*, uint32_t>> dmvrCus;   // CUs that run DMVR with their offset into the delta-MV output (vvr_read_dmvr)
};

// an affine CU: not the SbTMVP CUs, which carry the affine flag of the sub-block merge syntax too (see resolveMcMode)
static inline bool isAffine( const CodingUnit& cu ) { return cu.affineFlag() && !( cu.mergeFlag() && cu.mergeType() == MRG_TYPE_SUBPU_ATMVP ); }

// branch of InterPrediction::motionCompensation (InterPrediction.cpp:1372-1459) for one CU
static inline uint8_t resolveMcMode( const CodingUnit& cu )
{
  if( cu.geoFlag() ) return VVR_MC_GEO;
  // A CU that the parser read as a sub-block merge CU (merge_subblock_flag: CABACReader::subblock_merge_flag sets affineFlag) and whose candidate
  // turned out to be the SbTMVP one (DecCu.cpp:761-767) KEEPS its affine flag; the reference tells it from an affine CU by its merge type only
  // (InterPrediction.cpp:1414,1446: no BDOF, no DMVR, xSubPuMC).  Found with the first parser-fed stream that had such CUs.
  const bool subPu = cu.mergeFlag() && cu.mergeType() == MRG_TYPE_SUBPU_ATMVP;
  if( subPu ) return VVR_MC_SBTMVP;
  if( cu.affineFlag() ) return VVR_MC_AFFINE;
  const Slice& slice = *cu.slice;
  bool bio = false;
  if( cu.sps->getUseBIO() && !cu.cs->picHeader->getDisBdofFlag() && !cu.ciipFlag() && !cu.smvdMode() && !( cu.sps->getUseBcw() && cu.BcwIdx() != BCW_DEFAULT ) )
  {
    const WPScalingParam *wp0 = nullptr, *wp1 = nullptr;
    slice.getWpScaling( REF_PIC_LIST_0, cu.refIdx[0], wp0 );
    slice.getWpScaling( REF_PIC_LIST_1, cu.refIdx[1], wp1 );
    const bool anyWp = wp0[0].bPresentFlag || wp0[1].bPresentFlag || wp0[2].bPresentFlag || wp1[0].bPresentFlag || wp1[1].bPresentFlag || wp1[2].bPresentFlag;
    const bool chk0 = !( anyWp && slice.getSliceType() == B_SLICE ), chk1 = !( cu.pps->getUseWP() && slice.getSliceType() == P_SLICE );
    bio = chk0 && chk1 && PU::isBiPredFromDifferentDirEqDistPoc( cu ) && cu.Y().height >= 8 && cu.Y().width >= 8 && cu.Y().area() >= 128;
  }
  bool dmvr = !subPu && PU::checkDMVRCondition( cu );
  // neither with a scaled reference picture (InterPrediction.cpp:1431-1435)
  const bool refIsScaled = ( cu.refIdx[0] >= 0 && slice.getRefPic( REF_PIC_LIST_0, cu.refIdx[0] )->isRefScaled( cu.pps ) ) || ( cu.refIdx[1] >= 0 && slice.getRefPic( REF_PIC_LIST_1, cu.refIdx[1] )->isRefScaled( cu.pps ) );
  dmvr = dmvr && !refIsScaled; bio = bio && !refIsScaled;
  if( !subPu && bio && !dmvr ) return VVR_MC_BDOF;
  if( dmvr ) return bio ? VVR_MC_DMVR_BDOF : VVR_MC_DMVR;
  if( subPu ) return VVR_MC_SBTMVP;
  if( cu.refIdx[0] < 0 || cu.refIdx[1] < 0 ) return VVR_MC_UNI;
  // xCheckIdenticalMotion (:404): same reference picture and motion in both lists, not with weighted bi-prediction
  if( slice.isInterB() && !cu.pps->getWPBiPred() && slice.getRefPOC( REF_PIC_LIST_0, cu.refIdx[0] ) == slice.getRefPOC( REF_PIC_LIST_1, cu.refIdx[1] ) && cu.mv[0][0] == cu.mv[1][0] ) return VVR_MC_UNI;
  return VVR_MC_BI;
}

static inline uint32_t toolFlags( const CodingStructure& cs, const Slice& slice, const Picture& pic )
{
  const SPS& sps = *cs.sps; const PPS& pps = *cs.pps; const PicHeader& ph = *cs.picHeader;
  uint32_t f = 0;
  if( slice.getSaoEnabledFlag( CHANNEL_TYPE_LUMA ) )   f |= VVR_TOOL_SAO_LUMA;
  if( slice.getSaoEnabledFlag( CHANNEL_TYPE_CHROMA ) ) f |= VVR_TOOL_SAO_CHROMA;
  if( sps.getUseALF() && ( slice.getAlfEnabledFlag( COMPONENT_Y ) || slice.getAlfEnabledFlag( COMPONENT_Cb ) || slice.getAlfEnabledFlag( COMPONENT_Cr ) ) ) f |= VVR_TOOL_ALF;
  if( sps.getUseCCALF() && ( slice.getCcAlfCbEnabledFlag() || slice.getCcAlfCrEnabledFlag() ) ) f |= VVR_TOOL_CCALF;
  if( slice.getLmcsEnabledFlag() ) f |= VVR_TOOL_LMCS;
  if( ph.getLmcsChromaResidualScaleFlag() ) f |= VVR_TOOL_LMCS_CSCALE;
  if( slice.getDeblockingFilterDisable() ) f |= VVR_TOOL_DEBLOCK_OFF;
  if( slice.getDepQuantEnabledFlag() ) f |= VVR_TOOL_DEP_QUANT;
  if( sps.getUseBIO() && !ph.getDisBdofFlag() ) f |= VVR_TOOL_BDOF;
  if( sps.getUseDMVR() && !ph.getDisDmvrFlag() ) f |= VVR_TOOL_DMVR;
  if( sps.getUsePROF() && !ph.getDisProfFlag() ) f |= VVR_TOOL_PROF;
  if( ph.getJointCbCrSignFlag() ) f |= VVR_TOOL_JCCR_SIGN;
  if( pic.stillReferenced ) f |= VVR_TOOL_STILL_REF;
  if( sps.getUseLFNST() ) f |= VVR_TOOL_LFNST;
  if( sps.getUseMTS() ) f |= VVR_TOOL_MTS;
  if( sps.getVerCollocatedChromaFlag() ) f |= VVR_TOOL_CCLM_COLLOC;
  if( ( pps.getUseWP() && slice.getSliceType() == P_SLICE ) || ( pps.getWPBiPred() && slice.getSliceType() == B_SLICE ) ) f |= VVR_TOOL_WP;
  if( slice.getExplicitScalingListUsed() ) f |= VVR_TOOL_SCALING_LIST;
  if( sps.getDisableScalingMatrixForLfnstBlks() ) f |= VVR_TOOL_SCALING_LIST_NO_LFNST;
  if( sps.getUseImplicitMTS() ) f |= VVR_TOOL_IMPLICIT_MTS;
  if( sps.getIBCFlag() ) f |= VVR_TOOL_IBC;
  if( sps.getLadfEnabled() ) f |= VVR_TOOL_LADF;
  if( !pps.getLoopFilterAcrossSlicesEnabledFlag() ) f |= VVR_TOOL_NO_LF_ACROSS_SLICES;
  if( !pps.getLoopFilterAcrossTilesEnabledFlag() ) f |= VVR_TOOL_NO_LF_ACROSS_TILES;
  return f;
}

// The slices of a picture as the description numbers them: one entry per independent slice, in the order they appear (dependent slices
// continue the slice they depend on and share its header).
struct SliceTable
{
  std::vector<const Slice*> first;      // header-carrying Slice object of every entry
  std::vector<int>          ofIdx;      // getIndependentSliceIdx() -> entry
  int entryOf( const Slice& s ) const { return ofIdx[s.getIndependentSliceIdx()]; }
  explicit SliceTable( const Picture& pic )
  {
    for( const Slice* s : pic.slices )
    {
      const size_t k = s->getIndependentSliceIdx();
      if( ofIdx.size() <= k ) ofIdx.resize( k + 1, -1 );
      if( ofIdx[k] < 0 ) { ofIdx[k] = (int) first.size(); first.push_back( s ); }
    }
  }
};

// The reference picture lists of a description are the UNION of the slices' lists (vvr.h: vvr_slice_header): `uni[l]` collects the pictures,
// `map[slice][l][i]` is where entry i of the slice's list l sits in the union.  A picture a slice lists twice (the way streams give one
// picture two sets of prediction weights) takes two entries of the union, so the per-slice weight tables stay addressable by union index.
struct RefUnion
{
  std::vector<const Picture*> uni[2];
  std::vector<std::array<std::array<int8_t, MAX_NUM_REF>, 2>> map;
  RefUnion( const SliceTable& st )
  {
    map.resize( st.first.size() );
    for( size_t k = 0; k < st.first.size(); k++ )
    {
      const Slice& s = *st.first[k];
      for( int l = 0; l < 2; l++ )
      {
        map[k][l].fill( -1 );
        if( s.isIntra() ) continue;
        std::vector<char> used( uni[l].size(), 0 );
        for( int i = 0; i < s.getNumRefIdx( RefPicList( l ) ) && i < MAX_NUM_REF; i++ )
        {
          const Picture* rp = s.getRefPic( RefPicList( l ), i );
          size_t j = 0;
          while( j < uni[l].size() && ( uni[l][j] != rp || used[j] ) ) j++;
          if( j == uni[l].size() ) { uni[l].push_back( rp ); used.push_back( 0 ); }
          used[j] = 1; map[k][l][i] = (int8_t) std::min<size_t>( j, 127 );
        }
      }
    }
  }
};

// What a vvr_picture of this ABI version cannot express: such a picture must not be flattened (it would be reconstructed silently wrong).
// Returns VVR_OK, or VVR_ERR_UNSUPPORTED with the reason; the binding (DecLibReconAmd) turns that into the reference's own
// "not supported" error path.  Slices and tiles are carried as per-CTU indices (the reference restricts intra availability and the in-loop
// filters at their edges: CodingStructure::getCURestricted, SampleAdaptiveOffset.cpp:741-830, AdaptiveLoopFilter.cpp:118-291,
// LoopFilter.cpp:1078-1088) and every slice brings its own header (vvr_picture.slices); the reference picture lists of the slices are merged.
static inline int checkExpressible( const CodingStructure& cs, const Picture& pic, std::string& why )
{
  const SPS& sps = *cs.sps; const PPS& pps = *cs.pps; const PicHeader& ph = *cs.picHeader;
  if( sps.getChromaFormatIdc() != CHROMA_400 && sps.getChromaFormatIdc() != CHROMA_420 ) { why = "chroma format other than 4:0:0 / 4:2:0"; return VVR_ERR_UNSUPPORTED; }
  if( sps.getBitDepth() > 10 || sps.getBitDepth() < 8 ) { why = "bit depth outside 8..10"; return VVR_ERR_UNSUPPORTED; }
  if( sps.getLadfEnabled() && sps.getLadfNumIntervals() > 5 ) { why = "LADF with more than 5 intervals"; return VVR_ERR_UNSUPPORTED; }
  if( pps.getUseWrapAround() && ( pps.getWrapAroundOffset() == 0 || pps.getWrapAroundOffset() > 65535 || ( pps.getWrapAroundOffset() & 7 ) ) ) { why = "horizontal wrap-around motion compensation with a period off the 8-sample grid"; return VVR_ERR_UNSUPPORTED; }
  // virtual boundaries: the picture header holds the effective ones (its own or the SPS's, HLSyntaxReader.cpp:2924-2970), at most three per direction
  if( ph.getVirtualBoundariesPresentFlag() )
  {
    if( ph.getNumVerVirtualBoundaries() > 3 || ph.getNumHorVirtualBoundaries() > 3 ) { why = "more than three virtual boundaries per direction"; return VVR_ERR_UNSUPPORTED; }
    for( unsigned i = 0; i < ph.getNumVerVirtualBoundaries(); i++ ) if( ph.getVirtualBoundariesPosX( i ) & 7 ) { why = "virtual boundary off the 8-sample grid"; return VVR_ERR_UNSUPPORTED; }
    for( unsigned i = 0; i < ph.getNumHorVirtualBoundaries(); i++ ) if( ph.getVirtualBoundariesPosY( i ) & 7 ) { why = "virtual boundary off the 8-sample grid"; return VVR_ERR_UNSUPPORTED; }
  }
  if( sps.getUseColorTrans() ) { why = "adaptive colour transform"; return VVR_ERR_UNSUPPORTED; }
  if( pic.slices.empty() ) { why = "picture without a slice"; return VVR_ERR_UNSUPPORTED; }
  if( pps.getNumSubPics() > 1 && pps.getUseWrapAround() ) { why = "sub-pictures together with reference wrap-around"; return VVR_ERR_UNSUPPORTED; }
  if( pps.getNumSubPics() > 255 ) { why = "more than 255 sub-pictures"; return VVR_ERR_UNSUPPORTED; }
  // several slices and tiles are expressible (vvr_picture.ctu_slice / ctu_tile, one vvr_slice_header per slice)
  if( pic.slices.size() > 65535 || pps.getNumTiles() > 65535 ) { why = "more slices or tiles than a 16-bit index holds"; return VVR_ERR_UNSUPPORTED; }
  const SliceTable st( pic );
  if( st.first.size() > 256 ) { why = "more than 256 slices with headers of their own"; return VVR_ERR_UNSUPPORTED; }
  for( const Slice* sp : st.first )
  {
    const Slice& slice = *sp;
    if( !slice.isIntra() )
      for( int l = 0; l < 2; l++ ) for( int i = 0; i < slice.getNumRefIdx( RefPicList( l ) ); i++ )
      {
        const Picture* ref = slice.getRefPic( RefPicList( l ), i );
        if( !ref ) { why = "missing reference picture"; return VVR_ERR_UNSUPPORTED; }
        if( ref->isRefScaled( &pps ) )
        {
          // reference picture resampling is expressible (vvr_picture.rpr), not together with what the reference itself does not combine it with
          if( pps.getUseWrapAround() ) { why = "scaled reference picture together with reference wrap-around"; return VVR_ERR_UNSUPPORTED; }
          for( int k = 0; k < (int) pps.getNumSubPics() && pps.getNumSubPics() > 1; k++ ) if( pps.getSubPic( k ).getTreatedAsPicFlag() ) { why = "scaled reference picture together with sub-pictures treated as pictures"; return VVR_ERR_UNSUPPORTED; }
          if( ref->lwidth() > 65535 || ref->lheight() > 65535 ) { why = "reference picture larger than 65535 samples"; return VVR_ERR_UNSUPPORTED; }
        }
        if( i >= VVR_MAX_REFS ) { why = "more reference pictures than VVR_MAX_REFS"; return VVR_ERR_UNSUPPORTED; }
      }
  }
  const RefUnion ru( st );
  if( ru.uni[0].size() > VVR_MAX_REFS || ru.uni[1].size() > VVR_MAX_REFS ) { why = "the slices' reference picture lists together hold more pictures than VVR_MAX_REFS"; return VVR_ERR_UNSUPPORTED; }
  return VVR_OK;
}

// slotOf: DPB slot of a reference picture (the caller owns the mapping picture <-> slot); outSlot: slot of the picture itself
static inline void extractPicture( CodingStructure& cs, Slice& slice, Picture& pic, Reshape* reshaper, TrQuant& trQuant,
                                   const std::function<int( const Picture* )>& slotOf, int outSlot, Extracted& E, int threads = 1, bool subBlockMotionOnly = false,
                                   bool lfpOnDevice = false /* VVR_TOOL_LFP_ON_DEVICE: the back-end derives the edge parameters itself - LF_INIT need not have run, no table is copied */ )
{
  const SPS& sps = *cs.sps; const PPS& pps = *cs.pps; const PreCalcValues& pcv = *cs.pcv;
  const int W = pps.getPicWidthInLumaSamples(), H = pps.getPicHeightInLumaSamples();
  const int w4 = ( W + 3 ) >> 2, h4 = ( H + 3 ) >> 2, ctu4 = pcv.maxCUWidth >> 2, numCtu = pcv.sizeInCtus;
  const bool chroma = pcv.chrFormat != CHROMA_400;
  const int nComp = chroma ? 3 : 1;
  const int bd = sps.getBitDepth(), qpBd = sps.getQpBDOffset();

  // ---- header
  vvr_pic_header& h = E.pic.hdr; memset( &E.pic, 0, sizeof( E.pic ) );
  h.abi_version = VVR_ABI_VERSION;
  // what a slice header can set differently goes into E.slices (one vvr_slice_header per slice, below); the picture's own header holds the
  // first slice's values, with a tool on when any slice uses it, and the union of the slices' reference picture lists
  const SliceTable st( pic );
  const RefUnion   ru( st );
  const bool multi = st.first.size() > 1;
  h.tool_flags = toolFlags( cs, slice, pic );
  h.slice_type = (uint8_t) slice.getSliceType();
  bool allDbkOff = slice.getDeblockingFilterDisable();
  if( multi )
  {
    allDbkOff = true;
    for( const Slice* s : st.first )
    {
      h.tool_flags |= toolFlags( cs, *s, pic ) & ( VVR_SLICE_TOOL_MASK | VVR_TOOL_SAO_LUMA | VVR_TOOL_SAO_CHROMA | VVR_TOOL_ALF | VVR_TOOL_CCALF );
      allDbkOff &= s->getDeblockingFilterDisable();
      h.slice_type = std::min<uint8_t>( h.slice_type, (uint8_t) s->getSliceType() );          // B < P < I: the picture is of the most general kind
    }
    h.tool_flags = ( h.tool_flags & ~(uint32_t) VVR_TOOL_DEBLOCK_OFF ) | ( allDbkOff ? VVR_TOOL_DEBLOCK_OFF : 0 );
  }
  h.width = (uint16_t) W; h.height = (uint16_t) H; h.chroma_format = chroma ? 1 : 0; h.bit_depth = (uint8_t) bd;
  h.log2_ctu = (uint8_t) getLog2( pcv.maxCUWidth ); h.poc = slice.getPOC(); h.out_slot = (int16_t) outSlot;
  for( int l = 0; l < 2; l++ )
  {
    h.num_ref[l] = (int8_t) ru.uni[l].size();
    for( int i = 0; i < h.num_ref[l]; i++ ) { h.ref_poc[l][i] = ru.uni[l][i]->getPOC(); h.ref_slot[l][i] = (int16_t) slotOf( ru.uni[l][i] ); }
  }
  // reference picture resampling: how the picture sees its reference pictures (Slice::scaleRefPicList has set the ratios, Slice.cpp:1819); the table
  // is there when one of them is a scaled picture
  {
    bool any = false;
    for( int l = 0; l < 2; l++ ) for( const Picture* rp : ru.uni[l] ) any |= rp->isRefScaled( &pps );
    if( any )
    {
      const int ux = SPS::getWinUnitX( sps.getChromaFormatIdc() ), uy = SPS::getWinUnitY( sps.getChromaFormatIdc() );
      memset( &E.rpr, 0, sizeof( E.rpr ) );
      E.rpr.win_left = pps.getScalingWindow().getWindowLeftOffset() * ux; E.rpr.win_top = pps.getScalingWindow().getWindowTopOffset() * uy;
      for( int l = 0; l < 2; l++ ) for( int j = 0; j < h.num_ref[l]; j++ )
      {
        const Picture* rp = ru.uni[l][j];
        vvr_rpr_ref& r = E.rpr.ref[l][j];
        // (the ratio of a reference picture is the same in every slice that lists it: it follows from the two PPSs)
        for( size_t k = 0; k < st.first.size(); k++ ) for( int i = 0; i < MAX_NUM_REF; i++ )
          if( ru.map[k][l][i] == j ) { const auto& sr = st.first[k]->getScalingRatio( RefPicList( l ), i ); r.ratio[0] = sr.first; r.ratio[1] = sr.second; }
        const PPS* rpps = rp->slices[0]->getPPS();
        r.win_left = rpps->getScalingWindow().getWindowLeftOffset() * ux; r.win_top = rpps->getScalingWindow().getWindowTopOffset() * uy;
        r.width = (uint16_t) rp->lwidth(); r.height = (uint16_t) rp->lheight();
        r.scaled = rp->isRefScaled( &pps ) ? 1 : 0;
        r.hor_collocated_chroma = rp->cs->sps->getHorCollocatedChromaFlag() ? 1 : 0; r.ver_collocated_chroma = rp->cs->sps->getVerCollocatedChromaFlag() ? 1 : 0;
        if( !r.scaled ) { r.ratio[0] = r.ratio[1] = 1 << SCALE_RATIO_BITS; r.win_left = E.rpr.win_left; r.win_top = E.rpr.win_top; }     // (same size and window: Picture::isRefScaled)
      }
      E.pic.rpr = &E.rpr;
    }
  }
  auto sliceDbk = [&]( const Slice& s, int8_t beta[3], int8_t tc[3] )
  {
    beta[0] = (int8_t) s.getDeblockingFilterBetaOffsetDiv2();   tc[0] = (int8_t) s.getDeblockingFilterTcOffsetDiv2();
    beta[1] = (int8_t) s.getDeblockingFilterCbBetaOffsetDiv2(); tc[1] = (int8_t) s.getDeblockingFilterCbTcOffsetDiv2();
    beta[2] = (int8_t) s.getDeblockingFilterCrBetaOffsetDiv2(); tc[2] = (int8_t) s.getDeblockingFilterCrTcOffsetDiv2();
  };
  sliceDbk( slice, h.deblock_beta_offset_div2, h.deblock_tc_offset_div2 );
  // index of a reference picture in the union, for a CU / a motion record of slice entry `se`
  auto uniIdx = [&]( int se, int l, int refIdx ) -> int8_t { return refIdx < 0 || refIdx >= MAX_NUM_REF ? (int8_t) -1 : ru.map[se][l][refIdx]; };
  h.log2_sao_offset_scale[0] = h.log2_sao_offset_scale[1] = (uint8_t) std::max( 0, bd - MAX_SAO_TRUNCATED_BITDEPTH );
  h.min_qp_ts = (int8_t) ( 4 + 6 * sps.getInternalMinusInputBitDepth() );
  if( pps.getUseWrapAround() ) h.wrap_offset = (uint16_t) pps.getWrapAroundOffset();       // (Picture::isWrapAroundEnabled: scaled reference pictures in a picture with wrap-around are refused above)
  if( cs.picHeader->getVirtualBoundariesPresentFlag() )
  {
    const PicHeader& ph = *cs.picHeader;
    h.num_ver_vb = (uint8_t) ph.getNumVerVirtualBoundaries(); h.num_hor_vb = (uint8_t) ph.getNumHorVirtualBoundaries();
    for( int i = 0; i < h.num_ver_vb; i++ ) h.vb_pos_x[i] = (uint16_t) ph.getVirtualBoundariesPosX( i );
    for( int i = 0; i < h.num_hor_vb; i++ ) h.vb_pos_y[i] = (uint16_t) ph.getVirtualBoundariesPosY( i );
  }
  if( sps.getLadfEnabled() )
  {
    h.ladf_num_intervals = (uint8_t) sps.getLadfNumIntervals();
    for( int k = 0; k < sps.getLadfNumIntervals(); k++ ) { h.ladf_qp_offset[k] = (int8_t) sps.getLadfQpOffset( k ); h.ladf_lower_bound[k] = (int16_t) sps.getLadfIntervalLowerBound( k ); }
  }

  const auto tX0 = std::chrono::steady_clock::now();
  // ---- coding units, transform units, levels
  E.cu.resize( 0 ); E.tu.resize( 0 ); E.coef.resize( 0 ); E.ctuFirstCu.assign( numCtu + 1, 0 ); E.numDmvr = 0; E.dmvrCus.clear();
  PelUnitBuf reco = cs.getRecoBuf();
  auto isIntraAt = [&]( const CodingUnit& cur, const Position& p ) { const CodingUnit* n = cs.getCURestricted( p, cur, CHANNEL_TYPE_LUMA ); return n && CU::isIntra( *n ); };
  // (the walk is split into bands of CTUs, one per thread: every band writes records with band-local indices, the merge below makes them global)
  const int nParts = std::max( 1, std::min( threads, numCtu ) );
  if( (int) E.walk.size() < nParts ) E.walk.resize( nParts );
  auto walkCtus = [&]( int a0, int a1, Extracted::Walk& o )
  {
  o.cu.clear(); o.tu.clear(); o.coef.clear(); o.dmvrCus.clear(); o.numDmvr = 0; o.first.assign( a1 - a0, 0 );
  for( int a = a0; a < a1; a++ )
  {
    o.first[a - a0] = (uint32_t) o.cu.size();
    if( !cs.getCtuData( a ).firstCU ) continue;
    for( auto& cu : cs.traverseCUs( a ) )
    {
      vvr_cu c; memset( &c, 0, sizeof( c ) );
      const ChannelType cht = cu.chType();
      const bool treeL = CU::isSepTree( cu ) && isLuma( cht ) && chroma, treeC = CU::isSepTree( cu ) && isChroma( cht );
      const Area la = treeC ? Area( cu.Cb().x << 1, cu.Cb().y << 1, cu.Cb().width << 1, cu.Cb().height << 1 ) : Area( cu.Y() );
      c.x = (uint16_t) la.x; c.y = (uint16_t) la.y; c.w = (uint8_t) la.width; c.h = (uint8_t) la.height;
      c.tree = treeL ? VVR_TREE_LUMA : treeC ? VVR_TREE_CHROMA : VVR_TREE_JOINT;
      c.pred_mode = CU::isIntra( cu ) ? VVR_PRED_INTRA : CU::isIBC( cu ) ? VVR_PRED_IBC : VVR_PRED_INTER;
      c.flags = (uint16_t) ( ( cu.rootCbf() ? VVR_CU_ROOT_CBF : 0 ) | ( cu.skip() ? VVR_CU_SKIP : 0 ) | ( cu.mergeFlag() ? VVR_CU_MERGE : 0 ) | ( isAffine( cu ) ? VVR_CU_AFFINE : 0 )
                           | ( isAffine( cu ) && cu.affineType() == AFFINEMODEL_6PARAM ? VVR_CU_AFFINE_6P : 0 ) | ( cu.ciipFlag() ? VVR_CU_CIIP : 0 ) | ( cu.geoFlag() ? VVR_CU_GEO : 0 )
                           | ( cu.mergeFlag() && cu.mergeType() == MRG_TYPE_SUBPU_ATMVP ? VVR_CU_SBTMVP : 0 ) | ( cu.mipFlag() ? VVR_CU_MIP : 0 ) | ( cu.mipTransposedFlag() ? VVR_CU_MIP_TRANSP : 0 )
                           | ( cu.smvdMode() ? VVR_CU_SMVD : 0 ) | ( cu.mmvdFlag() ? VVR_CU_MMVD : 0 ) );
      c.qp = (int8_t) cu.qp;
      c.bcw_idx = 2; c.ref_idx[0] = c.ref_idx[1] = -1;
      if( c.pred_mode == VVR_PRED_INTRA )
      {
        // final modes (PU::getFinalIntraMode, UnitTools.cpp:587); MIP keeps its matrix index; the mode LFNST derives its set from for LM chroma
        c.intra_dir[0] = treeC ? (uint8_t) PU::getCoLocatedIntraLumaMode( cu ) : (uint8_t) cu.intraDir[0];
        c.intra_dir[1] = chroma && !treeL ? (uint8_t) PU::getFinalIntraMode( cu, CHANNEL_TYPE_CHROMA ) : (uint8_t) cu.intraDir[1];
        c.lfnst_intra_mode = (uint8_t) ( treeC || !cu.mipFlag() ? PU::getCoLocatedIntraLumaMode( cu ) : PLANAR_IDX );
        if( !treeC && cu.mipFlag() ) c.lfnst_intra_mode = PLANAR_IDX;
        c.multi_ref_idx = (uint8_t) cu.multiRefIdx(); c.isp_mode = (uint8_t) cu.ispMode();
        c.bdpcm[0] = (uint8_t) cu.bdpcmMode(); c.bdpcm[1] = (uint8_t) cu.bdpcmModeChroma();
        c.lfnst_idx = (uint8_t) cu.lfnstIdx();
      }
      else if( c.pred_mode == VVR_PRED_IBC )
      {
        c.intra_dir[0] = (uint8_t) cu.intraDir[0];
        c.inter_dir = 1;
        c.mv[0][0][0] = cu.mv[0][0].getHor(); c.mv[0][0][1] = cu.mv[0][0].getVer();
      }
      else
      {
        const int se = st.entryOf( *cu.slice );
        c.inter_dir = (uint8_t) cu.interDir(); c.ref_idx[0] = uniIdx( se, 0, cu.refIdx[0] ); c.ref_idx[1] = uniIdx( se, 1, cu.refIdx[1] );
        for( int k = 0; k < 5; k++ ) if( g_BcwInternFwd[k] == cu.BcwIdx() ) c.bcw_idx = (uint8_t) k;      // description: index into the weight table
        // A CIIP or GPM CU keeps the BCW index of the merge candidate it took its motion from, but its two predictions are averaged with equal weights
        // (xWeightedAverage, InterPrediction.cpp:1356: `BcwIdx != BCW_DEFAULT && !ciipFlag`; GPM blends by position): the description says what is APPLIED.
        // Found with the first parser-fed stream that had BCW (tools/mini_vvenc.py, round 4).
        if( cu.ciipFlag() || cu.geoFlag() ) c.bcw_idx = 2;
        c.imv = (uint8_t) cu.imv(); c.sbt_info = (uint8_t) cu.sbtInfo(); c.lfnst_idx = 0;
        // control-point MVs only for affine CUs (a GPM CU keeps its two MVs in mv[0][1] / mv[1][1], InterPrediction.cpp:1478,1489: they go to geo_mv)
        for( int l = 0; l < 2; l++ ) for( int k = 0; k < ( isAffine( cu ) ? 3 : 1 ); k++ ) { c.mv[l][k][0] = cu.mv[l][k].getHor(); c.mv[l][k][1] = cu.mv[l][k].getVer(); }
        if( cu.geoFlag() )
        {
          c.geo_split_dir = cu.geoSplitDir;
          const uint8_t g[2] = { cu.interDirrefIdxGeo0(), cu.interDirrefIdxGeo1() };      // (interDir << 4) | refIdx of the partition's list
          for( int k = 0; k < 2; k++ ) c.geo_dir_ref[k] = (uint8_t) ( ( g[k] & 0xf0 ) | ( uniIdx( se, ( g[k] >> 4 ) == 1 ? 0 : 1, g[k] & 15 ) & 15 ) );
          c.geo_mv[0][0] = cu.mv[0][1].getHor(); c.geo_mv[0][1] = cu.mv[0][1].getVer(); c.geo_mv[1][0] = cu.mv[1][1].getHor(); c.geo_mv[1][1] = cu.mv[1][1].getVer();
        }
        if( cu.ciipFlag() )
        {
          // IntraPrediction::predBlendIntraCiip (IntraPrediction.cpp:917-927): the CU left of the bottom-left sample and the CU above the top-right sample
          const Position posBL = cu.Y().bottomLeft(), posTR = cu.Y().topRight();
          c.ciip_neigh_intra = (uint8_t) ( ( isIntraAt( cu, posBL.offset( -1, 0 ) ) ? 1 : 0 ) | ( isIntraAt( cu, posTR.offset( 0, -1 ) ) ? 2 : 0 ) );
        }
        c.mc_mode = resolveMcMode( cu );
        if( c.mc_mode == VVR_MC_DMVR || c.mc_mode == VVR_MC_DMVR_BDOF ) { c.dmvr_off = o.numDmvr; o.dmvrCus.emplace_back( &cu, o.numDmvr ); o.numDmvr += ( ( la.width + 15 ) / 16 ) * ( ( la.height + 15 ) / 16 ); }
      }
      c.first_tu = (uint32_t) o.tu.size();
      const uint32_t cuIdx = (uint32_t) o.cu.size();
      for( auto& tu : TUTraverser( &cu.firstTU, cu.lastTU->next ) )
      {
        vvr_tu t; memset( &t, 0, sizeof( t ) );
        // (a CU of a separate tree carries the blocks of its tree only - but the EMPTY transform unit the parser gives a skipped or residual-free IBC CU of the luma
        // tree spans all components, CodingStructure::addEmptyTUs: found with the first parser-fed IBC streams in dual-tree pictures, round 5)
        const bool hasL = tu.blocks[0].valid() && !treeC, hasC = chroma && !treeL && tu.blocks.size() > 1 && tu.blocks[1].valid();
        const Area ta = hasL ? Area( tu.blocks[0] ) : Area( tu.blocks[1].x << 1, tu.blocks[1].y << 1, tu.blocks[1].width << 1, tu.blocks[1].height << 1 );
        t.x = (uint16_t) ta.x; t.y = (uint16_t) ta.y; t.w = (uint8_t) ta.width; t.h = (uint8_t) ta.height;
        t.comp_mask = (uint8_t) ( ( hasL ? 1 : 0 ) | ( hasC ? 6 : 0 ) );
        t.cbf = tu.cbf; t.joint_cbcr = tu.jointCbCr; t.cu = cuIdx;
        t.qp[0] = (int8_t) ( cu.qp + qpBd ); t.qp[1] = (int8_t) tu.chromaQp[0]; t.qp[2] = (int8_t) tu.chromaQp[1];
        for( int k = 0; k < nComp; k++ )
        {
          t.mts_idx[k] = (uint8_t) tu.mtsIdx( ComponentID( k ) ); t.max_scan_x[k] = (uint8_t) tu.maxScanPosX[k]; t.max_scan_y[k] = (uint8_t) tu.maxScanPosY[k];
          if( !( t.comp_mask & ( 1 << k ) ) ) continue;
          // joint Cb-Cr: one coded block carries the levels of both components (Cb for modes 2 / 3, Cr for mode 1, TrQuant.cpp:320)
          const bool coded = ( ( tu.cbf >> k ) & 1 ) && !( k && tu.jointCbCr && k != ( ( tu.jointCbCr >> 1 ) ? 1 : 2 ) );
          if( coded && t.mts_idx[k] != VVR_MTS_SKIP )
          {
            int trHor = 0, trVer = 0;
            trQuant.getTrTypes( tu, ComponentID( k ), trHor, trVer );          // (TrQuant.cpp:330-407) DCT2 0, DCT8 1, DST7 2
            t.tr_type[k] = (uint8_t) ( ( trVer << 2 ) | trHor );
          }
          if( !coded ) continue;
          // levels: the parser leaves them in the reconstruction buffer at the block position (CABACReader.cpp:2457-2478); only the
          // corner up to the last significant position is meaningful (the whole block for BDPCM)
          const CompArea& blk = tu.blocks[k];
          const bool full = ( k == 0 ? cu.bdpcmMode() : cu.bdpcmModeChroma() ) != 0;
          const int cw = full ? (int) blk.width : t.max_scan_x[k] + 1, ch = full ? (int) blk.height : t.max_scan_y[k] + 1;
          t.coef_off[k] = (uint32_t) o.coef.size();
          const PelBuf src = reco.bufs[k].subBuf( blk.pos(), blk.size() );
          {
            // (rows of the corner at once: Pel is a 16-bit sample)
            static_assert( sizeof( Pel ) == sizeof( int16_t ), "levels are copied as 16-bit values" );
            const size_t at = o.coef.size();
            o.coef.resize( at + (size_t) cw * ch );
            for( int y = 0; y < ch; y++ ) memcpy( &o.coef[at + (size_t) y * cw], &src.at( 0, y ), sizeof( int16_t ) * cw );
          }
        }
        o.tu.push_back( t );
      }
      c.num_tu = (uint32_t) o.tu.size() - c.first_tu;
      o.cu.push_back( c );
    }
  }
  };
  parallelFor( nParts, nParts, [&]( int r ) { walkCtus( (int) ( (int64_t) numCtu * r / nParts ), (int) ( (int64_t) numCtu * ( r + 1 ) / nParts ), E.walk[r] ); } );
  {
    size_t nCu = 0, nTu = 0, nCoef = 0;
    for( int r = 0; r < nParts; r++ ) { nCu += E.walk[r].cu.size(); nTu += E.walk[r].tu.size(); nCoef += E.walk[r].coef.size(); }
    E.cu.resize( nCu ); E.tu.resize( nTu ); E.coef.resize( std::max<size_t>( 1, nCoef ) ); if( !nCoef ) E.coef[0] = 0;
    // (where every band's records go follows from the bands' sizes; the bands are then copied side by side)
    std::vector<uint32_t> cuB( nParts + 1, 0 ), tuB( nParts + 1, 0 ), coefB( nParts + 1, 0 ), dmvrB( nParts + 1, 0 );
    for( int r = 0; r < nParts; r++ )
    {
      cuB[r + 1] = cuB[r] + (uint32_t) E.walk[r].cu.size(); tuB[r + 1] = tuB[r] + (uint32_t) E.walk[r].tu.size();
      coefB[r + 1] = coefB[r] + (uint32_t) E.walk[r].coef.size(); dmvrB[r + 1] = dmvrB[r] + E.walk[r].numDmvr;
    }
    parallelFor( nParts, nParts, [&]( int r )
    {
      Extracted::Walk& o = E.walk[r];
      const uint32_t cuBase = cuB[r], tuBase = tuB[r], coefBase = coefB[r], dmvrBase = dmvrB[r];
      const int a0 = (int) ( (int64_t) numCtu * r / nParts );
      for( size_t k = 0; k < o.first.size(); k++ ) E.ctuFirstCu[a0 + k] = o.first[k] + cuBase;
      for( size_t k = 0; k < o.cu.size(); k++ )
      {
        vvr_cu c = o.cu[k];
        c.first_tu += tuBase;
        if( c.pred_mode == VVR_PRED_INTER && ( c.mc_mode == VVR_MC_DMVR || c.mc_mode == VVR_MC_DMVR_BDOF ) ) c.dmvr_off += dmvrBase;
        E.cu[cuBase + k] = c;
      }
      for( size_t k = 0; k < o.tu.size(); k++ )
      {
        vvr_tu t = o.tu[k];
        t.cu += cuBase;
        // (only the components with coded levels carry an offset into the level stream: the condition of the walk above)
        for( int q = 0; q < nComp; q++ ) if( ( t.comp_mask & ( 1 << q ) ) && ( ( t.cbf >> q ) & 1 ) && !( q && t.joint_cbcr && q != ( ( t.joint_cbcr >> 1 ) ? 1 : 2 ) ) ) t.coef_off[q] += coefBase;
        E.tu[tuBase + k] = t;
      }
      if( !o.coef.empty() ) memcpy( &E.coef[coefBase], o.coef.data(), sizeof( int16_t ) * o.coef.size() );
    } );
    for( int r = 0; r < nParts; r++ ) for( auto& d : E.walk[r].dmvrCus ) E.dmvrCus.emplace_back( d.first, d.second + dmvrB[r] );
    const uint32_t dmvrBase = dmvrB[nParts];
    E.numDmvr = dmvrBase;
  }
  E.ctuFirstCu[numCtu] = (uint32_t) E.cu.size();

  const auto tX1 = std::chrono::steady_clock::now();
  // ---- per-4x4 tables: motion (after MIDER), edge parameters (after LF_INIT)
  if( !lfpOnDevice ) { E.lfp[0].resize( (size_t) w4 * h4 ); E.lfp[1].resize( (size_t) w4 * h4 ); }
  else h.tool_flags |= VVR_TOOL_LFP_ON_DEVICE;
  if( !subBlockMotionOnly ) E.motion.resize( (size_t) w4 * h4 );
  else if( E.motionSparseCells < (size_t) w4 * h4 ) { E.motionSparse.reset( new vvr_motion[(size_t) w4 * h4] ); E.motionSparseCells = (size_t) w4 * h4; }
  vvr_motion* const motionOut = subBlockMotionOnly ? E.motionSparse.get() : E.motion.data();
  // motion of one 4x4 cell.  The back-end reads the motion field only under affine and SbTMVP CUs (their sub-block MVs) - and everywhere when it keeps the
  // collocated motion (VVR_TOOL_COL_MOTION); subBlockMotionOnly: only those cells are written (20 of the 36 bytes per cell the tables take otherwise)
  auto motionCell = [&]( int x, int y )
  {
    const int a = ( y / ctu4 ) * pcv.widthInCtus + ( x / ctu4 ), in = ( y % ctu4 ) * ctu4 + ( x % ctu4 );
    const CtuData& cd = cs.getCtuData( a );
    const int se = multi && cd.slice ? st.entryOf( *cd.slice ) : 0;
    vvr_motion& m = motionOut[(size_t) y * w4 + x]; memset( &m, 0, sizeof( m ) );
    const MotionInfo& mi = cd.motion[in];
    for( int l = 0; l < 2; l++ )
    {
      m.ref_idx[l] = isMotionValid( mi.miRefIdx[l], MI_NOT_VALID ) ? uniIdx( se, l, mi.miRefIdx[l] ) : (int8_t) -1;
      m.mv[l][0] = mi.mv[l].getHor(); m.mv[l][1] = mi.mv[l].getVer();
    }
  };
  // (row y of the picture = one run of cells per CTU it crosses: the CTU, its slice and the row inside the CTU's tables are looked up once per run)
  parallelFor( h4, threads, [&]( int y )
  {
    const int cy = y / ctu4, iy = y % ctu4;
    for( int cx = 0; cx < (int) pcv.widthInCtus; cx++ )
    {
      const CtuData& cd = cs.getCtuData( cy * pcv.widthInCtus + cx );
      const int x0 = cx * ctu4, n = std::min( ctu4, w4 - x0 );
      // a slice with deblocking switched off in a picture that deblocks: LF_INIT leaves the edge parameters of its CTUs untouched and the
      // filter skips them (LoopFilter.cpp:366,423) - here they carry no edge
      const bool noEdges = !allDbkOff && cd.slice && cd.slice->getDeblockingFilterDisable();
      for( int d = 0; d < 2 && !lfpOnDevice; d++ )
      {
        vvr_lfp* o = &E.lfp[d][(size_t) y * w4 + x0];
        memset( o, 0, sizeof( vvr_lfp ) * n );
        if( noEdges ) continue;
        const LoopFilterParam* s = &cd.lfParam[d][iy * ctu4];
        for( int k = 0; k < n; k++ ) { o[k].qp[0] = s[k].qp[0]; o[k].qp[1] = s[k].qp[1]; o[k].qp[2] = s[k].qp[2]; o[k].bs = s[k].bs; o[k].side_max_filt_length = s[k].sideMaxFiltLength; o[k].flags = s[k].flags; }
      }
      if( !subBlockMotionOnly ) for( int k = 0; k < n; k++ ) motionCell( x0 + k, y );
    }
  } );

  if( subBlockMotionOnly )
    parallelFor( threads, threads, [&]( int r )
    {
      for( size_t k = E.cu.size() * r / threads; k < E.cu.size() * ( r + 1 ) / threads; k++ )
      {
        const vvr_cu& c = E.cu[k];
        // (with the edge parameters left to the back-end it reads the motion of GPM CUs as well: which of its two predictions a cell kept)
        if( c.pred_mode == VVR_PRED_INTER && ( c.flags & ( VVR_CU_AFFINE | VVR_CU_SBTMVP | ( lfpOnDevice ? VVR_CU_GEO : 0 ) ) ) )
          for( int y = c.y >> 2; y < ( c.y + c.h + 3 ) >> 2; y++ ) for( int x = c.x >> 2; x < ( c.x + c.w + 3 ) >> 2; x++ ) motionCell( x, y );
      }
    } );
  const auto tX2 = std::chrono::steady_clock::now();
  if( getenv( "VVR_EXTRACT_TIMES" ) ) fprintf( stderr, "[extract] CU/TU/levels %.2f ms, 4x4 tables %.2f ms\n", std::chrono::duration<double, std::milli>( tX1 - tX0 ).count(), std::chrono::duration<double, std::milli>( tX2 - tX1 ).count() );
  // ---- per-CTU loop filter controls: SAO with merges resolved and offsets scaled (SampleAdaptiveOffset::reconstructBlkSAOParam), ALF
  E.sao.assign( numCtu, vvr_sao_ctu() ); E.alf.assign( numCtu, vvr_alf_ctu() );
  for( int a = 0; a < numCtu; a++ )
  {
    vvr_sao_ctu& s = E.sao[a]; memset( &s, 0, sizeof( s ) );
    for( int k = 0; k < nComp; k++ )
    {
      int src = a;
      const SAOOffset* o = &cs.getCtuData( src ).saoParam[k];
      while( o->modeIdc == SAO_MODE_MERGE ) { src = o->typeIdc == SAO_MERGE_LEFT ? src - 1 : src - (int) pcv.widthInCtus; o = &cs.getCtuData( src ).saoParam[k]; }
      if( o->modeIdc == SAO_MODE_OFF ) continue;
      const int sc = h.log2_sao_offset_scale[k ? 1 : 0];
      s.mode[k] = 1; s.type[k] = (uint8_t) o->typeIdc;
      if( o->typeIdc == SAO_TYPE_BO ) { s.band_pos[k] = (uint8_t) o->typeAuxInfo; for( int i = 0; i < 4; i++ ) s.offset[k][i] = (int8_t) ( o->offset[( o->typeAuxInfo + i ) % NUM_SAO_BO_CLASSES] << sc ); }
      else
      {
        s.offset[k][0] = (int8_t) ( o->offset[SAO_CLASS_EO_FULL_VALLEY] << sc ); s.offset[k][1] = (int8_t) ( o->offset[SAO_CLASS_EO_HALF_VALLEY] << sc );
        s.offset[k][2] = (int8_t) ( o->offset[SAO_CLASS_EO_HALF_PEAK] << sc );   s.offset[k][3] = (int8_t) ( o->offset[SAO_CLASS_EO_FULL_PEAK] << sc );
      }
    }
    vvr_alf_ctu& f = E.alf[a]; memset( &f, 0, sizeof( f ) );
    const CtuAlfData& ad = cs.getCtuData( a ).alfParam;
    for( int k = 0; k < 3; k++ ) f.enable[k] = ad.alfCtuEnableFlag[k];
    const Slice* cs_ = cs.getCtuData( a ).slice;      // (the reference filters with the control value only where the CTU's slice has CC-ALF on, AdaptiveLoopFilter.cpp:623-625)
    for( int k = 0; k < 2; k++ ) { f.alt[k] = ad.alfCtuAlternative[k]; f.cc_idc[k] = !cs_ || cs_->getCcAlfEnabledFlag( k + 1 ) ? ad.ccAlfFilterControl[k] : 0; }
    f.luma_filter_idx = ad.alfCtbFilterIndex;
  }

  // ---- final ALF filters of the APSs the slices refer to (after AdaptiveLoopFilter::reconstructCoeffAPSs): one table per distinct choice
  E.slices.assign( st.first.size(), vvr_slice_header() ); memset( (void*) E.slices.data(), 0, E.slices.size() * sizeof( vvr_slice_header ) );
  E.alfSets.clear();
  auto alfOf = [&]( const Slice& sl, vvr_alf_params& A )
  {
    memset( &A, 0, sizeof( A ) );
    const APS* const* apss = sl.getAlfAPSs();
    // (an APS id the slice header did not carry - chroma ALF or CC-ALF of a component switched off - is -1: never an index.  Found by tools/fuzz_dropin_on_the_oracle.py,
    // round 4: a single-slice picture with CC-ALF on for one chroma component only read apss[-1])
    auto apsAt = [&]( int id ) -> const APS* { return id >= 0 && id < ALF_CTB_MAX_NUM_APS ? apss[id] : nullptr; };
    A.num_luma_aps = (uint8_t) sl.getNumAlfAps();
    for( int i = 0; i < sl.getNumAlfAps() && i < VVR_MAX_ALF_APS; i++ )
    {
      const APS* la = apsAt( sl.getAlfApsIdsLuma()[i] );
      if( !la ) continue;
      const AlfSliceParam& p = la->getAlfAPSParam();
      for( int cl = 0; cl < VVR_ALF_CLASSES; cl++ ) for( int k = 0; k < MAX_NUM_ALF_LUMA_COEFF - 1; k++ )      // (the centre tap is implied)
      { A.luma_coeff[i][cl][k] = p.lumaCoeffFinal[cl * MAX_NUM_ALF_LUMA_COEFF + k]; A.luma_clip[i][cl][k] = p.lumaClippFinal[cl * MAX_NUM_ALF_LUMA_COEFF + k]; }
    }
    if( chroma && ( sl.getAlfEnabledFlag( COMPONENT_Cb ) || sl.getAlfEnabledFlag( COMPONENT_Cr ) || !multi ) && apsAt( sl.getAlfApsIdChroma() ) )
    {
      const AlfSliceParam& p = apsAt( sl.getAlfApsIdChroma() )->getAlfAPSParam();
      for( int alt = 0; alt < VVR_ALF_MAX_CHR_ALT && alt < p.numAlternativesChroma; alt++ ) for( int k = 0; k < MAX_NUM_ALF_CHROMA_COEFF - 1; k++ )
      { A.chroma_coeff[alt][k] = p.chromaCoeff[alt * MAX_NUM_ALF_CHROMA_COEFF + k]; A.chroma_clip[alt][k] = p.chrmClippFinal[alt * MAX_NUM_ALF_CHROMA_COEFF + k]; }
    }
    if( chroma && ( h.tool_flags & VVR_TOOL_CCALF ) )
      for( int k = 0; k < 2; k++ )
      {
        if( !sl.getCcAlfEnabledFlag( k + 1 ) ) continue;
        const APS* aps = apsAt( k == 0 ? sl.getCcAlfCbApsId() : sl.getCcAlfCrApsId() );
        if( !aps ) continue;
        const CcAlfFilterParam& cc = aps->getCcAlfAPSParam();
        for( int fI = 0; fI < VVR_CCALF_FILTERS; fI++ ) for( int j = 0; j < VVR_CCALF_TAPS + 1 && j < MAX_NUM_CC_ALF_CHROMA_COEFF; j++ ) A.ccalf_coeff[k][fI][j] = cc.ccAlfCoeff[k][fI][j];
      }
  };
  auto setOf = [&]( auto& sets, const auto& t ) -> uint8_t
  {
    for( size_t k = 0; k < sets.size(); k++ ) if( !memcmp( &sets[k], &t, sizeof( t ) ) ) return (uint8_t) k;
    sets.push_back( t ); return (uint8_t) ( sets.size() - 1 );
  };
  if( h.tool_flags & VVR_TOOL_ALF )
    for( size_t k = 0; k < st.first.size(); k++ )
    {
      const Slice& sl = *st.first[k];
      if( multi && !sl.getAlfEnabledFlag( COMPONENT_Y ) ) continue;        // (sh_alf_enabled_flag off: no APS ids, no CTU of the slice is filtered)
      vvr_alf_params A; alfOf( sl, A );
      E.slices[k].alf_set = setOf( E.alfSets, A );
    }
  if( E.alfSets.empty() ) { E.alfSets.emplace_back(); memset( &E.alfSets[0], 0, sizeof( vvr_alf_params ) ); }

  // ---- LMCS: the tables Reshape::constructReshaper built (Reshape.cpp:318-374); the forward map is tabulated with rspFwdCore's formula
  memset( &E.lmcs, 0, sizeof( E.lmcs ) );
  if( ( h.tool_flags & VVR_TOOL_LMCS ) && reshaper ) LmcsTables::read( *reshaper, bd, E.lmcs );

  // ---- explicit weighted prediction (the slices' tables re-indexed to the union of the reference lists), scaling lists
  E.wpSets.clear();
  if( h.tool_flags & VVR_TOOL_WP )
    for( size_t k = 0; k < st.first.size(); k++ )
    {
      const Slice& sl = *st.first[k];
      if( !( toolFlags( cs, sl, pic ) & VVR_TOOL_WP ) ) continue;
      vvr_wp_params T; memset( &T, 0, sizeof( T ) );
      for( int l = 0; l < 2; l++ ) for( int i = 0; i < sl.getNumRefIdx( RefPicList( l ) ) && i < MAX_NUM_REF; i++ )
      {
        const WPScalingParam* wp = nullptr;
        sl.getWpScaling( RefPicList( l ), i, wp );
        for( int c = 0; c < 3; c++ )
        {
          T.log2_denom[c ? 1 : 0] = (uint8_t) wp[c].uiLog2WeightDenom;
          vvr_wp_entry& e = T.e[l][ru.map[k][l][i]][c]; e.present = wp[c].bPresentFlag; e.weight = (int16_t) wp[c].iWeight; e.offset = (int16_t) wp[c].iOffset;
        }
      }
      E.slices[k].wp_set = setOf( E.wpSets, T );
    }
  if( E.wpSets.empty() ) { E.wpSets.emplace_back(); memset( &E.wpSets[0], 0, sizeof( vvr_wp_params ) ); }
  memset( &E.scaling, 0, sizeof( E.scaling ) );
  if( ( h.tool_flags & VVR_TOOL_SCALING_LIST ) && cs.picHeader->getScalingListAPS() )
  {
    const ScalingList& sl = cs.picHeader->getScalingListAPS()->getScalingList();
    for( int id = 0; id < 28; id++ )
    {
      const int n = ScalingList::matrixSize( id );
      const int* src = sl.getScalingListAddress( id );
      for( int k = 0; k < n * n; k++ ) E.scaling.coef[id][k] = (uint8_t) src[k];
      E.scaling.dc[id] = (uint8_t) sl.getScalingListDC( id );
    }
  }

  // ---- the picture
  E.pic.num_cu = (uint32_t) E.cu.size(); E.pic.num_tu = (uint32_t) E.tu.size();
  E.pic.cu = E.cu.data(); E.pic.tu = E.tu.data(); E.pic.ctu_first_cu = E.ctuFirstCu.data(); E.pic.coef = E.coef.data(); E.pic.num_coef = E.coef.size();
  E.pic.motion = subBlockMotionOnly ? E.motionSparse.get() : E.motion.data(); E.pic.lfp[0] = lfpOnDevice ? nullptr : E.lfp[0].data(); E.pic.lfp[1] = lfpOnDevice ? nullptr : E.lfp[1].data();
  E.pic.sao = ( h.tool_flags & ( VVR_TOOL_SAO_LUMA | VVR_TOOL_SAO_CHROMA ) ) ? E.sao.data() : nullptr;
  E.pic.alf = ( h.tool_flags & VVR_TOOL_ALF ) ? E.alf.data() : nullptr;
  E.pic.alf_params = ( h.tool_flags & VVR_TOOL_ALF ) ? E.alfSets.data() : nullptr; E.pic.num_alf_sets = (uint32_t) E.alfSets.size();
  E.pic.lmcs = ( h.tool_flags & VVR_TOOL_LMCS ) ? &E.lmcs : nullptr;
  E.pic.wp = ( h.tool_flags & VVR_TOOL_WP ) ? E.wpSets.data() : nullptr; E.pic.num_wp_sets = (uint32_t) E.wpSets.size();
  E.pic.scaling = ( h.tool_flags & VVR_TOOL_SCALING_LIST ) ? &E.scaling : nullptr;
  // ---- slices and tiles: index of every CTU (picture raster order)
  E.ctuSlice.clear(); E.ctuTile.clear();
  if( multi || pps.getNumTiles() > 1 )
  {
    E.ctuSlice.resize( numCtu ); E.ctuTile.resize( numCtu );
    for( int a = 0; a < numCtu; a++ )
    {
      const CodingUnit* first = cs.getCtuData( a ).cuPtr[0][0];
      E.ctuSlice[a] = first ? (uint16_t) st.entryOf( *first->slice ) : 0;
      E.ctuTile[a] = first ? (uint16_t) first->tileIdx : 0;
    }
    if( multi ) E.pic.ctu_slice = E.ctuSlice.data();
    if( pps.getNumTiles() > 1 ) E.pic.ctu_tile = E.ctuTile.data();
  }
  // ---- slices with headers of their own
  E.pic.slices = nullptr; E.pic.num_slices = 0;
  if( multi )
  {
    for( size_t k = 0; k < st.first.size(); k++ )
    {
      const Slice& sl = *st.first[k];
      vvr_slice_header& o = E.slices[k];
      o.tool_flags = toolFlags( cs, sl, pic ) & ( VVR_SLICE_TOOL_MASK | VVR_TOOL_DEBLOCK_OFF );      // (deblocking switched off by a slice of a picture that deblocks: read with VVR_TOOL_LFP_ON_DEVICE)
      if( !( o.tool_flags & VVR_TOOL_LMCS ) ) o.tool_flags &= ~(uint32_t) VVR_TOOL_LMCS_CSCALE;       // (the picture's flag and sh_lmcs_used_flag)
      sliceDbk( sl, o.deblock_beta_offset_div2, o.deblock_tc_offset_div2 );
      o.slice_type = (uint8_t) sl.getSliceType();
    }
    E.pic.slices = E.slices.data(); E.pic.num_slices = (uint32_t) E.slices.size();
  }
  // ---- sub-pictures (PPS::initSubPic has the rectangles; the flags come from the SPS)
  E.subpics.clear(); E.pic.subpics = nullptr; E.pic.num_subpics = 0;
  if( pps.getNumSubPics() > 1 )
  {
    for( int k = 0; k < pps.getNumSubPics(); k++ )
    {
      const SubPic& sp = pps.getSubPic( k );
      vvr_subpic o; memset( &o, 0, sizeof( o ) );
      o.x0 = (uint16_t) sp.getSubPicLeft(); o.y0 = (uint16_t) sp.getSubPicTop(); o.x1 = (uint16_t) sp.getSubPicRight(); o.y1 = (uint16_t) sp.getSubPicBottom();
      o.treated_as_pic = sp.getTreatedAsPicFlag(); o.lf_across = sp.getloopFilterAcrossSubPicEnabledFlag();
      E.subpics.push_back( o );
    }
    E.pic.subpics = E.subpics.data(); E.pic.num_subpics = (uint32_t) E.subpics.size();
  }
  E.pic.resident = 0;
}

}   // namespace vvr_glue
