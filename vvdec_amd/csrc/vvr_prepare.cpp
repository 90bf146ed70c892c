// vvdec_amd/csrc/vvr_prepare.cpp — host glue of the reconstruction back-end: validation of a picture description and the device work lists
// built from it (what DecCu::TaskTrafoCtu / TaskInterCtu / TaskCriticalIntraKernel iterate over in the reference, DecCu.cpp:106-160).
//
// This stage runs once per picture on a host thread (several pictures are prepared concurrently by the context's worker threads,
// vvr_api.cpp), so it is written for throughput: all containers live in a per-thread PrepScratch that is reused from picture to picture
// (no allocation in the steady state), per-block producer lists sit in one flat pool, and the result is packed straight into the pinned
// staging memory of the upload ring.  No device call happens here.
#include "vvr_host.h"
#include "vvr_lf_init.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <condition_variable>

static inline int ilog2i( int v ) { int l = 0; while( ( 1 << l ) < v ) l++; return l; }

struct BBox { int y0 = 255, y1 = 0, c0 = 255, c1 = 0; };   // rows relative to (CTU top - 3), 8-sample chunks relative to (CTU left - 8), chunk index + 1
// a block of the intra stage on the host: its CTU, the part of the CTU tile it may read, the blocks it reads from (range of the flat pool;
// entries are ( component << 28 ) | block index)
struct ItemH { uint32_t ctu; BBox bb; uint32_t p0, pn; };
// Intra-stage work units: a unit is a set of blocks of one (component, CTU) that are connected through the reference samples they read from
// each other (a whole CTU in an intra picture, a few blocks around an isolated intra CU in a B picture); one workgroup processes one unit,
// its blocks in coding order.  Units depend on exactly those other units that produced a sample they read (inter samples are final before
// the stage starts).
// the units a unit waits for: a handful (the neighbouring CTUs' units, the luma unit of the CTU), kept in the record itself - a heap block per unit
// and picture was 1500 allocations per picture; the rare longer list moves to the heap
struct DepList
{
  enum { INLINE = 8 };
  uint32_t n = 0, cap = INLINE; uint32_t in[INLINE]; uint32_t* heap = nullptr;
  DepList() {}
  DepList( const DepList& o ) { *this = o; }
  DepList( DepList&& o ) noexcept { n = o.n; cap = o.cap; heap = o.heap; memcpy( in, o.in, sizeof( in ) ); o.heap = nullptr; o.n = 0; o.cap = INLINE; }
  DepList& operator=( const DepList& o ) { if( this != &o ) { n = 0; for( uint32_t v : o ) push_back( v ); } return *this; }
  DepList& operator=( DepList&& o ) noexcept { if( this != &o ) { delete[] heap; n = o.n; cap = o.cap; heap = o.heap; memcpy( in, o.in, sizeof( in ) ); o.heap = nullptr; o.n = 0; o.cap = INLINE; } return *this; }
  ~DepList() { delete[] heap; }
  uint32_t* data() { return heap ? heap : in; } const uint32_t* data() const { return heap ? heap : in; }
  uint32_t* begin() { return data(); } uint32_t* end() { return data() + n; } const uint32_t* begin() const { return data(); } const uint32_t* end() const { return data() + n; }
  size_t size() const { return n; } bool empty() const { return n == 0; }
  uint32_t& operator[]( size_t i ) { return data()[i]; } const uint32_t& operator[]( size_t i ) const { return data()[i]; }
  void push_back( uint32_t v ) { if( n == cap ) { uint32_t* h = new uint32_t[2 * cap]; memcpy( h, data(), sizeof( uint32_t ) * n ); delete[] heap; heap = h; cap *= 2; } data()[n++] = v; }
  void resize( size_t k ) { n = (uint32_t) std::min<size_t>( k, n ); }                                   // (only ever shortened)
  void assign( const uint32_t* a, const uint32_t* b ) { n = 0; for( ; a != b; a++ ) push_back( *a ); }
  void clear() { n = 0; }
};
struct UnitH { uint32_t comp, ctu, i0, i1, iA = 0; bool hasCs = false; BBox bb; DepList deps; bool waited = false; int rank = 0; };

struct Part { const void* src; size_t n, off; bool direct; };

struct PrepScratch
{
  // ---- the picture being prepared
  const vvr_picture* p = nullptr;
  vvr_pic_header h;
  int ncomp = 0, w4 = 0, h4 = 0, ctu = 0, ctusX = 0, ctusY = 0, numCtu = 0, vpduLog2 = 0, vpdusX = 0, vpdusY = 0;
  bool wpOn = false, cscale = false, lmcs = false;
  // ---- work lists
  std::vector<McItem> mcRpr;                // tiles of CUs that predict from a scaled reference picture (k_mc_rpr): plain, SbTMVP, GPM and affine ones
  std::vector<McItem> mc, mcBdof, mcDmvr, mcAff;      // tiles the host writes: SbTMVP sub-blocks (mc), affine tiles (mcAff); mcBdof / mcDmvr stay empty (k_expand_mc)
  std::vector<McCuRef> mcCus;              // CUs whose tiles the device writes, with where (list, first tile)
  uint32_t devTiles[3] = { 0, 0, 0 };      // tiles the device writes per list (plain, BDOF, DMVR)
  std::vector<uint16_t> ctuSubpicV;        // sub-picture of every CTU, built from the rectangles (layout)
  std::vector<vvr_motion> affMv;           // motion of the 4x4 sub-blocks of the affine tiles, 16 entries per tile (the only part of the motion field a kernel reads)
  bool lfpOnDevice = false;                // VVR_TOOL_LFP_ON_DEVICE on a picture that deblocks: the edge parameters are derived on the device (k_lf_init)
  std::vector<LfSbCell> lfSb;              // ... which reads the motion of the cells of SbTMVP and GPM CUs (affine: unless the device spans it) from this list
  uint32_t numDmvr = 0;
  std::vector<TbItem> tb[3];
  std::vector<IntraItem> intra[3], intraTmp[3], intraAll;
  std::vector<uint32_t> posAfterGrouping[3]; uint32_t groupFill[3] = { 0, 0, 0 };      // (groupUnits) block -> its place in the regrouped list
  std::vector<uint32_t> itemMap[3];                       // block of a component -> its first item in intraAll (large blocks become several items)
  std::vector<ItemH> itemH[3], itemHTmp;
  std::vector<uint32_t> prodPool[3];
  std::vector<uint32_t> ctuStartV;
  double bytes[K_NUM] = { 0 };
  double bytesBdof = 0, bytesIntraLuma = 0, bytesTb[3] = { 0, 0, 0 };
  // decode-order index of the transform block covering every 4x4 luma unit (both channel types): reference availability
  // = "inside the picture and reconstructed before me" (CodingStructure::getCURestricted, CodingStructure.cpp:464, and the
  // TU-index test of isAboveAvailable / isLeftAvailable, IntraPrediction.cpp:1343-1400)
  std::vector<int32_t> order;
  std::vector<uint8_t> fastCtu;            // per CTU: every CU is an intra CU.  Its blocks form one unit per component whatever they read from each other, and that unit reads
                                           // from the CTUs left, above-left, above and above-right only: no per-block producer analysis (formUnits)
  // per component and 4x4 luma cell: the block that reconstructs it in the intra stage, stamped with the number of the picture it was written for
  // ( epoch << 22 | block ): the maps are never cleared, an entry of another picture reads as "none"
  std::vector<uint32_t> itemAtE[3];
  // all-intra CTUs: which block of the CTU's (component) list reconstructs a cell of the CTU - 32 x 32 cells at most, rewritten per CTU: enough to tell
  // which of the blocks before it a block reads from (IntraItem `indep`), without the picture-wide producer analysis
  uint16_t fastCell[3][32 * 32]; uint32_t fastFirst[3] = { 0, 0, 0 };
  uint32_t epoch = 0;
  // (cells of one CTU lie together: a block and what it reads stay within a few KB)
  size_t cellIdx( int cx, int cy ) const { const int l = h.log2_ctu - 2, m = ( 1 << l ) - 1; return ( ( (size_t) ( cy >> l ) * ctusX + ( cx >> l ) ) << ( 2 * l ) ) | (size_t) ( ( cy & m ) << l ) | (size_t) ( cx & m ); }
  int32_t itemAtGet( int k, size_t cell ) const { const uint32_t v = itemAtE[k][cell]; return ( v >> 22 ) == epoch ? (int32_t) ( v & 0x3fffffu ) : -1; }
  std::vector<UnitH> units, unitsTmp;
  // LMCS chroma residual scaling: per VPDU the luma neighbourhood its factor is averaged over (Reshape::calculateChromaAdjVpduNei,
  // Reshape.cpp:192-274): left column / above row of the CU at the VPDU origin, where that neighbour precedes it in decoding order
  std::vector<uint32_t> csVpduV;
  std::vector<std::pair<uint32_t, uint32_t>> csProdRange;   // per VPDU: the luma blocks that produce that neighbourhood (range of csProdPool), looked up on first use
  std::vector<uint32_t> csProdPool;
  std::vector<IntraUnit> unitsDev;
  int intraWorkgroups = 0, intraWorkgroupsChroma = 0, numLumaUnits = 0;
  std::vector<IntraItem> resiAdd;          // residual-add blocks (inter chroma blocks with LMCS chroma residual scaling): k_resi_add, outside the stage's dependency graph
  size_t intraChunk = (size_t) 1 << 30;      // blocks per unit of a long intra cluster (formUnits); off: measured, no gain
  // union-find / grouping scratch
  std::vector<uint32_t> parent, newIdx, firstOf, perm, inv, unitCount, unitOfItem[3];
  std::vector<uint64_t> sortKey;
  std::vector<int32_t> unitOfRoot, target;
  std::vector<std::vector<uint32_t>> members, groups;
  // ---- layout of the H2D image
  std::vector<Part> parts;
  size_t total = 0, numDirect = 0;          // parts [0, numDirect) are copied from the caller's pinned arrays
  int iLfSb, iLfTu, iLfTuC, iLfMotion;
  int iCu, iTu, iCoef, iAffMv, iL0, iL1, iSao, iAlf, iAlfP, iSlices, iLmcs, iSl, iCtuSlice, iCtuTile, iSubpics, iCtuSubpic, iWp, iCsVpdu, iMc, iMcB, iMcD, iMcA, iTb[3], iIntra, iResi, iUnits, iMcCus, iMcDev[3], iRpr, iMcR;
  size_t stagedEndOff = 0;                  // end of the uploaded part of the image

  void begin( const vvr_picture* pic )
  {
    p = pic; h = pic->hdr;
#if defined( VVR_WATCHDOG ) || defined( VVR_DEV_ENV )
    if( const char* e = getenv( "VVR_INTRA_CHUNK" ) ) intraChunk = (size_t) atoi( e );      // developer build: sweep of the piece length
#endif
    ncomp = h.chroma_format ? 3 : 1;
    // slices with headers of their own: the switches a slice header carries hold per slice (vvr_slice_header.tool_flags); the working copy of the
    // picture header gets their UNION - "does any slice use the tool" is what decides which tables travel and which passes run, the kernels and
    // the per-block decisions below look the slice up
    if( p->slices && p->num_slices )
    {
      uint32_t any = 0;
      for( uint32_t i = 0; i < p->num_slices; i++ ) any |= p->slices[i].tool_flags & VVR_SLICE_TOOL_MASK;
      h.tool_flags = ( h.tool_flags & ~(uint32_t) VVR_SLICE_TOOL_MASK ) | any;
    }
    wpOn = ( h.tool_flags & VVR_TOOL_WP ) && h.slice_type != 2;
    w4 = ( h.width + 3 ) >> 2; h4 = ( h.height + 3 ) >> 2; ctu = 1 << h.log2_ctu;
    ctusX = ( h.width + ctu - 1 ) / ctu; ctusY = ( h.height + ctu - 1 ) / ctu; numCtu = ctusX * ctusY;
    lmcs = ( h.tool_flags & VVR_TOOL_LMCS ) != 0;
    cscale = lmcs && ( h.tool_flags & VVR_TOOL_LMCS_CSCALE ) && ncomp == 3;
    vpduLog2 = std::min<int>( 6, h.log2_ctu ); vpdusX = ( h.width + ( 1 << vpduLog2 ) - 1 ) >> vpduLog2; vpdusY = ( h.height + ( 1 << vpduLog2 ) - 1 ) >> vpduLog2;
    mc.clear(); mcBdof.clear(); mcDmvr.clear(); mcAff.clear(); mcRpr.clear(); affMv.clear(); numDmvr = 0;
    lfSb.clear(); lfpOnDevice = ( h.tool_flags & VVR_TOOL_LFP_ON_DEVICE ) && !( h.tool_flags & VVR_TOOL_DEBLOCK_OFF );
    mcCus.clear(); devTiles[0] = devTiles[1] = devTiles[2] = 0;
    for( int k = 0; k < 3; k++ ) { tb[k].clear(); intra[k].clear(); itemH[k].clear(); prodPool[k].clear(); }
    resiAdd.clear(); intraAll.clear(); units.clear(); unitsDev.clear(); csVpduV.clear(); intraFine = false;
    ctuStartV.assign( 3 * (size_t) ( numCtu + 1 ), 0 );
    for( double& b : bytes ) b = 0;
    bytesBdof = bytesIntraLuma = 0; bytesTb[0] = bytesTb[1] = bytesTb[2] = 0;
  }

  // `order` holds the cells of ONE CTU (both channel types): it is only ever asked about the CTU being analysed
  size_t orderIdx( int chn, int lx, int ly ) const { const int m = ( 1 << ( h.log2_ctu - 2 ) ) - 1; return ( ( (size_t) chn << ( h.log2_ctu - 2 ) | (size_t) ( ( ly >> 2 ) & m ) ) << ( h.log2_ctu - 2 ) ) | (size_t) ( ( lx >> 2 ) & m ); }
  int unitAvail( int chn, int x, int y, int32_t cur ) const
  {
    const int cs = chn ? 1 : 0, lx = x << cs, ly = y << cs;
    if( x < 0 || y < 0 || lx >= h.width || ly >= h.height ) return 0;
    // CTUs are decoded in raster order (inside a tile; CTUs of another tile are never available): everything in an earlier CTU is there, nothing
    // in a later one (whose cells are not even mapped yet).  Nothing is available across a slice or tile boundary (CodingStructure::
    // getCURestricted, CodingStructure.cpp:464).
    const uint32_t c = (uint32_t) ( ( ly >> h.log2_ctu ) * ctusX + ( lx >> h.log2_ctu ) );
    if( c != curCtuIdx ) return c < curCtuIdx && sameSliceAndTile( c, curCtuIdx );
    return order[orderIdx( chn, lx, ly )] < cur;
  }

  // the luma blocks of the intra stage that produce what the chroma scaling factor of VPDU `vp` is averaged over (they all precede the VPDU's
  // first CU in decoding order and are the same for every chroma block of the VPDU: looked up once)
  void lookUpCsProducers( size_t vp )
  {
    const uint32_t d = csVpduV[vp];
    const int xPos = d & 0x1fff, yPos = ( d >> 13 ) & 0x1fff, n = 1 << vpduLog2;
    const uint32_t start = (uint32_t) csProdPool.size();
    auto look = [&]( int lx, int ly )
    {
      if( ly < partTopY )
      {
        if( ( ly >> 2 ) != ( partTopY >> 2 ) - 1 ) { partBad = true; return; }
        csProdPool.push_back( 0x80000000u | (uint32_t) pending.size() ); pending.push_back( (uint32_t) ( lx >> 2 ) );      // (component 0)
        return;
      }
      const int32_t id = itemAtGet( 0, cellIdx( lx >> 2, ly >> 2 ) );
      if( id >= 0 && std::find( csProdPool.begin() + start, csProdPool.end(), (uint32_t) id ) == csProdPool.end() ) csProdPool.push_back( (uint32_t) id );
    };
    if( ( d >> 26 ) & 1 ) for( int k = 0; k < n; k += 4 ) look( xPos - 1, std::min( yPos + k, (int) h.height - 1 ) );
    if( ( d >> 27 ) & 1 ) for( int k = 0; k < n; k += 4 ) look( std::min( xPos + k, (int) h.width - 1 ), yPos - 1 );
    csProdRange[vp] = std::make_pair( start, (uint32_t) csProdPool.size() - start );
  }
  // the switches of the slice CTU `c` lies in
  uint32_t flagsOfCtu( uint32_t c ) const { return ( p->slices && p->ctu_slice ) ? ( h.tool_flags & ~(uint32_t) VVR_SLICE_TOOL_MASK ) | ( p->slices[p->ctu_slice[c]].tool_flags & VVR_SLICE_TOOL_MASK ) : h.tool_flags; }
  bool cscaleCtu( uint32_t c ) const { const uint32_t f = flagsOfCtu( c ); return cscale && ( f & VVR_TOOL_LMCS ) && ( f & VVR_TOOL_LMCS_CSCALE ); }
  bool sameSliceAndTile( uint32_t a, uint32_t b ) const { return ( !p->ctu_slice || p->ctu_slice[a] == p->ctu_slice[b] ) && ( !p->ctu_tile || p->ctu_tile[a] == p->ctu_tile[b] ); }
  uint32_t ctuAt( int lx, int ly ) const { return (uint32_t) ( ( ly >> h.log2_ctu ) * ctusX + ( lx >> h.log2_ctu ) ); }
  int beginMaps( const PrepScratch* like = nullptr );
  int mapCtu( uint32_t i0, uint32_t i1, uint32_t ctuIdx, std::string& err );
  bool anyIntra = false; uint32_t curCtuIdx = 0;
  uint32_t partCtu0 = 0, partCtu1 = 0xffffffffu;      // (buildInParts) the CTUs the CUs of the range being built have to lie in
  // (buildInParts, pictures with producer analysis) A band of CTU rows is analysed without the band above it: what a block reads of the last cell row of that
  // band (reference lines, CCLM templates, the luma of a chroma scaling factor) is entered as a PENDING producer - ( 1 << 31 | index into `pending` ), the entry
  // = component << 16 | cell column - and looked up when the bands are joined, in `edgeRow` (owner: per component and cell column the block, in picture-wide
  // numbering, that reconstructs the cell of the last cell row of the band joined before)
  int partTopY = 0; bool partBad = false;
  bool partsForAll = false;                // (set by whoever owns the scratch) pictures with inter CUs are built in bands too
  std::vector<uint32_t> pending;
  std::vector<int32_t> edgeRow[3];
  bool allIntraCus = false;                // every CU of the picture is an intra CU: every CTU takes the fast path, nobody ever looks a producer up
  // The intra stage of a picture with SCATTERED intra blocks (any picture that is not all intra CUs and has no IBC CU) runs one wavefront per block, ordered on the
  // device through per-cell words (k_intra_leaf, vvr_intra_leaf.inc): the host only lists the blocks in decoding order - no producer analysis, no block map, no
  // units.  `leafOn`: the owner of the scratch allows it (vvr_scratch_intra_leaf); `leaf`: this picture takes that path.
  bool leafOn = true, leaf = false;
  bool intraFine = false;                   // (tile path) every unit is a whole CTU of intra CUs: the launch may order the CTUs block by block
  std::vector<uint8_t> csNeeded;            // (leaf) per VPDU: a chroma block of the stage scales its residual with the VPDU's factor (an IT_MODE_CSFAC item computes it)
  // (leaf) The device orders the blocks by itself, but a wavefront that waits holds its slot: in decoding order the list front-loads the device with blocks deep in
  // a chain of neighbours.  `level` of a block: 1 + the highest level among the blocks that own the cells its wavefront polls (1: polls nothing of this stage); the
  // list sorted by level (stable: decoding order within a level) is still producers-first, and wavefronts find their cells cleared or about to be.  Needs the
  // blocks above in the map: pictures built in bands by several threads keep decoding order.  MEASURED (round 5, 4K B picture alone): 83 us against 84 us in
  // decoding order - the launch is as long as its longest chain of blocks (17 hops of 4..5 us each: the latency of one wavefront filling, predicting, storing and
  // publishing one block), not as the wavefronts that wait make it; off unless the owner of the scratch asks (VVR_LEAF_BY_LEVEL=1), kept with its test.
  bool leafSortOn = false, leafSort = false;
  std::vector<uint16_t> levMap;             // per component and cell: level of the intra-stage block that owns it in this picture, 0: none
  std::vector<uint16_t> lev[3], csLev;      // per block of intra[k]; per VPDU (0: not looked up yet)
  std::vector<uint16_t> levAllV;            // per item of the list
  uint16_t ispLev = 0;
  void beginLevels();
  uint16_t csLevelOf( size_t vp );
  inline void levLook( int k, int cx0, int cy0, int cx1, int cy1, uint32_t& d ) const
  {
    cx0 = std::max( cx0, 0 ); cy0 = std::max( cy0, 0 ); cx1 = std::min( cx1, w4 - 1 ); cy1 = std::min( cy1, h4 - 1 );
    const uint16_t* m = &levMap[(size_t) k * w4 * h4];
    for( int y = cy0; y <= cy1; y++ ) for( int x = cx0; x <= cx1; x++ ) d = std::max<uint32_t>( d, m[(size_t) y * w4 + x] );
  }
  inline void levOwn( int k, int cx0, int cy0, int cx1, int cy1, uint16_t v )
  {
    cx1 = std::min( cx1, w4 - 1 ); cy1 = std::min( cy1, h4 - 1 );
    uint16_t* m = &levMap[(size_t) k * w4 * h4];
    for( int y = cy0; y <= cy1; y++ ) std::fill( m + (size_t) y * w4 + cx0, m + (size_t) y * w4 + cx1 + 1, v );
  }
  int emitLeafItems( std::string& err );
  int buildWorkLists( std::string& err, uint32_t cu0 = 0, uint32_t cu1 = 0xffffffffu );
  int buildInParts( const vvr_config& cfg, HostHelpers& helpers, bool validate, std::string& err );
  int formUnits();
  int groupUnits();
  int emitUnitTable( std::string& err );
  void layout( PinnedRanges* pinned );
  void foldLongDepLists();
  void rankUnits();
};

PrepScratch* vvr_scratch_create() { return new PrepScratch(); }
void vvr_scratch_parts_for_all( PrepScratch* S, bool on ) { S->partsForAll = on; }
void vvr_scratch_intra_leaf( PrepScratch* S, bool on, bool byLevel ) { S->leafOn = on; S->leafSortOn = byLevel; }
// the transform units of a CU 64 wide and / or high in a sequence whose largest transform is 32: min( w, 32 ) x min( h, 32 ) each, in raster order over the CU
static bool tusAreTheSplitAt32( const vvr_picture* p, const vvr_cu& cu )
{
  const int tw = cu.w < 32 ? cu.w : 32, th = cu.h < 32 ? cu.h : 32, nx = cu.w / tw, ny = cu.h / th;
  if( (int) cu.num_tu != nx * ny || cu.num_tu < 2 ) return false;
  for( int k = 0; k < nx * ny; k++ )
  {
    const vvr_tu& tu = p->tu[cu.first_tu + k];
    if( tu.w != tw || tu.h != th || tu.x != cu.x + ( k % nx ) * tw || tu.y != cu.y + ( k / nx ) * th ) return false;
  }
  return true;
}

static std::atomic<int> g_bandPictures{ 0 };
int vvr_host_band_pictures() { return g_bandPictures.load(); }      // (tests) pictures with inter CUs that were built in bands so far

// Room for the lists of an ordinary picture of the context's size, allocated AND written once, by the thread that is going to use the scratch
// (first touch decides where the pages live).  A vector that has to grow in the middle of a picture is a new mapping, a copy and a page fault
// per 4 KB of it - under the process-wide mmap lock when 16 workers do it at the same time.  Measured (4K, one thread): the first / second /
// third picture on a fresh scratch take 28.6 / 14.4 / 10.6 ms, the first I picture after B pictures 16.6 instead of 13.5; with 16 workers a
// benchmark of 50 pictures never got past the second picture per worker, and its host stage took 6.0 ms instead of 4.5 (8 workers).
// The bounds are per 4x4 cell of the picture and generous for typical content (an all-intra picture at QP 22 has one block per 10 cells);
// a picture that needs more makes the vectors grow as before.
void vvr_scratch_warm( PrepScratch* S, const vvr_config& cfg )
{
  const size_t w4 = ( (size_t) cfg.max_width + 3 ) >> 2, h4 = ( (size_t) cfg.max_height + 3 ) >> 2, cells = w4 * h4;
  auto warm = [&]( auto& v, size_t n ) { if( v.capacity() < n ) { v.resize( n ); memset( (void*) v.data(), 0, n * sizeof( v[0] ) ); } v.clear(); };
  const size_t items = cells / 8 + 64;
  const int l2 = cfg.log2_ctu ? cfg.log2_ctu : 7;
  const size_t blocked = ( ( ( (size_t) cfg.max_width + ( 1u << l2 ) - 1 ) >> l2 ) * ( ( (size_t) cfg.max_height + ( 1u << l2 ) - 1 ) >> l2 ) ) << ( 2 * ( l2 - 2 ) );      // cells of whole CTUs (itemAtE)
  warm( S->mc, cells / 16 + 64 ); warm( S->mcBdof, cells / 32 + 64 ); warm( S->mcDmvr, cells / 32 + 64 ); warm( S->mcAff, cells / 32 + 64 );
  warm( S->affMv, cells / 4 + 64 );
  for( int k = 0; k < 3; k++ )
  {
    warm( S->tb[k], items ); warm( S->intra[k], items ); warm( S->intraTmp[k], items ); warm( S->itemH[k], items ); warm( S->prodPool[k], 2 * items );
    warm( S->unitOfItem[k], items ); warm( S->itemAtE[k], blocked );
  }
  warm( S->itemHTmp, items ); warm( S->intraAll, 3 * items );
  warm( S->order, 2 * 32 * 32 );
  warm( S->parent, items ); warm( S->newIdx, items ); warm( S->perm, items ); warm( S->inv, items ); warm( S->unitCount, items );
  warm( S->unitOfRoot, items ); warm( S->target, items ); warm( S->csProdPool, items ); warm( S->unitsDev, items / 8 );
  S->units.reserve( items / 8 ); S->unitsTmp.reserve( items / 8 );
}
void vvr_scratch_destroy( PrepScratch* s ) { delete s; }

// ---------------------------------------------------------------------------------------------------------------------
// validation
// ---------------------------------------------------------------------------------------------------------------------
#define FAIL( code, msg ) do { err = ( msg ); return ( code ); } while( 0 )

// the header, the tables and the presence of the arrays: a few hundred bytes, checked where the picture is submitted
static bool wpOnAny( const vvr_pic_header& h ) { return ( h.tool_flags & VVR_TOOL_WP ) && h.slice_type != 2; }
int vvr_host_validate_header( const vvr_config& cfg, const vvr_picture* p, std::string& err )
{
  vvr_pic_header h = p->hdr;           // (working copy: with slice headers the slice-level switches become their union)
  if( h.abi_version != VVR_ABI_VERSION ) FAIL( VVR_ERR_PARAMETER, "abi_version mismatch" );
  // (a picture may be smaller than the context's pictures - a coded video sequence with reference picture resampling changes its picture size -:
  // it occupies the top left corner of its DPB slot)
  if( !h.width || !h.height || h.width > cfg.max_width || h.height > cfg.max_height || h.chroma_format != cfg.chroma_format || h.bit_depth != cfg.bit_depth || h.log2_ctu != cfg.log2_ctu )
    FAIL( VVR_ERR_PARAMETER, "picture geometry differs from the context configuration" );
  if( ( h.width & 7 ) || ( h.height & 7 ) ) FAIL( VVR_ERR_PARAMETER, "picture size must be a multiple of 8 (minimum CU size)" );
  if( h.out_slot < 0 || h.out_slot >= cfg.num_slots ) FAIL( VVR_ERR_PARAMETER, "out_slot out of range" );
  if( h.slice_type > 2 ) FAIL( VVR_ERR_PARAMETER, "unknown slice type" );
  if( h.ladf_num_intervals == 1 || h.ladf_num_intervals > 5 ) FAIL( VVR_ERR_PARAMETER, "LADF: 2..5 intervals" );
  if( ( h.tool_flags & VVR_TOOL_COL_MOTION ) && !p->motion ) FAIL( VVR_ERR_PARAMETER, "collocated motion requested without a motion field" );
  if( h.wrap_offset && ( ( h.wrap_offset & 7 ) || h.wrap_offset < ( 1 << h.log2_ctu ) + 16 || h.wrap_offset > h.width ) ) FAIL( VVR_ERR_PARAMETER, "reference wrap-around offset: a multiple of 8 between CTU size + 16 and the picture width" );
  if( p->num_subpics > 1 )
  {
    // sub-pictures: rectangles of whole CTUs that tile the picture
    if( !p->subpics || p->num_subpics > 255 ) FAIL( VVR_ERR_PARAMETER, "sub-pictures: at most 255, with their table" );
    // a sub-picture consists of whole slices (VVC 6.3.1), and the availability rule of the intra stage looks at slices and tiles only: without
    // the slice map intra prediction would read across sub-picture boundaries
    if( !p->ctu_slice ) FAIL( VVR_ERR_PARAMETER, "sub-pictures need the slice map (ctu_slice): a sub-picture consists of whole slices" );
    if( h.wrap_offset ) FAIL( VVR_ERR_UNSUPPORTED, "sub-pictures together with reference wrap-around (the reference decoder does not support the pair either)" );
    const int ctuM = ( 1 << h.log2_ctu ) - 1;
    uint64_t area = 0;
    for( uint32_t k = 0; k < p->num_subpics; k++ )
    {
      const vvr_subpic& sp = p->subpics[k];
      if( ( sp.x0 & ctuM ) || ( sp.y0 & ctuM ) || sp.x1 < sp.x0 || sp.y1 < sp.y0 || sp.x1 >= h.width || sp.y1 >= h.height
       || ( sp.x1 != h.width - 1 && ( ( sp.x1 + 1 ) & ctuM ) ) || ( sp.y1 != h.height - 1 && ( ( sp.y1 + 1 ) & ctuM ) ) ) FAIL( VVR_ERR_PARAMETER, "sub-picture rectangle off the CTU grid or outside the picture" );
      for( uint32_t j = 0; j < k; j++ ) { const vvr_subpic& o = p->subpics[j]; if( sp.x0 <= o.x1 && o.x0 <= sp.x1 && sp.y0 <= o.y1 && o.y0 <= sp.y1 ) FAIL( VVR_ERR_PARAMETER, "sub-pictures overlap" ); }
      area += (uint64_t) ( sp.x1 - sp.x0 + 1 ) * ( sp.y1 - sp.y0 + 1 );
    }
    if( area != (uint64_t) h.width * h.height ) FAIL( VVR_ERR_PARAMETER, "sub-pictures do not cover the picture" );
  }
  if( h.num_ver_vb > 3 || h.num_hor_vb > 3 ) FAIL( VVR_ERR_PARAMETER, "at most three virtual boundaries per direction" );
  for( int d = 0; d < 2; d++ )
  {
    const uint16_t* pos = d ? h.vb_pos_y : h.vb_pos_x; const int n = d ? h.num_hor_vb : h.num_ver_vb, lim = d ? h.height : h.width;
    for( int i = 0; i < n; i++ )
      if( ( pos[i] & 7 ) || pos[i] == 0 || pos[i] >= lim || ( i && pos[i] <= pos[i - 1] ) ) FAIL( VVR_ERR_PARAMETER, "virtual boundary positions: multiples of 8 inside the picture, ascending" );
  }
  if( p->slices )
  {
    // slices with headers of their own (ABI 4)
    const int numCtuV = ( ( h.width + ( 1 << h.log2_ctu ) - 1 ) >> h.log2_ctu ) * ( ( h.height + ( 1 << h.log2_ctu ) - 1 ) >> h.log2_ctu );
    if( !p->ctu_slice || !p->num_slices || p->num_slices > 256 ) FAIL( VVR_ERR_PARAMETER, "slice headers: 1..256 of them, with the slice map (ctu_slice)" );
    for( int a = 0; a < numCtuV; a++ ) if( p->ctu_slice[a] >= p->num_slices ) FAIL( VVR_ERR_PARAMETER, "ctu_slice names a slice without a header" );
    if( p->num_alf_sets > 64 || p->num_wp_sets > 64 ) FAIL( VVR_ERR_PARAMETER, "at most 64 ALF / weighted-prediction tables" );
    uint32_t any = 0;
    for( uint32_t i = 0; i < p->num_slices; i++ )
    {
      const vvr_slice_header& sh = p->slices[i];
      any |= sh.tool_flags & VVR_SLICE_TOOL_MASK;
      if( sh.slice_type > 2 ) FAIL( VVR_ERR_PARAMETER, "unknown slice type" );
      if( ( sh.tool_flags & VVR_TOOL_LMCS_CSCALE ) && !( sh.tool_flags & VVR_TOOL_LMCS ) ) FAIL( VVR_ERR_PARAMETER, "LMCS chroma residual scaling without LMCS" );
      if( sh.alf_set >= std::max<uint32_t>( 1, p->num_alf_sets ) || sh.wp_set >= std::max<uint32_t>( 1, p->num_wp_sets ) ) FAIL( VVR_ERR_PARAMETER, "slice header names an ALF / weight table that is not there" );
    }
    h.tool_flags = ( h.tool_flags & ~(uint32_t) VVR_SLICE_TOOL_MASK ) | any;       // (the checks below: is the table there when ANY slice uses the tool)
  }
  for( uint32_t k = 0; wpOnAny( h ) && p->wp && k < std::max<uint32_t>( 1, p->num_wp_sets ); k++ )
    if( p->wp[k].log2_denom[0] > 7 || p->wp[k].log2_denom[1] > 7 ) FAIL( VVR_ERR_PARAMETER, "weighted prediction: log2 denominator out of range" );
  if( ( h.tool_flags & VVR_TOOL_LMCS_CSCALE ) && !( h.tool_flags & VVR_TOOL_LMCS ) ) FAIL( VVR_ERR_PARAMETER, "LMCS chroma residual scaling without LMCS" );
  if( ( h.tool_flags & VVR_TOOL_LMCS ) && !p->lmcs ) FAIL( VVR_ERR_PARAMETER, "LMCS enabled without tables" );
  const bool wpOn = ( h.tool_flags & VVR_TOOL_WP ) && h.slice_type != 2;
  if( wpOn && !p->wp ) FAIL( VVR_ERR_PARAMETER, "weighted prediction enabled without the weight table" );

  if( ( h.tool_flags & VVR_TOOL_SCALING_LIST ) && !p->scaling ) FAIL( VVR_ERR_PARAMETER, "explicit scaling lists enabled without the lists" );
  if( h.tool_flags & VVR_TOOL_SCALING_LIST )
    for( int id = 0; id < 28; id++ )
    {
      // (a 4:0:0 sequence sends the luma matrices only - ids 2, 5, 8, .. and 27, ScalingList::isLumaScalingList -, the others are never looked at and may hold anything:
      // found with the first parser-fed 4:0:0 stream that had scaling lists, round 4)
      if( !h.chroma_format && !( id % 3 == 2 || id == 27 ) ) continue;
      for( int k = 0; k < ( id < 2 ? 4 : id < 8 ? 16 : 64 ); k++ ) if( !p->scaling->coef[id][k] ) FAIL( VVR_ERR_PARAMETER, "scaling list entry 0" );
    }
  if( !p->cu || !p->tu || !p->coef || ( !( h.tool_flags & VVR_TOOL_LFP_ON_DEVICE ) && ( !p->lfp[0] || !p->lfp[1] ) ) ) FAIL( VVR_ERR_PARAMETER, "missing arrays" );
  if( ( h.tool_flags & VVR_TOOL_ALF ) && ( !p->alf || !p->alf_params ) ) FAIL( VVR_ERR_PARAMETER, "ALF enabled without parameters" );
  if( ( h.tool_flags & ( VVR_TOOL_SAO_LUMA | VVR_TOOL_SAO_CHROMA ) ) && !p->sao ) FAIL( VVR_ERR_PARAMETER, "SAO enabled without parameters" );
  if( h.slice_type != 2 )
    for( int l = 0; l < 2; l++ )
    {
      if( h.num_ref[l] < 0 || h.num_ref[l] > VVR_MAX_REFS ) FAIL( VVR_ERR_PARAMETER, "bad number of reference pictures" );
      for( int i = 0; i < h.num_ref[l]; i++ )
        if( h.ref_slot[l][i] < 0 || h.ref_slot[l][i] >= cfg.num_slots || h.ref_slot[l][i] == h.out_slot ) FAIL( VVR_ERR_PARAMETER, "bad reference slot" );
    }
  if( p->rpr && h.slice_type != 2 )
  {
    // reference picture resampling: the table is consistent in itself, and the pair of tools the reference does not combine it with is refused
    const vvr_rpr_params& R = *p->rpr;
    const int unit = h.chroma_format ? 2 : 1;
    if( h.wrap_offset ) FAIL( VVR_ERR_UNSUPPORTED, "scaled reference pictures together with reference wrap-around (the reference keeps no wrap copy of a scaled picture, Picture.h:278)" );
    for( uint32_t k = 0; k < p->num_subpics && p->num_subpics > 1; k++ ) if( p->subpics[k].treated_as_pic ) FAIL( VVR_ERR_UNSUPPORTED, "scaled reference pictures together with sub-pictures treated as pictures" );
    if( R.win_left % unit || R.win_top % unit ) FAIL( VVR_ERR_PARAMETER, "scaling window offsets are multiples of the chroma sub-sampling" );
    for( int l = 0; l < 2; l++ ) for( int i = 0; i < h.num_ref[l]; i++ )
    {
      const vvr_rpr_ref& r = R.ref[l][i];
      if( !r.width || !r.height || ( r.width & 7 ) || ( r.height & 7 ) || r.width > cfg.max_width || r.height > cfg.max_height ) FAIL( VVR_ERR_PARAMETER, "reference picture size: a multiple of 8 within the context's picture size" );
      if( r.win_left % unit || r.win_top % unit ) FAIL( VVR_ERR_PARAMETER, "scaling window offsets are multiples of the chroma sub-sampling" );
      // CU::getRprScaling (UnitTools.cpp:113-116): the reference picture is at most twice and at least an eighth as large as the current picture
      if( r.ratio[0] < ( 1 << 11 ) || r.ratio[0] > ( 1 << 15 ) || r.ratio[1] < ( 1 << 11 ) || r.ratio[1] > ( 1 << 15 ) ) FAIL( VVR_ERR_PARAMETER, "scaling ratio outside 1/8 .. 2" );
      if( r.scaled > 1 || r.hor_collocated_chroma > 1 || r.ver_collocated_chroma > 1 ) FAIL( VVR_ERR_PARAMETER, "reference picture resampling: flags are 0 or 1" );
      if( !r.scaled && ( r.width != h.width || r.height != h.height || r.win_left != R.win_left || r.win_top != R.win_top || r.ratio[0] != ( 1 << 14 ) || r.ratio[1] != ( 1 << 14 ) ) )
        FAIL( VVR_ERR_PARAMETER, "a reference picture of another size, scaling window or ratio is a scaled one (Picture::isRefScaled)" );
      for( int l2 = 0; l2 < 2; l2++ ) for( int j = 0; j < h.num_ref[l2]; j++ )
        if( h.ref_slot[l2][j] == h.ref_slot[l][i] && memcmp( &R.ref[l2][j], &r, sizeof( r ) ) ) FAIL( VVR_ERR_PARAMETER, "one reference picture described in two ways" );
    }
  }
  return VVR_OK;
}

// the CU / TU records (0.3 ms for a 4K picture): by the thread that builds the picture's work lists
// the CU / TU records [cu0, cu1) (a picture's records are checked in one go, or in parts by the threads that build its work lists in parts);
// area[0 / 1]: luma / chroma area the CUs of the range cover
static int validate_records_range( const vvr_picture* p, uint32_t cu0, uint32_t cu1, uint64_t area[2], std::string& err )
{
  vvr_pic_header h = p->hdr;
  if( p->slices ) { uint32_t any = 0; for( uint32_t i = 0; i < p->num_slices; i++ ) any |= p->slices[i].tool_flags & VVR_SLICE_TOOL_MASK; h.tool_flags = ( h.tool_flags & ~(uint32_t) VVR_SLICE_TOOL_MASK ) | any; }
  const bool wpOn = ( h.tool_flags & VVR_TOOL_WP ) && h.slice_type != 2;
  const int ncomp = h.chroma_format ? 3 : 1;
  uint64_t areaLuma = 0, areaChroma = 0;
  const bool lfpOnDev = ( h.tool_flags & VVR_TOOL_LFP_ON_DEVICE ) && !( h.tool_flags & VVR_TOOL_DEBLOCK_OFF );
  if( lfpOnDev && p->num_cu >= ( 1u << 22 ) ) FAIL( VVR_ERR_UNSUPPORTED, "more than 4M coding units (the per-cell records of k_lf_maps hold the CU index in 22 bits)" );
  for( uint32_t i = cu0; i < cu1; i++ )
  {
    const vvr_cu& cu = p->cu[i];
    if( !cu.w || !cu.h || cu.x + cu.w > h.width || cu.y + cu.h > h.height || cu.first_tu + cu.num_tu > p->num_tu ) FAIL( VVR_ERR_PARAMETER, "CU outside the picture / bad TU range" );
    if( cu.tree != VVR_TREE_CHROMA ) areaLuma += (uint64_t) cu.w * cu.h;
    if( cu.tree != VVR_TREE_LUMA ) areaChroma += (uint64_t) cu.w * cu.h;
    // ---- the CU's transform units: inside the CU, owned by it, tiling it, coded corners inside the level stream
    uint32_t tuAreaL = 0, tuAreaC = 0;
    for( uint32_t t = cu.first_tu; t < cu.first_tu + cu.num_tu; t++ )
    {
      const vvr_tu& tu = p->tu[t];
      // (edge parameters derived on the device: the per-cell records of k_lf_maps hold the transform unit's size in 7 bits - units are at most 64 wide and high)
      if( lfpOnDev && ( tu.w > 64 || tu.h > 64 ) ) FAIL( VVR_ERR_PARAMETER, "transform unit larger than 64 samples (VVR_TOOL_LFP_ON_DEVICE)" );
      if( tu.comp_mask & 1 ) tuAreaL += (uint32_t) tu.w * tu.h;
      if( tu.comp_mask & 6 ) tuAreaC += cu.isp_mode ? (uint32_t) cu.w * cu.h : (uint32_t) tu.w * tu.h;      // ISP: the unsplit chroma blocks sit in the last TU
      if( tu.cu != i ) FAIL( VVR_ERR_PARAMETER, "TU does not name its CU" );
      if( !tu.w || !tu.h || tu.x < cu.x || tu.y < cu.y || tu.x + tu.w > cu.x + cu.w || tu.y + tu.h > cu.y + cu.h ) FAIL( VVR_ERR_PARAMETER, "TU outside its CU" );
      if( tu.w > 64 || tu.h > 64 || ( tu.comp_mask & ~( ncomp == 3 ? 7 : 1 ) ) ) FAIL( VVR_ERR_PARAMETER, "TU larger than 64 samples or with components the format does not have" );
      if( tu.joint_cbcr > 3 ) FAIL( VVR_ERR_PARAMETER, "TU: joint Cb-Cr mode out of range" );
      for( int c = 0; c < ncomp; c++ )
      {
        if( !( tu.comp_mask & ( 1 << c ) ) ) continue;
        // (joint Cb-Cr: the levels belong to Cb for modes 2 / 3, to Cr for mode 1)
        const bool coded = ( c && tu.joint_cbcr ) ? c == ( ( tu.joint_cbcr >> 1 ) ? 1 : 2 ) : ( ( tu.cbf >> c ) & 1 ) != 0;
        if( !coded || !( cu.flags & VVR_CU_ROOT_CBF ) ) continue;
        const int bw = ( ( c && cu.isp_mode ) ? cu.w : tu.w ) >> ( c ? 1 : 0 ), bh = ( ( c && cu.isp_mode ) ? cu.h : tu.h ) >> ( c ? 1 : 0 );
        if( tu.mts_idx[c] > VVR_MTS_DCT8_DCT8 || ( tu.tr_type[c] & 3 ) > 2 || ( tu.tr_type[c] >> 2 ) > 2 ) FAIL( VVR_ERR_PARAMETER, "TU: transform type out of range" );
        const int bdp = c ? cu.bdpcm[1] : cu.bdpcm[0];
        if( !bdp && ( tu.max_scan_x[c] >= bw || tu.max_scan_y[c] >= bh ) ) FAIL( VVR_ERR_PARAMETER, "TU: last significant position outside the block" );
        const uint64_t n = bdp ? (uint64_t) bw * bh : (uint64_t) ( tu.max_scan_x[c] + 1 ) * ( tu.max_scan_y[c] + 1 );
        if( (uint64_t) tu.coef_off[c] + n > p->num_coef ) FAIL( VVR_ERR_PARAMETER, "TU: coded corner outside the level stream" );
      }
    }
    if( ( cu.tree != VVR_TREE_CHROMA && tuAreaL != (uint32_t) cu.w * cu.h ) || ( ncomp == 3 && cu.tree != VVR_TREE_LUMA && tuAreaC != (uint32_t) cu.w * cu.h ) )
      FAIL( VVR_ERR_PARAMETER, "the TUs of a CU do not cover it" );
    if( cu.pred_mode == VVR_PRED_INTER )
    {
      const bool isDmvr = cu.mc_mode == VVR_MC_DMVR || cu.mc_mode == VVR_MC_DMVR_BDOF;
      const bool isAff = cu.mc_mode == VVR_MC_AFFINE;
      const bool isGeo = cu.mc_mode == VVR_MC_GEO;
      const bool isSbt = cu.mc_mode == VVR_MC_SBTMVP;
      if( h.slice_type == 2 ) FAIL( VVR_ERR_PARAMETER, "inter CU in an I picture" );
      if( cu.mc_mode != VVR_MC_UNI && cu.mc_mode != VVR_MC_BI && cu.mc_mode != VVR_MC_BDOF && !isDmvr && !isAff && !isGeo && !isSbt ) FAIL( VVR_ERR_PARAMETER, "unknown mc_mode" );
      if( isSbt != ( ( cu.flags & VVR_CU_SBTMVP ) != 0 ) || ( isSbt && ( !p->motion || cu.w < 8 || cu.h < 8 ) ) ) FAIL( VVR_ERR_PARAMETER, "SbTMVP CU: mc_mode / flag mismatch, missing motion field or CU smaller than 8x8" );
      if( isSbt )
        for( int y = 0; y < cu.h; y += 8 ) for( int x = 0; x < cu.w; x += 8 )
        {
          const vvr_motion& m = p->motion[(size_t) ( ( cu.y + y ) >> 2 ) * ( ( h.width + 3 ) >> 2 ) + ( ( cu.x + x ) >> 2 )];
          if( ( m.ref_idx[0] < 0 && m.ref_idx[1] < 0 ) || m.ref_idx[0] >= h.num_ref[0] || m.ref_idx[1] >= h.num_ref[1] ) FAIL( VVR_ERR_PARAMETER, "SbTMVP CU: bad sub-block motion" );
        }
      if( isGeo != ( ( cu.flags & VVR_CU_GEO ) != 0 ) ) FAIL( VVR_ERR_PARAMETER, "GPM CU: mc_mode / flag mismatch" );
      if( isGeo && ( h.tool_flags & VVR_TOOL_LFP_ON_DEVICE ) && !( h.tool_flags & VVR_TOOL_DEBLOCK_OFF ) )
      {
        // the edge parameters are derived by the back-end: it needs the motion the CU left in the motion field (which of its two predictions a cell keeps)
        if( !p->motion ) FAIL( VVR_ERR_PARAMETER, "GPM CU: missing motion field (VVR_TOOL_LFP_ON_DEVICE)" );
        for( int y = 0; y < cu.h; y += 4 ) for( int x = 0; x < cu.w; x += 4 )
        {
          const vvr_motion& m = p->motion[(size_t) ( ( cu.y + y ) >> 2 ) * ( ( h.width + 3 ) >> 2 ) + ( ( cu.x + x ) >> 2 )];
          if( m.ref_idx[0] >= h.num_ref[0] || m.ref_idx[1] >= h.num_ref[1] ) FAIL( VVR_ERR_PARAMETER, "GPM CU: bad motion" );
        }
      }
      if( isGeo )
      {
        if( cu.w < 8 || cu.h < 8 || cu.w > 64 || cu.h > 64 || cu.w >= 8 * cu.h || cu.h >= 8 * cu.w || cu.geo_split_dir >= 64 ) FAIL( VVR_ERR_PARAMETER, "GPM CU: size / split direction out of range" );
        for( int k = 0; k < 2; k++ )
        {
          const int l = ( cu.geo_dir_ref[k] >> 4 ) - 1, ri = cu.geo_dir_ref[k] & 15;
          if( l < 0 || l > 1 || ri >= h.num_ref[l] ) FAIL( VVR_ERR_PARAMETER, "GPM CU: bad reference" );
        }
      }
      if( isAff != ( ( cu.flags & VVR_CU_AFFINE ) != 0 ) || ( isAff && ( ( !p->motion && !( h.tool_flags & VVR_TOOL_AFFINE_MV_ON_DEVICE ) ) || cu.w < 8 || cu.h < 8 ) ) ) FAIL( VVR_ERR_PARAMETER, "affine CU: mc_mode / flag mismatch, missing motion field or CU smaller than 8x8" );
      if( isDmvr && ( !( h.tool_flags & VVR_TOOL_DMVR ) || ( cu.mc_mode == VVR_MC_DMVR_BDOF && !( h.tool_flags & VVR_TOOL_BDOF ) ) || cu.ref_idx[0] < 0 || cu.ref_idx[1] < 0 || cu.w < 8 || cu.h < 8 || cu.w * cu.h < 128 || cu.bcw_idx != 2 ) )
        FAIL( VVR_ERR_PARAMETER, "mc_mode DMVR on a CU that cannot use DMVR (UnitTools.cpp:1277)" );
      if( cu.mc_mode == VVR_MC_BDOF && ( !( h.tool_flags & VVR_TOOL_BDOF ) || cu.ref_idx[0] < 0 || cu.ref_idx[1] < 0 || cu.w < 8 || cu.h < 8 || cu.w * cu.h < 128 || cu.bcw_idx != 2 ) )
        FAIL( VVR_ERR_PARAMETER, "mc_mode BDOF on a CU that cannot use BDOF (InterPrediction.cpp:1407-1427)" );
      // CIIP: one transform unit - or, in a sequence whose largest transform is 32, the four (two) of a CU that is 64 wide and / or high (round 6: the CU is
      // predicted and blended as a whole, the residuals are added transform unit by transform unit, DecCu.cpp:449-470)
      if( ( cu.flags & VVR_CU_CIIP ) && ( ( cu.mc_mode != VVR_MC_UNI && cu.mc_mode != VVR_MC_BI ) || cu.w * cu.h < 64 || cu.w > 64 || cu.h > 64 || ( cu.num_tu != 1 && !tusAreTheSplitAt32( p, cu ) ) ) )
        FAIL( VVR_ERR_PARAMETER, "CIIP CU: needs plain uni/bi prediction, at least 64 luma samples, sides of at most 64 and one TU (or the split at the largest transform size)" );
      if( cu.bcw_idx > 4 ) FAIL( VVR_ERR_PARAMETER, "BCW index out of range" );
      for( int l = 0; l < 2; l++ ) if( cu.ref_idx[l] >= h.num_ref[l] ) FAIL( VVR_ERR_PARAMETER, "ref_idx out of range" );
      if( cu.ref_idx[0] < 0 && cu.ref_idx[1] < 0 && !isGeo && !isSbt ) FAIL( VVR_ERR_PARAMETER, "inter CU without reference" );
      // motion vectors are 18-bit quantities (Mv::clipToStorageBitDepth, Mv.h; the refined vectors of DMVR are clamped to that range and the padded local
      // copy holds +-2 samples around the START vector: beyond the range the reference itself stops, InterPrediction.cpp:1768)
      for( int l = 0; l < 2; l++ ) if( cu.ref_idx[l] >= 0 && !isGeo && !isSbt ) for( int k = 0; k < 2; k++ )
        if( cu.mv[l][0][k] < -( 1 << 17 ) || cu.mv[l][0][k] > ( 1 << 17 ) - 1 ) FAIL( VVR_ERR_PARAMETER, "motion vector outside the 18-bit range" );
      // (the switch and the table are those of the CU's slice)
      const vvr_slice_header* sh = p->slices ? &p->slices[p->ctu_slice[( cu.y >> h.log2_ctu ) * ( ( h.width + ( 1 << h.log2_ctu ) - 1 ) >> h.log2_ctu ) + ( cu.x >> h.log2_ctu )]] : nullptr;
      if( ( sh ? ( sh->tool_flags & VVR_TOOL_WP ) != 0 : wpOn ) && cu.ref_idx[0] >= 0 && cu.ref_idx[1] >= 0 )
      {
        // weighted prediction: BDOF / DMVR only between references with default weights (InterPrediction.cpp:1420, UnitTools.cpp:1297-1302);
        // no identical-motion shortcut (:408)
        const vvr_wp_params& wpT = p->wp[sh && p->num_wp_sets > 1 ? sh->wp_set : 0];
        bool present = false;
        for( int l = 0; l < 2; l++ ) for( int k = 0; k < 3; k++ ) present |= wpT.e[l][cu.ref_idx[l]][k].present != 0;
        if( present && ( isDmvr || cu.mc_mode == VVR_MC_BDOF ) ) FAIL( VVR_ERR_PARAMETER, "mc_mode BDOF / DMVR between references with explicit prediction weights" );
        if( cu.mc_mode == VVR_MC_UNI ) FAIL( VVR_ERR_PARAMETER, "mc_mode UNI on a bi-predicted CU of a picture with weighted prediction" );
      }
      if( p->rpr && ( isDmvr || cu.mc_mode == VVR_MC_BDOF ) && ( p->rpr->ref[0][cu.ref_idx[0]].scaled || p->rpr->ref[1][cu.ref_idx[1]].scaled ) )
        FAIL( VVR_ERR_PARAMETER, "mc_mode BDOF / DMVR on a CU with a scaled reference picture (InterPrediction.cpp:1431-1435)" );
      if( cu.tree != VVR_TREE_JOINT && h.chroma_format ) FAIL( VVR_ERR_PARAMETER, "inter CU must be single tree" );
      if( cu.w == 4 && cu.h == 4 ) FAIL( VVR_ERR_PARAMETER, "4x4 inter CU (never inter predicted, InterPrediction.cpp:634)" );
    }
    else if( cu.pred_mode == VVR_PRED_INTRA )
    {
      if( cu.isp_mode )
      {
        // intra sub-partitions (CU::canUseISP, UnitTools.cpp): luma split in four, chroma unsplit in the last TU
        const uint32_t np = ( ( cu.w == 4 && cu.h == 8 ) || ( cu.w == 8 && cu.h == 4 ) ) ? 2 : 4;      // 4x8 / 8x4: two partitions
        if( cu.isp_mode > 2 || cu.multi_ref_idx || cu.bdpcm[0] || ( cu.flags & VVR_CU_MIP ) || cu.num_tu != np || cu.w * cu.h <= 16 ) FAIL( VVR_ERR_PARAMETER, "ISP CU: bad split mode, combined with MRL / BDPCM / MIP, or wrong number of TUs" );
        for( uint32_t k = 0; k < np; k++ )
        {
          const vvr_tu& t4 = p->tu[cu.first_tu + k];
          const bool ok = cu.isp_mode == 1 ? ( t4.x == cu.x && t4.w == cu.w && t4.h * (int) np == cu.h && t4.y == cu.y + (int) k * t4.h ) : ( t4.y == cu.y && t4.h == cu.h && t4.w * (int) np == cu.w && t4.x == cu.x + (int) k * t4.w );
          if( !ok || ( t4.comp_mask & 6 ) != ( k == np - 1 && h.chroma_format && cu.tree == VVR_TREE_JOINT ? 6 : 0 ) || t4.mts_idx[0] == VVR_MTS_SKIP ) FAIL( VVR_ERR_PARAMETER, "ISP CU: TU layout" );
        }
      }
      if( cu.flags & VVR_CU_MIP )
      {
        const int sizeId = ( cu.w == 4 && cu.h == 4 ) ? 0 : ( cu.w == 4 || cu.h == 4 || ( cu.w == 8 && cu.h == 8 ) ) ? 1 : 2;
        if( cu.intra_dir[0] >= ( sizeId == 0 ? 16 : sizeId == 1 ? 8 : 6 ) || cu.multi_ref_idx || cu.bdpcm[0] ) FAIL( VVR_ERR_PARAMETER, "MIP CU: mode index out of range for the block size, or combined with MRL / BDPCM" );
      }
      if( h.chroma_format && cu.tree != VVR_TREE_LUMA && cu.intra_dir[1] > 69 ) FAIL( VVR_ERR_PARAMETER, "chroma intra mode out of range" );
      {
        // luma-tree CUs go down to 4x4; CUs with chroma need 8 luma samples of width (no 2-wide intra chroma blocks) and 4 of height
        // (4:0:0: no CU carries chroma - found by the randomised GPU leg, tests/test_gpu_fuzz.py: a monochrome picture with 4-wide intra CUs was refused)
        const bool lumaOnly = cu.tree == VVR_TREE_LUMA || !h.chroma_format;
        const int minW = lumaOnly ? 4 : 8;
        // (a 128-wide or -high CU of a single tree is four or two transform units of 64: prediction and reconstruction go transform unit by transform unit,
        // DecCu::xIntraRecQT, so nothing here is larger than 64; found with the first parser-fed stream that left a CTU of 128 unsplit)
        if( cu.w > 128 || cu.h > 128 || cu.w < minW || cu.h < 4 || ( !lumaOnly && cu.w * cu.h < 64 ) ) FAIL( VVR_ERR_PARAMETER, "intra CU size out of range (4..128, with chroma at least 8 wide and 16 chroma samples)" );
        if( cu.w > 64 || cu.h > 64 )
          for( uint32_t t = cu.first_tu; t < cu.first_tu + cu.num_tu; t++ ) if( p->tu[t].w > 64 || p->tu[t].h > 64 ) FAIL( VVR_ERR_PARAMETER, "intra CU of more than 64 samples: transform units of at most 64 expected" );
      }
      if( cu.tree != VVR_TREE_JOINT )
      {
        // dual tree (I slices, qtbtt_dual_tree_intra_flag) and local dual tree (intra-only sub-trees of small blocks in any slice):
        // luma CUs carry luma blocks only, chroma CUs chroma blocks only
        if( cu.tree > VVR_TREE_CHROMA || !h.chroma_format ) FAIL( VVR_ERR_PARAMETER, "bad tree type" );
        const int want = cu.tree == VVR_TREE_LUMA ? 1 : 6;
        for( uint32_t t = cu.first_tu; t < cu.first_tu + cu.num_tu; t++ ) if( p->tu[t].comp_mask != want ) FAIL( VVR_ERR_PARAMETER, "separate-tree CU: TU component mask" );
        if( cu.tree == VVR_TREE_CHROMA && ( cu.isp_mode || cu.multi_ref_idx || cu.bdpcm[0] || ( cu.flags & VVR_CU_MIP ) ) ) FAIL( VVR_ERR_PARAMETER, "chroma-tree CU with luma tools" );
      }
      if( cu.intra_dir[0] > 66 || cu.multi_ref_idx > 2 ) FAIL( VVR_ERR_UNSUPPORTED, "bad intra mode" );
      if( cu.bdpcm[0] > 2 || cu.bdpcm[1] > 2 ) FAIL( VVR_ERR_PARAMETER, "bad BDPCM direction" );
      // (chroma BDPCM, round 4: the chroma blocks of the CU are predicted horizontally / vertically from the unfiltered neighbours and their transform-skip
      // levels accumulate along that direction - the same code paths as luma BDPCM, selected by bdpcm[1])
      if( cu.bdpcm[1] && ( ( cu.w >> 1 ) > 32 || ( cu.h >> 1 ) > 32 || cu.intra_dir[1] != ( cu.bdpcm[1] == 1 ? 18 : 50 ) ) ) FAIL( VVR_ERR_PARAMETER, "chroma BDPCM: block size / mode" );
    }
    else if( cu.pred_mode == VVR_PRED_IBC )
    {
      // intra block copy (InterPrediction::xIntraBlockCopy, InterPrediction.cpp:1995): integer block vector in mv[0][0], luma at most 64x64
      // (IBC_MAX_CU_SIZE), no intra / inter tools; sizes as for intra CUs.  One transform unit - or, in a sequence whose largest transform is 32, the four (two) of a
      // CU that is 64 wide and / or high (round 5, finding 13: such a CU was refused; the stage copies and reconstructs transform unit by transform unit anyway).
      // That the reference block precedes the CU in decoding order is checked where the work lists are built.
      if( !( h.tool_flags & VVR_TOOL_IBC ) ) FAIL( VVR_ERR_PARAMETER, "IBC CU in a picture without VVR_TOOL_IBC" );
      const bool lumaOnly = cu.tree == VVR_TREE_LUMA || !h.chroma_format;      // (4:0:0: no CU carries chroma)
      const int minW = lumaOnly ? 4 : 8;
      if( cu.tree == VVR_TREE_CHROMA || cu.w > 64 || cu.h > 64 || cu.w < minW || cu.h < 4 || ( !lumaOnly && cu.w * cu.h < 64 ) || cu.num_tu < 1 || cu.num_tu > 4 )
        FAIL( VVR_ERR_PARAMETER, "IBC CU: chroma tree, size out of range or a bad number of TUs" );
      if( cu.num_tu > 1 && !tusAreTheSplitAt32( p, cu ) ) FAIL( VVR_ERR_PARAMETER, "IBC CU: several TUs that are not the split at the largest transform size" );
      if( ( cu.mv[0][0][0] | cu.mv[0][0][1] ) & 15 ) FAIL( VVR_ERR_PARAMETER, "IBC CU: fractional block vector" );
      if( cu.isp_mode || cu.bdpcm[0] || cu.bdpcm[1] || cu.lfnst_idx || cu.sbt_info || ( cu.flags & ( VVR_CU_MIP | VVR_CU_CIIP | VVR_CU_AFFINE | VVR_CU_GEO | VVR_CU_SBTMVP ) ) )
        FAIL( VVR_ERR_PARAMETER, "IBC CU combined with an intra / inter tool" );
      if( cu.tree == VVR_TREE_LUMA && h.chroma_format )
        for( uint32_t t = cu.first_tu; t < cu.first_tu + cu.num_tu; t++ ) if( p->tu[t].comp_mask != 1 ) FAIL( VVR_ERR_PARAMETER, "separate-tree CU: TU component mask" );
      const int bvx = cu.mv[0][0][0] >> 4, bvy = cu.mv[0][0][1] >> 4, ctuS = 1 << h.log2_ctu, rowTop = cu.y & ~( ctuS - 1 );
      const int bufW = 256 * 128 / ctuS;                              // width of the IBC virtual buffer (Rom.h:210, CodingStructure.cpp:543)
      bool ok = cu.x + bvx >= 0 && cu.x + bvx + cu.w <= h.width && cu.y + bvy >= rowTop && cu.y + bvy + cu.h <= std::min<int>( h.height, rowTop + ctuS )
             && cu.x + bvx + cu.w <= ( ( cu.x >> h.log2_ctu ) + 1 ) * ctuS && cu.x + bvx >= ( cu.x & ~( ctuS - 1 ) ) - ( bufW - ctuS );
      if( ok && cu.tree == VVR_TREE_JOINT && h.chroma_format )
      {
        const int cxr = ( cu.x >> 1 ) + ( bvx >> 1 ), cyr = ( cu.y >> 1 ) + ( bvy >> 1 );
        ok = cxr >= 0 && cxr + ( cu.w >> 1 ) <= ( h.width >> 1 ) && cyr >= ( rowTop >> 1 ) && cyr + ( cu.h >> 1 ) <= std::min<int>( h.height, rowTop + ctuS ) >> 1 && 2 * cxr >= ( cu.x & ~( ctuS - 1 ) ) - ( bufW - ctuS );
      }
      if( !ok ) FAIL( VVR_ERR_PARAMETER, "IBC CU: reference block outside the picture, the CTU row or the reach of the IBC buffer" );
    }
    else FAIL( VVR_ERR_PARAMETER, "unknown prediction mode" );
  }
  area[0] = areaLuma; area[1] = areaChroma;
  return VVR_OK;
}
// the CUs tile the picture: with every CU inside the picture, equal areas leave no cell uncovered unless two CUs overlap (which the device
// tolerates: it only ever addresses samples inside the CUs it was given)
static int validate_cover( const vvr_picture* p, const uint64_t area[2], std::string& err )
{
  const vvr_pic_header& h = p->hdr;
  if( area[0] != (uint64_t) h.width * h.height || ( h.chroma_format && area[1] != (uint64_t) h.width * h.height ) ) FAIL( VVR_ERR_PARAMETER, "the CUs do not cover the picture" );
  return VVR_OK;
}
int vvr_host_validate_records( const vvr_config& cfg, const vvr_picture* p, std::string& err )
{
  (void) cfg;
  uint64_t area[2];
  const int rc = validate_records_range( p, 0, p->num_cu, area, err );
  return rc != VVR_OK ? rc : validate_cover( p, area, err );
}

int vvr_host_validate( const vvr_config& cfg, const vvr_picture* p, std::string& err )
{
  const int rc = vvr_host_validate_header( cfg, p, err );
  return rc != VVR_OK ? rc : vvr_host_validate_records( cfg, p, err );
}

// ---------------------------------------------------------------------------------------------------------------------
// work lists
// ---------------------------------------------------------------------------------------------------------------------
// The per-cell maps the intra-stage analysis looks things up in are filled CTU by CTU, right before the CTU's blocks are analysed: everything a
// block looks at lies in its own CTU, the CTU to the left or the CTU row above, so the lines it touches are still in the cache (a whole-picture
// pass in front would have been evicted again: the lookups are what this stage spends its time on).  Nothing is cleared per picture: cells of
// later CTUs are known to be "not decoded yet" from their position, the block map carries the picture's epoch.
int PrepScratch::beginMaps( const PrepScratch* like /* the same picture in another thread's scratch: what it found out about the picture as a whole */ )
{
  leafSort = false;
  if( like ) { anyIntra = like->anyIntra; allIntraCus = like->allIntraCus; leaf = like->leaf; }
  else
  {
  anyIntra = ( h.tool_flags & VVR_TOOL_LMCS_CSCALE ) != 0;      // (inter blocks with scaled chroma residuals are intra-stage items)
  for( uint32_t i = 0; i < p->num_cu && !anyIntra; i++ ) anyIntra = p->cu[i].pred_mode == VVR_PRED_INTRA || p->cu[i].pred_mode == VVR_PRED_IBC || ( p->cu[i].flags & VVR_CU_CIIP );
  allIntraCus = h.slice_type == 2;
  for( uint32_t i = 0; i < p->num_cu && allIntraCus; i++ ) allIntraCus = p->cu[i].pred_mode == VVR_PRED_INTRA;
  // scattered intra blocks: one wavefront per block (an IBC block copies samples from anywhere in its CTU row: such pictures keep the CTU-tile path)
  leaf = leafOn && anyIntra && !allIntraCus;
  if( leaf && ( h.tool_flags & VVR_TOOL_IBC ) ) for( uint32_t i = 0; i < p->num_cu && leaf; i++ ) leaf = p->cu[i].pred_mode != VVR_PRED_IBC;
  }
  fastCtu.assign( (size_t) numCtu, 0 );
  if( anyIntra )
  {
    const size_t cells = (size_t) w4 * h4;
    order.assign( (size_t) 2 << ( 2 * ( h.log2_ctu - 2 ) ), 0x7fffffff );
    if( !leaf )
    {
      epoch = ( epoch + 1 ) & 0x3ff;
      const size_t blocked = (size_t) numCtu << ( 2 * ( h.log2_ctu - 2 ) );
      for( int k = 0; k < ncomp; k++ ) if( itemAtE[k].size() != blocked || epoch == 0 ) itemAtE[k].assign( blocked, 0xffffffffu );
      if( epoch == 0 ) epoch = 1;
    }
  }
  if( cscale )
  {
    csVpduV.assign( (size_t) vpdusX * vpdusY, 0 );
    if( leaf ) csNeeded.assign( (size_t) vpdusX * vpdusY, 0 );
    else { csProdRange.assign( (size_t) vpdusX * vpdusY, std::make_pair( 0xffffffffu, 0u ) ); csProdPool.clear(); }
  }
  return VVR_OK;
}

// (leaf, one thread builds the whole picture) levels of the blocks: see `leafSort`
void PrepScratch::beginLevels()
{
  if( !leaf || !leafSortOn ) return;
  leafSort = true;
  levMap.assign( (size_t) 3 * w4 * h4, 0 );
  csLev.assign( cscale ? (size_t) vpdusX * vpdusY : 0, 0 );
  for( int k = 0; k < 3; k++ ) lev[k].clear();
}
// level of the item that computes the chroma scaling factor of VPDU `vp`: the luma column left of / row above the VPDU's first CU (what its wavefront polls)
uint16_t PrepScratch::csLevelOf( size_t vp )
{
  if( csLev[vp] ) return csLev[vp];
  const uint32_t v = csVpduV[vp];
  const int xPos = v & 0x1fff, yPos = ( v >> 13 ) & 0x1fff, n = 1 << vpduLog2;
  uint32_t d = 0;
  if( ( v >> 26 ) & 1 ) levLook( 0, ( xPos - 1 ) >> 2, yPos >> 2, ( xPos - 1 ) >> 2, std::min( yPos + n - 1, (int) h.height - 1 ) >> 2, d );
  if( ( v >> 27 ) & 1 ) levLook( 0, xPos >> 2, ( yPos - 1 ) >> 2, std::min( xPos + n - 1, (int) h.width - 1 ) >> 2, ( yPos - 1 ) >> 2, d );
  return csLev[vp] = (uint16_t) std::min<uint32_t>( d + 1, 0xfffe );
}

// decoding order of the transform blocks of CTU `ctuIdx` (CUs [i0, i1)), its cells covered by intra CUs, the luma neighbourhood of the chroma
// scaling factor of its VPDUs
int PrepScratch::mapCtu( uint32_t i0, uint32_t i1, uint32_t ctuIdx, std::string& err )
{
  curCtuIdx = ctuIdx;
  if( !anyIntra ) return VVR_OK;
  {
    bool fast = true;
    for( uint32_t i = i0; i < i1 && fast; i++ ) fast = p->cu[i].pred_mode == VVR_PRED_INTRA;
    if( leaf ) fast = false;        // (the CTU-tile path's shortcut for CTUs of intra CUs)
    fastCtu[ctuIdx] = fast;
    if( fast ) { memset( fastCell, 0xff, sizeof( fastCell ) ); for( int k = 0; k < 3; k++ ) fastFirst[k] = (uint32_t) intra[k].size(); }
  }
  const size_t cells = (size_t) w4 * h4;
  for( uint32_t i = i0; i < i1; i++ )
  {
    const vvr_cu& cu = p->cu[i];
    for( uint32_t t = cu.first_tu; t < cu.first_tu + cu.num_tu; t++ )
    {
      const vvr_tu& tu = p->tu[t];
      if( ( tu.comp_mask & 1 ) && ( tu.comp_mask & 6 ) && !cu.isp_mode )
      {
        // (the usual unit: both channels cover the same cells - one walk over its rows, the channel maps lie half the array apart)
        const int x0 = tu.x >> 2, x1 = std::min( ( tu.x + tu.w + 3 ) >> 2, w4 ), y1 = std::min( ( tu.y + tu.h + 3 ) >> 2, h4 );
        const size_t chroma = orderIdx( 1, 0, 0 ) - orderIdx( 0, 0, 0 );
        for( int y = tu.y >> 2; y < y1; y++ ) { int32_t* row = &order[orderIdx( 0, x0 << 2, y << 2 )]; std::fill( row, row + ( x1 - x0 ), (int32_t) t ); std::fill( row + chroma, row + chroma + ( x1 - x0 ), (int32_t) t ); }
        continue;
      }
      for( int chn = 0; chn < 2; chn++ )
      {
        if( chn == 0 && !( tu.comp_mask & 1 ) ) continue;
        if( chn == 1 && !( tu.comp_mask & 6 ) ) continue;
        int ax = tu.x, ay = tu.y, aw = tu.w, ah = tu.h;
        if( chn == 1 && cu.isp_mode ) { ax = cu.x; ay = cu.y; aw = cu.w; ah = cu.h; }      // ISP: the unsplit chroma blocks sit in the last TU
        const int x0 = ax >> 2, x1 = std::min( ( ax + aw + 3 ) >> 2, w4 ), y1 = std::min( ( ay + ah + 3 ) >> 2, h4 );
        for( int y = ay >> 2; y < y1; y++ ) { int32_t* row = &order[orderIdx( chn, x0 << 2, y << 2 )]; std::fill( row, row + ( x1 - x0 ), (int32_t) t ); }
      }
    }
  }
  if( cscale )
  {
    // the luma CU that covers a cell of this CTU: owner of the transform block recorded there
    // (asked about positions of this CTU only: the origin of a VPDU, and its left / above neighbour where that lies in the same CTU)
    auto cuAt = [&]( int x, int y ) -> int32_t { const int32_t t = order[orderIdx( 0, x, y )]; return ( t < 0 || (uint32_t) t >= p->num_tu ) ? -1 : (int32_t) p->tu[t].cu; };
    const int cx = (int) ( ctuIdx % ctusX ) << h.log2_ctu, cy = (int) ( ctuIdx / ctusX ) << h.log2_ctu, nv = 1 << ( h.log2_ctu - vpduLog2 );
    for( int jy = 0; jy < nv; jy++ ) for( int jx = 0; jx < nv; jx++ )
    {
      const int vxp = cx + ( jx << vpduLog2 ), vyp = cy + ( jy << vpduLog2 );
      if( vxp >= h.width || vyp >= h.height ) continue;
      const int32_t tl = cuAt( vxp, vyp );
      if( tl < (int32_t) i0 || tl >= (int32_t) i1 ) FAIL( VVR_ERR_PARAMETER, "no luma CU at the origin of a VPDU" );
      const int xPos = p->cu[tl].x, yPos = p->cu[tl].y;
      // the neighbouring CU has to be available: decoded before, same slice, same tile (getCURestricted in calculateChromaAdjVpduNei)
      bool hasLeft = xPos > 0 && sameSliceAndTile( ctuAt( xPos - 1, yPos ), ctuIdx ), hasAbove = yPos > 0 && sameSliceAndTile( ctuAt( xPos, yPos - 1 ), ctuIdx );
      if( hasLeft && ( ( xPos - 1 ) >> h.log2_ctu ) == ( xPos >> h.log2_ctu ) && cuAt( xPos - 1, yPos ) > tl ) hasLeft = false;
      if( hasAbove && ( ( yPos - 1 ) >> h.log2_ctu ) == ( yPos >> h.log2_ctu ) && cuAt( xPos, yPos - 1 ) > tl ) hasAbove = false;
      csVpduV[(size_t) ( vyp >> vpduLog2 ) * vpdusX + ( vxp >> vpduLog2 )] = (uint32_t) xPos | ( (uint32_t) yPos << 13 ) | ( hasLeft ? 1u << 26 : 0 ) | ( hasAbove ? 1u << 27 : 0 );
    }
  }
  return VVR_OK;
}

// the work lists: intra-stage blocks with the blocks they read from, motion-compensation tiles, transform blocks
int PrepScratch::buildWorkLists( std::string& err, uint32_t cu0, uint32_t cu1 )
{
  cu1 = std::min( cu1, p->num_cu );
  uint32_t curCtu = cu0 < cu1 ? (uint32_t) ( ( p->cu[cu0].y >> h.log2_ctu ) * ctusX + ( p->cu[cu0].x >> h.log2_ctu ) ) : 0, mappedEnd = cu0;
  for( uint32_t i = cu0; i < cu1; i++ )
  {
    const vvr_cu& cu = p->cu[i];
    // CTU bookkeeping for the per-CTU intra lists (CUs arrive in CTU raster order)
    const uint32_t ctuOfCu = (uint32_t) ( ( cu.y >> h.log2_ctu ) * ctusX + ( cu.x >> h.log2_ctu ) );
    {
      if( ctuOfCu < curCtu || ctuOfCu < partCtu0 || ctuOfCu >= partCtu1 ) FAIL( VVR_ERR_PARAMETER, "CUs are not in CTU raster order" );
      while( curCtu < ctuOfCu ) { curCtu++; for( int k = 0; k < 3; k++ ) ctuStartV[(size_t) k * ( numCtu + 1 ) + curCtu] = (uint32_t) intra[k].size(); }
    }
    if( i == mappedEnd )
    {
      // first CU of a CTU: map the CTU's cells before its blocks are analysed
      uint32_t j = i + 1;
      while( j < cu1 && (uint32_t) ( ( p->cu[j].y >> h.log2_ctu ) * ctusX + ( p->cu[j].x >> h.log2_ctu ) ) == ctuOfCu ) j++;
      const int rc = mapCtu( i, j, ctuOfCu, err );
      if( rc != VVR_OK ) return rc;
      mappedEnd = j;
    }
    const bool isCiipCu = cu.pred_mode == VVR_PRED_INTER && ( cu.flags & VVR_CU_CIIP );
    // LMCS chroma residual scaling of an inter block: its factor reads reconstructed luma that the intra stage may still have to
    // produce, and intra blocks next to it read its reconstructed chroma, so the residual add of such a block is an item of the
    // intra stage too (IT_MODE_RESI_ADD: no prediction, scaled residual onto the inter prediction; finishLMCSAndReco, DecCu.cpp:483)
    const bool isCsInterCu = cscaleCtu( ctuOfCu ) && cu.pred_mode == VVR_PRED_INTER && ( !isCiipCu || cu.w == 4 ) && ( cu.flags & VVR_CU_ROOT_CBF );
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
          // a CIIP CU of several transform units (the split at a largest transform size of 32): ONE block per component, the CU - it is predicted from the CU's
          // neighbours and blended as a whole (DecCu.cpp:449-470) -, carried by the first unit; which units bring a residual rides in the item's `tu` word
          const bool ciipCu = isCiip && cu.num_tu > 1;
          if( ciipCu && t != cu.first_tu ) continue;
          const bool isCsInter = cscaleCtu( ctuOfCu ) && cu.pred_mode == VVR_PRED_INTER && !isCiip && ( cu.flags & VVR_CU_ROOT_CBF );
          if( cu.pred_mode != VVR_PRED_INTRA && !isCiip && !isCsInter && !isIbcCu ) continue;
          if( isCsInter && ( !comp || !( ( ( tu.cbf >> comp ) & 1 ) || tu.joint_cbcr ) || ( tu.w >> 1 ) * ( tu.h >> 1 ) <= 4 ) ) continue;
          if( isCsInter )
          {
            // the scaled residual of an inter chroma block is added by k_resi_add between the luma and the chroma units of the stage: every luma
            // sample its factor is averaged over is final by then, and the chroma blocks that read its samples come later - no place in the
            // dependency graph, no producer analysis
            IntraItem it; memset( &it, 0, sizeof( it ) );
            it.tu = t; it.comp = (uint8_t) comp; it.x = (uint16_t) ( tu.x >> 1 ); it.y = (uint16_t) ( tu.y >> 1 ); it.lw = (uint8_t) ilog2i( tu.w >> 1 ); it.lh = (uint8_t) ilog2i( tu.h >> 1 );
            it.mode = IT_MODE_RESI_ADD; it.flags = IT_F_RESI | IT_F_CSCALE;
            if( leaf )
            {
              // one wavefront per block - but these blocks are many and small, and all they wait for is the factor of their VPDU: the wavefront that computes
              // the factor (IT_MODE_CSFAC item) adds the residuals of the VPDU's blocks as well (emitLeafItems groups them); `tu` carries the VPDU
              it.tu = (uint32_t) ( ( tu.y >> vpduLog2 ) * vpdusX + ( tu.x >> vpduLog2 ) );
              resiAdd.push_back( it );
              csNeeded[it.tu] = 1;
              if( leafSort ) levOwn( comp, it.x >> 1, it.y >> 1, ( it.x + ( tu.w >> 1 ) - 1 ) >> 1, ( it.y + ( tu.h >> 1 ) - 1 ) >> 1, csLevelOf( it.tu ) );      // (the factor's wavefront clears them)
              bytes[K_INTRA_LEAF] += (double) ( tu.w >> 1 ) * ( tu.h >> 1 ) * 6 + sizeof( IntraItem );
              continue;
            }
            resiAdd.push_back( it );
            bytes[K_RESI_ADD] += (double) ( tu.w >> 1 ) * ( tu.h >> 1 ) * 6 + sizeof( IntraItem );      // prediction read, residual read, sample written
            continue;
          }
          const int cs = comp ? 1 : 0, chn = comp ? 1 : 0, unit = 4 >> cs;
          // intra sub-partitions: luma partitions are blocks of their own that share the reference line of the whole CU
          // (initIntraPatternChTypeISP, IntraPrediction.cpp:966); partitions narrower than 4 are predicted in pairs (DecCu.cpp:333-371):
          // one item of width 4 carries both; the unsplit chroma blocks come with the last TU
          const bool ispL = cu.isp_mode && !comp, ispC = cu.isp_mode && comp;
          const bool ispPair = ispL && cu.isp_mode == 2 && tu.w < 4;                              // group of 4 / tu.w partitions
          if( ispPair && ( ( tu.x - cu.x ) & 3 ) ) continue;                                      // not the first of its group: part of the group's item
          const bool wholeCu = ispC || ciipCu;
          const int x0 = ( wholeCu ? cu.x : tu.x ) >> cs, y0 = ( wholeCu ? cu.y : tu.y ) >> cs, w = ispPair ? 4 : ( wholeCu ? cu.w : tu.w ) >> cs, hh = ( wholeCu ? cu.h : tu.h ) >> cs;
          // block whose neighbourhood decides the availability of the reference samples
          const int rx0 = ispL ? cu.x : x0, ry0 = ispL ? cu.y : y0, rw = ispL ? cu.w : w, rh = ispL ? cu.h : hh;
          const int32_t rcur = ispL ? (int32_t) cu.first_tu : (int32_t) t;
          const int totalAbove = ( 2 * rw + unit - 1 ) / unit, totalLeft = ( 2 * rh + unit - 1 ) / unit;
          IntraItem it; memset( &it, 0, sizeof( it ) );
          it.tu = t; it.comp = (uint8_t) comp;
          it.x = (uint16_t) x0; it.y = (uint16_t) y0;
          it.lw = (uint8_t) ilog2i( w ); it.lh = (uint8_t) ilog2i( hh );
          it.mode = isIbcCu ? IT_MODE_IBC : isCsInter ? IT_MODE_RESI_ADD : isCiip ? 0 : cu.intra_dir[chn];       // CIIP: planar
          // IBC: the block vector in samples of the component (chroma: halved, InterPrediction.cpp:2010-2011)
          const int ibcDx = isIbcCu ? ( cu.mv[0][0][0] >> 4 ) >> cs : 0, ibcDy = isIbcCu ? ( cu.mv[0][0][1] >> 4 ) >> cs : 0;
          if( isIbcCu ) it.tu = ( (uint32_t) ibcDx & 0xffff ) | ( (uint32_t) ibcDy << 16 );
          bool hasResi = ( ( tu.cbf >> comp ) & 1 ) || ( comp && tu.joint_cbcr );
          if( ciipCu )
          {
            // bit k: transform unit k of the CU (raster order) has a residual for this component; bits 4-5: units per row - 1; bit 31: the marker
            uint32_t mask = 0;
            for( uint32_t k = 0; k < cu.num_tu; k++ ) { const vvr_tu& tk = p->tu[cu.first_tu + k]; if( ( ( tk.cbf >> comp ) & 1 ) || ( comp && tk.joint_cbcr ) ) mask |= 1u << k; }
            it.tu = 0x80000000u | mask | ( (uint32_t) ( cu.w > 32 ? 1 : 0 ) << 4 );
            hasResi = mask != 0;
          }
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
          int cclmTop = 0, cclmLeft = 0, cclmBLeft = 0, lmTop = 0, lmLeft = 0; bool isCclm = false;
          const bool csItem = cscaleCtu( ctuOfCu ) && comp && hasResi && w * hh > 4;             // DecCu.cpp:383-388 / :500-505
          if( csItem ) it.flags |= IT_F_CSCALE;
          if( comp && !isCiip && !isCsInter && !isIbcCu && cu.intra_dir[1] >= 67 )
          {
            // CCLM / MDLM: template sizes and flags of IntraPrediction::xGetLMParameters (:1694-1800) and the border handling of
            // xGetLumaRecPixels (:1403-1470); they ride in the item's `tu` word
            const int mode = cu.intra_dir[1];
            // cu.above / cu.left: the neighbouring CU exists in this slice and tile (for a block inside its CU: the CU itself lies above / left)
            const bool aboveCu = ( y0 << 1 ) > cu.y || ( cu.y > 0 && sameSliceAndTile( ctuAt( cu.x, cu.y - 1 ), ctuOfCu ) );
            const bool leftCu = ( x0 << 1 ) > cu.x || ( cu.x > 0 && sameSliceAndTile( ctuAt( cu.x - 1, cu.y ), ctuOfCu ) );
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
            cclmTop = aboveAvail ? actualTop : 0; cclmLeft = leftAvail ? actualLeft : 0; cclmBLeft = bLeft; isCclm = true; lmTop = actualTop; lmLeft = actualLeft;
          }
          // ---- the blocks this one reads from, its part of the CTU tile
          const uint32_t myId = (uint32_t) intra[comp].size();
          if( myId >= 0x3fffffu ) FAIL( VVR_ERR_UNSUPPORTED, "too many intra-stage blocks" );
          intra[comp].push_back( it );
          if( leaf )
          {
            // one wavefront per block, ordered on the device: the list in decoding order is all there is to do
            if( csItem ) csNeeded[(size_t) ( tu.y >> vpduLog2 ) * vpdusX + ( tu.x >> vpduLog2 )] = 1;
            bytes[K_INTRA_LEAF] += (double) w * hh * ( hasResi ? 4 : 2 ) + sizeof( IntraItem );
            if( leafSort )
            {
              // the cells the block's wavefront polls (k_intra_leaf, "wait for the blocks that produce what this one reads"): corner, above and left reference
              // lines, the VPDU's factor, the luma a cross-component prediction reads; the later partitions of an ISP coding unit ride with the first
              const int u = comp ? 1 : 2;
              if( ispL && ( x0 != rx0 || y0 != ry0 ) ) lev[comp].push_back( ispLev );
              else
              {
                uint32_t d = 0;
                const int mrl = ( comp || ( it.flags & IT_F_MIP ) ) ? 0 : ( it.flags >> 4 ) & 3;
                const int qx = rx0 - 1 - mrl, qy = ry0 - 1 - mrl, szA = std::min( it.nA * unit, 2 * rw ), szL = std::min( it.nL * unit, 2 * rh );
                if( it.nTL ) levLook( comp, qx >> u, qy >> u, qx >> u, qy >> u, d );
                if( it.nA ) levLook( comp, rx0 >> u, qy >> u, ( rx0 + szA - 1 ) >> u, qy >> u, d );
                if( it.nL ) levLook( comp, qx >> u, ry0 >> u, qx >> u, ( ry0 + szL - 1 ) >> u, d );
                if( csItem ) d = std::max<uint32_t>( d, csLevelOf( (size_t) ( tu.y >> vpduLog2 ) * vpdusX + ( tu.x >> vpduLog2 ) ) );
                if( isCclm )
                {
                  const int lx0 = x0 << 1, ly0 = y0 << 1;
                  levLook( 0, lx0 >> 2, ly0 >> 2, ( lx0 + 2 * w - 1 ) >> 2, ( ly0 + 2 * hh - 1 ) >> 2, d );
                  if( ly0 > 0 ) levLook( 0, ( lx0 - 4 ) >> 2, ( ly0 - 4 ) >> 2, ( lx0 + 2 * std::max( w, lmTop ) - 1 ) >> 2, ( ly0 - 4 ) >> 2, d );
                  if( lx0 > 0 ) levLook( 0, ( lx0 - 4 ) >> 2, ly0 >> 2, ( lx0 - 4 ) >> 2, ( ly0 + 2 * std::max( hh, lmLeft ) - 1 ) >> 2, d );
                }
                const uint16_t lv = (uint16_t) std::min<uint32_t>( d + 1, 0xfffe );
                lev[comp].push_back( lv ); ispLev = lv;
                levOwn( comp, rx0 >> u, ry0 >> u, ( rx0 + rw - 1 ) >> u, ( ry0 + rh - 1 ) >> u, lv );
              }
            }
            continue;
          }
          std::vector<uint32_t>& pool = prodPool[comp];
          ItemH IH; IH.ctu = ctuOfCu; IH.p0 = (uint32_t) pool.size(); IH.pn = 0;
          {
            const int ctuX = cu.x >> h.log2_ctu, ctuY = cu.y >> h.log2_ctu;
            const int mrl = ( comp || isIbcCu ) ? 0 : cu.multi_ref_idx;
            uint32_t lastKey = 0xffffffffu;
            // (returns the block found at the sample when it belongs to this band's map, else -1: the walks along a reference line skip the rest of that block)
            auto touch = [&]( int k, int xc, int yc ) -> int32_t   // component k, component coordinates of a sample that is read
            {
              const int sh = k ? 1 : 0, lx = xc << sh, ly = yc << sh;
              if( lx < 0 || ly < 0 || lx >= h.width || ly >= h.height ) return -1;
              if( ly < partTopY )
              {
                // a cell of the band above (its last cell row: nothing else is ever read across a CTU row): looked up when the bands are joined
                if( ( ly >> 2 ) != ( partTopY >> 2 ) - 1 ) { partBad = true; return -1; }
                const uint32_t pe = ( (uint32_t) k << 16 ) | (uint32_t) ( lx >> 2 );
                if( !pending.empty() && pending.back() == pe && pool.size() > IH.p0 && pool.back() == ( 0x80000000u | (uint32_t) ( pending.size() - 1 ) ) ) return -1;
                pool.push_back( 0x80000000u | (uint32_t) pending.size() ); pending.push_back( pe ); lastKey = 0xffffffffu;
                return -1;
              }
              const int32_t d = itemAtGet( k, cellIdx( lx >> 2, ly >> 2 ) );
              if( d < 0 ) return -1;
              const uint32_t key = ( (uint32_t) k << 28 ) | (uint32_t) d;
              if( key == lastKey || ( k == comp && (uint32_t) d == myId ) ) return d;      // (neighbouring cells mostly belong to the same block)
              lastKey = key;
              if( std::find( pool.begin() + IH.p0, pool.end(), key ) == pool.end() ) pool.push_back( key );
              return d;
            };
            // a reference line, cell by cell - but a block found on the line is known with its extent: the cells it covers further along the line name it again
            // and are passed over (blocks whose edges lie on the cell grid; the narrow ones of ISP CUs and 2-wide chroma blocks share cells and are walked)
            auto walk = [&]( int n, bool alongX, int xc, int yc )
            {
              for( int k = 0; k < n; )
              {
                const int32_t d = touch( comp, alongX ? xc + k : xc, alongX ? yc : yc + k );
                int nk = k + unit;
                if( d >= 0 )
                {
                  const IntraItem& pb = intra[comp][d];
                  const int b0 = alongX ? pb.x : pb.y, bs = 1 << ( alongX ? pb.lw : pb.lh );
                  const bool ispBlock = !comp && ( pb.flags & ~IT_F_RESI ) == IT_F_ISP;
                  if( !( ( b0 << cs ) & 3 ) && !( ( bs << cs ) & 3 ) && !ispBlock ) nk = std::max( nk, b0 + bs - ( alongX ? xc : yc ) );
                }
                k = nk;
              }
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
            if( !fastCtu[ctuOfCu] )
            {
            if( it.nTL ) touch( comp, rx0 - 1 - mrl, ry0 - 1 - mrl );
            walk( it.nA * unit, true, rx0, ry0 - 1 - mrl );
            walk( it.nL * unit, false, rx0 - 1 - mrl, ry0 );
            if( ispL && ( x0 != rx0 || y0 != ry0 ) ) touch( 0, cu.isp_mode == 2 ? x0 - 1 : x0, cu.isp_mode == 2 ? y0 : y0 - 1 );   // ISP: the previous partition
            if( isIbcCu )
            {
              // the reference block: every cell must precede this block in decoding order; the intra-stage blocks that produce it are producers
              const int qx = x0 + ibcDx, qy = y0 + ibcDy;
              for( int yy = 0; yy < hh + unit - 1; yy += unit ) for( int xx = 0; xx < w + unit - 1; xx += unit )
              {
                const int sx = qx + std::min( xx, w - 1 ), sy = qy + std::min( yy, hh - 1 );
                if( !unitAvail( chn, sx, sy, (int32_t) t ) ) FAIL( VVR_ERR_PARAMETER, "IBC CU: the reference block is not reconstructed before the CU" );
                touch( comp, sx, sy );
              }
            }
            if( csItem )
            {
              // luma the chroma scaling factor is averaged over (the unit must wait for the luma units that reconstruct it).  The luma blocks
              // that produce it are the same for every chroma block of the VPDU and all precede the VPDU's first CU in decoding order, so
              // they are looked up once per VPDU
              const size_t vp = (size_t) ( tu.y >> vpduLog2 ) * vpdusX + ( tu.x >> vpduLog2 );
              if( csProdRange[vp].first == 0xffffffffu ) lookUpCsProducers( vp );
              for( uint32_t q = csProdRange[vp].first; q < csProdRange[vp].first + csProdRange[vp].second; q++ )
              {
                const uint32_t key = csProdPool[q];          // (component 0)
                if( std::find( pool.begin() + IH.p0, pool.end(), key ) == pool.end() ) pool.push_back( key );
              }
              lastKey = 0xffffffffu;
            }
            if( isCclm )
            {
              // luma the prediction reads: the co-located block and the template rows / columns around it (luma coordinates)
              const int lx0 = x0 << 1, ly0 = y0 << 1;
              for( int yy = 0; yy < 2 * hh; yy += 4 ) for( int xx = ( cclmBLeft ? -4 : 0 ); xx < 2 * w; xx += 4 ) touch( 0, lx0 + xx, ly0 + yy );
              for( int xx = ( cclmBLeft ? -4 : 0 ); xx < 2 * cclmTop + 4; xx += 4 ) touch( 0, lx0 + xx, ly0 - 1 );
              for( int yy = 0; yy < 2 * cclmLeft + 4; yy += 4 ) touch( 0, lx0 - 1, ly0 + yy );
            }
            }     // (not an all-intra CTU)
            else if( !comp )
            {
              // all-intra CTU: the unit is the whole (component, CTU) in coding order; the last block of it this one reads a reference sample from
              // (same lines the kernel fills: corner, above incl. above-right, left incl. below-left, the previous ISP partition).  Luma only: the CTU
              // wavefront of an I picture advances with the luma units (33 blocks per CTU against a dozen chroma blocks, which stay serial)
              const int l2 = h.log2_ctu, m4 = ( 1 << ( l2 - 2 ) ) - 1, ctuX0 = ( cu.x >> l2 ) << l2, ctuY0 = ( cu.y >> l2 ) << l2;
              int last = -1;
              auto look = [&]( int xc, int yc )
              {
                const int lx = xc << cs, ly = yc << cs;
                if( lx < ctuX0 || ly < ctuY0 || lx >= ctuX0 + ( 1 << l2 ) || ly >= ctuY0 + ( 1 << l2 ) ) return;
                const uint16_t j = fastCell[comp][( ( ( ly >> 2 ) & m4 ) << ( l2 - 2 ) ) | ( ( lx >> 2 ) & m4 )];
                if( j != 0xffff ) last = std::max<int>( last, j );
              };
              // (nearly nine blocks in ten read the block directly before them - their left or above neighbour in z order: those places first, and no
              // further look once that block is found)
              const int local = (int) ( myId - fastFirst[comp] );
              if( ispL && ( x0 != rx0 || y0 != ry0 ) ) look( cu.isp_mode == 2 ? x0 - 1 : x0, cu.isp_mode == 2 ? y0 : y0 - 1 );
              if( last < local - 1 && it.nL ) look( rx0 - 1 - mrl, ry0 );
              if( last < local - 1 && it.nA ) look( rx0, ry0 - 1 - mrl );
              for( int k = it.nA * unit - unit; k > 0 && last < local - 1; k -= unit ) look( rx0 + k, ry0 - 1 - mrl );       // (above-right first: the latest blocks)
              for( int k = it.nL * unit - unit; k > 0 && last < local - 1; k -= unit ) look( rx0 - 1 - mrl, ry0 + k );
              if( last < local - 1 && it.nTL ) look( rx0 - 1 - mrl, ry0 - 1 - mrl );
              const uint32_t indep = (uint32_t) std::min( 63, std::max( 0, local - 1 - last ) );
              intra[comp].back().comp = (uint8_t) ( comp | ( indep << 2 ) );
              if( local < 0xffff )
              {
                const int cx0 = ( x0 << cs ) >> 2, cx1 = std::min( ( ( ( x0 + w ) << cs ) + 3 ) >> 2, w4 ), cy1 = std::min( ( ( ( y0 + hh ) << cs ) + 3 ) >> 2, h4 );
                // (later blocks only ever look at the cells along a block's right and bottom edge: their reference lines run there)
                const int cy0 = ( y0 << cs ) >> 2;
                for( int cx = cx0; cx < cx1; cx++ ) fastCell[comp][( ( ( cy1 - 1 ) & m4 ) << ( l2 - 2 ) ) | ( cx & m4 )] = (uint16_t) local;
                for( int cy = cy0; cy < cy1 - 1; cy++ ) fastCell[comp][( ( cy & m4 ) << ( l2 - 2 ) ) | ( ( cx1 - 1 ) & m4 )] = (uint16_t) local;
              }
            }
            // the cells this block reconstructs (the map is only ever read by the producer analysis of CTUs that are not all intra)
            if( !allIntraCus )
            {
              const int cx0 = ( x0 << cs ) >> 2, cx1 = std::min( ( ( ( x0 + w ) << cs ) + 3 ) >> 2, w4 ), cy1 = std::min( ( ( ( y0 + hh ) << cs ) + 3 ) >> 2, h4 );
              for( int cy = ( y0 << cs ) >> 2; cy < cy1; cy++ ) { uint32_t* row = &itemAtE[comp][cellIdx( cx0, cy )]; std::fill( row, row + std::max( 0, cx1 - cx0 ), ( epoch << 22 ) | myId ); }
            }
          }
          IH.pn = (uint32_t) pool.size() - IH.p0;
          itemH[comp].push_back( IH );
          bytes[K_INTRA] += (double) w * hh * ( hasResi ? 4 : 2 ) + sizeof( IntraItem );
          if( !comp ) bytesIntraLuma += (double) w * hh * ( hasResi ? 4 : 2 ) + sizeof( IntraItem );
        }
      }
    }
    if( cu.pred_mode == VVR_PRED_INTER )
    {
      const int nl = cu.mc_mode == VVR_MC_UNI ? 1 : 2;     // (SbTMVP: upper bound, sub-blocks may be uni-directional)
      const bool sbt = cu.mc_mode == VVR_MC_SBTMVP;
      const int ts = sbt ? 8 : 16;                       // SbTMVP: one item per 8x8 sub-block (ATMVP_SUB_BLOCK_SIZE)
      const bool dm = cu.mc_mode == VVR_MC_DMVR || cu.mc_mode == VVR_MC_DMVR_BDOF;
      const bool af = cu.mc_mode == VVR_MC_AFFINE;
      // reference picture resampling: a CU that reads a scaled reference picture goes to k_mc_rpr as a whole (tiles written here); the sub-blocks of an
      // SbTMVP CU are sorted tile by tile below
      bool rprCu = false;
      if( p->rpr && !sbt )
      {
        if( cu.mc_mode == VVR_MC_GEO ) for( int k = 0; k < 2; k++ ) rprCu |= p->rpr->ref[( cu.geo_dir_ref[k] >> 4 ) - 1][cu.geo_dir_ref[k] & 15].scaled != 0;
        else for( int l = 0; l < 2; l++ ) rprCu |= cu.ref_idx[l] >= 0 && p->rpr->ref[l][cu.ref_idx[l]].scaled;
      }
      if( lfpOnDevice && ( sbt || cu.mc_mode == VVR_MC_GEO || ( af && !( h.tool_flags & VVR_TOOL_AFFINE_MV_ON_DEVICE ) ) ) )
      {
        // the edge parameters are derived on the device: the motion of a CU whose motion varies inside it is not in its record
        const int cx0 = cu.x >> 2, cy0 = cu.y >> 2, cx1 = std::min( ( cu.x + cu.w + 3 ) >> 2, w4 ), cy1 = std::min( ( cu.y + cu.h + 3 ) >> 2, h4 );
        size_t at = lfSb.size();
        lfSb.resize( at + (size_t) ( cy1 - cy0 ) * ( cx1 - cx0 ) );           // (one growth per CU, not one per cell)
        for( int cy = cy0; cy < cy1; cy++ ) for( int cx = cx0; cx < cx1; cx++, at++ ) { const uint32_t cell = (uint32_t) ( cy * w4 + cx ); lfSb[at].cell = cell; lfSb[at].m = p->motion[cell]; }
      }
      std::vector<McItem>& list = rprCu ? mcRpr : dm ? mcDmvr : af ? mcAff : cu.mc_mode == VVR_MC_BDOF ? mcBdof : mc;
      const int nla = af ? ( ( cu.ref_idx[0] >= 0 && cu.ref_idx[1] >= 0 ) ? 2 : 1 ) : nl;
      if( !af && !sbt && !rprCu )
      {
        // plain, BDOF and DMVR tiles are a function of the CU record: counted here, written on the device (k_expand_mc)
        const int cls = dm ? 2 : cu.mc_mode == VVR_MC_BDOF ? 1 : 0;
        const uint32_t nt = (uint32_t) ( ( cu.w + 15 ) >> 4 ) * ( ( cu.h + 15 ) >> 4 );
        mcCus.push_back( McCuRef{ i, ( (uint32_t) cls << 30 ) | devTiles[cls] } );
        devTiles[cls] += nt;
        const double smp = (double) cu.w * cu.h * ( ncomp == 3 ? 1.5 : 1.0 );
        const double bts = smp * 2 * nla + smp * 2 + nt * ( sizeof( McItem ) + ( dm ? 8 : 0 ) );
        bytes[dm ? K_MC_DMVR : K_MC] += bts;
        if( cls == 1 ) bytesBdof += bts;
      }
      else
      {
      // one record per tile: what the tiles of a CU share is filled once
      McItem base; memset( &base, 0, sizeof( base ) );
      base.flags = ( sbt ? MC_ITEM_SUBBLOCK : 0 ) | ( af && rprCu ? MC_ITEM_AFFINE : 0 ); base.cu = i;
      const bool plain = !af && !dm;
      if( af && ( h.tool_flags & VVR_TOOL_AFFINE_MV_ON_DEVICE ) ) base.mv[0][0] = -1;      // the kernel spans the sub-block MVs from the control points itself
      if( plain )
      {
        // everything k_mc needs about the motion of the tile
        base.ref[0] = cu.ref_idx[0]; base.ref[1] = cu.ref_idx[1];
        for( int l = 0; l < 2; l++ ) { base.mv[l][0] = cu.mv[l][0][0]; base.mv[l][1] = cu.mv[l][0][1]; }
        base.clipX = cu.x; base.clipY = cu.y;
        base.bcw = cu.bcw_idx;
        base.flags |= ( cu.mc_mode == VVR_MC_UNI ? MC_ITEM_UNI : 0 ) | ( cu.imv == 3 ? MC_ITEM_HPEL : 0 ) | ( cu.mc_mode == VVR_MC_GEO ? MC_ITEM_GEO : 0 );
      }
      const size_t first = list.size();
      list.resize( first + (size_t) ( ( cu.w + ts - 1 ) / ts ) * ( ( cu.h + ts - 1 ) / ts ) );
      McItem* out = &list[first];
      for( int y = 0; y < cu.h; y += ts ) for( int x = 0; x < cu.w; x += ts )
      {
        McItem& it = *out++;
        it = base;
        it.x = (uint16_t) ( cu.x + x ); it.y = (uint16_t) ( cu.y + y ); it.w = (uint8_t) std::min( ts, cu.w - x ); it.h = (uint8_t) std::min( ts, cu.h - y );
        if( af && !( h.tool_flags & VVR_TOOL_AFFINE_MV_ON_DEVICE ) )
        {
          // the motion of the tile's 4x4 sub-blocks (MotionInfo of the affine CU, filled by PU::setAllAffineMv, UnitTools.cpp:3005): the kernel reads
          // them from a compact array, 4 x 4 entries per tile, so the motion field itself never crosses PCIe
          it.mv[0][0] = (int32_t) affMv.size();
          const size_t base = affMv.size();
          affMv.resize( base + 16 );
          for( int sy = 0; sy < it.h >> 2; sy++ ) memcpy( &affMv[base + 4 * sy], &p->motion[(size_t) ( ( it.y >> 2 ) + sy ) * w4 + ( it.x >> 2 )], sizeof( vvr_motion ) * ( it.w >> 2 ) );
        }
        else if( sbt )
        {
          // SbTMVP (xSubPuMC, InterPrediction.cpp:438): the motion of the 8x8 sub-block from the motion field, the identical-motion
          // shortcut (xCheckIdenticalMotion :404, not with weighted bi-prediction :408) decided per sub-block, clipped at its own position
          const vvr_motion& m = p->motion[(size_t) ( it.y >> 2 ) * w4 + ( it.x >> 2 )];
          for( int l = 0; l < 2; l++ ) { it.ref[l] = m.ref_idx[l]; it.mv[l][0] = m.mv[l][0]; it.mv[l][1] = m.mv[l][1]; }
          const bool two = it.ref[0] >= 0 && it.ref[1] >= 0;
          const bool uni = !two || ( h.ref_poc[0][it.ref[0]] == h.ref_poc[1][it.ref[1]] && it.mv[0][0] == it.mv[1][0] && it.mv[0][1] == it.mv[1][1] && !( wpOn && ( flagsOfCtu( ctuAt( it.x, it.y ) ) & VVR_TOOL_WP ) ) );
          it.clipX = it.x; it.clipY = it.y;
          it.flags = (uint16_t) ( ( it.flags & ~MC_ITEM_UNI ) | ( uni ? MC_ITEM_UNI : 0 ) );
        }
      }
      if( sbt && h.wrap_offset )
      {
        // reference wrap-around: wrapClipMv (Mv.cpp:112) depends on the position AND the width of the block that is predicted - for SbTMVP the pieces
        // xSubPuMC forms (InterPrediction.cpp:466-543): sub-blocks of equal motion (MotionInfo::operator==: the MV of an unused list does not count) joined
        // along the CU's longer side - not when the first reference picture of a list is scaled -, a joined run of more than 16 samples that is no multiple
        // of 16 cut into its multiple-of-16 part and the rest.  Every sub-block tile carries the area of its piece.
        const int nx = cu.w >> 3, ny = cu.h >> 3;
        const bool verMC = cu.h > cu.w;
        const int nFst = verMC ? nx : ny, nSec = verMC ? ny : nx;
        const bool scaled = p->rpr && ( p->rpr->ref[0][0].scaled || ( h.num_ref[1] > 0 && p->rpr->ref[1][0].scaled ) );
        McItem* sb = &list[first];
        auto at = [&]( int f, int s ) -> McItem& { return verMC ? sb[s * nx + f] : sb[f * nx + s]; };
        for( int f = 0; f < nFst; f++ )
          for( int s0 = 0; s0 < nSec; )
          {
            const McItem& a = at( f, s0 );
            int s1 = s0 + 1;
            for( ; s1 < nSec && !scaled; s1++ )
            {
              const McItem& b = at( f, s1 );
              if( b.ref[0] != a.ref[0] || b.ref[1] != a.ref[1] ) break;
              if( a.ref[0] >= 0 && ( b.mv[0][0] != a.mv[0][0] || b.mv[0][1] != a.mv[0][1] ) ) break;
              if( a.ref[1] >= 0 && ( b.mv[1][0] != a.mv[1][0] || b.mv[1][1] != a.mv[1][1] ) ) break;
            }
            const int len = 8 * ( s1 - s0 ), cut = ( len > 16 && ( len & 15 ) ) ? ( len & ~15 ) : len;
            for( int s = s0; s < s1; s++ )
            {
              McItem& t = at( f, s );
              if( p->rpr && ( ( t.ref[0] >= 0 && p->rpr->ref[0][t.ref[0]].scaled ) || ( t.ref[1] >= 0 && p->rpr->ref[1][t.ref[1]].scaled ) ) ) continue;      // (k_mc_rpr: the tile is its own block)
              const bool inFirst = 8 * ( s - s0 ) < cut;
              const int p0 = 8 * s0 + ( inFirst ? 0 : cut ), pl = inFirst ? cut : len - cut;
              if( verMC ) { t.clipX = t.x; t.clipW4 = 2; t.clipY = (uint16_t) ( cu.y + p0 ); }
              else        { t.clipY = t.y; t.clipX = (uint16_t) ( cu.x + p0 ); t.clipW4 = (uint8_t) ( pl >> 2 ); }
            }
            s0 = s1;
          }
      }
      if( p->rpr && sbt )
      {   // the sub-blocks that read a scaled picture: to k_mc_rpr
        size_t keep = first;
        for( size_t k = first; k < list.size(); k++ )
        {
          const McItem& t = list[k];
          if( ( t.ref[0] >= 0 && p->rpr->ref[0][t.ref[0]].scaled ) || ( t.ref[1] >= 0 && p->rpr->ref[1][t.ref[1]].scaled ) ) mcRpr.push_back( t ); else list[keep++] = t;
        }
        list.resize( keep );
      }
      {
        const double smp = (double) cu.w * cu.h * ( ncomp == 3 ? 1.5 : 1.0 ), nt = (double) ( list.size() - first );
        const double bts = smp * 2 * nla + smp * 2 + nt * ( sizeof( McItem ) + ( dm ? 8 : 0 ) ) + ( af ? cu.w * cu.h / 16.0 * sizeof( vvr_motion ) : 0 );
        bytes[dm ? K_MC_DMVR : af ? K_MC_AFFINE : K_MC] += bts;
        if( &list == &mcBdof ) bytesBdof += bts;
      }
      }     // (tiles written by the host)
      if( dm ) numDmvr = std::max<uint32_t>( numDmvr, cu.dmvr_off + ( ( cu.w + 15 ) / 16 ) * ( ( cu.h + 15 ) / 16 ) );
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
        if( ( bw < 2 || bh < 2 ) && !( cu.isp_mode && !it.comp && bw * bh >= 16 ) ) FAIL( VVR_ERR_PARAMETER, "1-D transform block outside an ISP CU" );
        it.pad = (uint8_t) ( ( ( it.comp && cu.isp_mode ) ? TB_P_CUGEOM : 0 ) | ( ( it.comp ? cu.bdpcm[1] : cu.bdpcm[0] ) ? TB_P_BDPCM : 0 )
                             | ( ( cu.lfnst_idx && ( cu.tree != VVR_TREE_JOINT || it.comp == 0 ) ) ? TB_P_LFNST : 0 ) );
        const int cls = std::max( bw, bh ) <= 16 ? 0 : std::max( bw, bh ) <= 32 ? 1 : 2;
        // LMCS chroma residual scaling of an inter block: the factor needs the reconstructed luma around the VPDU, which the intra stage
        // may still have to produce, so the block's residual is stored and added (scaled) by a residual-add item of the intra stage
        if( cscaleCtu( ctuOfCu ) && it.comp && it.mode == TB_ADD && bw * bh > 4 ) it.mode = TB_STORE;      // added (scaled) by the intra stage, see isCsInter above
        tb[cls].push_back( it );        // ADD (inter: onto the prediction) and STORE (intra / CIIP: into the residual planes) items share a launch
        const int bdp = it.comp ? cu.bdpcm[1] : cu.bdpcm[0];
        const double ncoef = bdp ? (double) bw * bh : (double) ( tu.max_scan_x[it.comp] + 1 ) * ( tu.max_scan_y[it.comp] + 1 );
        const double bts = ncoef * 2 + (double) bw * bh * 4 * ( it.ict ? 2 : 1 ) + sizeof( TbItem ) + sizeof( vvr_tu ) / 3.0;
        bytes[K_ITRANS] += bts; bytesTb[cls] += bts;
      }
    }
  }
  return VVR_OK;
}

// a unit lists at most VVR_INTRA_MAX_DEPS producers: longer lists are folded through empty join units
void PrepScratch::foldLongDepLists()
{
  for( size_t u = 0; u < units.size(); u++ )
    while( units[u].deps.size() > VVR_INTRA_MAX_DEPS )
    {
      UnitH j; j.comp = units[u].comp; j.ctu = units[u].ctu; j.i0 = j.i1 = j.iA = units[u].i0; j.bb.y0 = j.bb.y1 = 0; j.bb.c0 = j.bb.c1 = 1;
      j.deps.assign( units[u].deps.end() - VVR_INTRA_MAX_DEPS, units[u].deps.end() );
      units[u].deps.resize( units[u].deps.size() - VVR_INTRA_MAX_DEPS );
      units[u].deps.push_back( (uint32_t) units.size() );
      units.push_back( j );
    }
}

// rank = length of the longest dependency chain below a unit (the unit graph is acyclic: luma never reads chroma, residual-add units only
// read luma, other CTUs' units only earlier CTUs'); computed by relaxation in creation order until stable
void PrepScratch::rankUnits()
{
  for( auto& U : units ) U.rank = 0;
  bool changed = true;
  for( size_t pass = 0; changed && pass <= units.size(); pass++ )      // (acyclic: stable after at most one pass per level; creation order makes it 2-3)
  {
    changed = false;
    for( auto& U : units ) for( uint32_t d : U.deps ) if( units[d].rank + 1 > U.rank ) { U.rank = units[d].rank + 1; changed = true; }
  }
}

int PrepScratch::formUnits()
{
  // ---- form the units: blocks of one (component, CTU) that read from each other belong together (union-find); the residual-add items of
  // inter blocks (LMCS chroma scaling) of a (component, CTU) form a unit of their own that the kernel processes in parallel
  for( int k = 0; k < ncomp; k++ )
  {
    const size_t n = intra[k].size();
    if( !n ) continue;
    if( allIntraCus && intraChunk >= n )
    {
      // a picture whose CUs are all intra CUs: the blocks of a (component, CTU) are one unit and already lie together, in coding order - the units
      // are the runs of equal CTU (what the general way below arrives at through union-find, member lists and a sorted copy of the blocks)
      // (one pass over the blocks: the runs, their bounding boxes and the per-CTU counts - this is the serial tail behind an I picture's parts, on the path the
      // next GOP waits for)
      uint32_t* cnt = &ctuStartV[(size_t) k * ( numCtu + 1 )];
      std::fill( cnt, cnt + numCtu + 1, 0u );
      const ItemH* IH = itemH[k].data(); const IntraItem* IT = intra[k].data();
      for( size_t i = 0; i < n; )
      {
        UnitH u; u.comp = (uint32_t) k; u.ctu = IH[i].ctu; u.i0 = (uint32_t) i;
        for( ; i < n && IH[i].ctu == u.ctu; i++ )
        {
          const BBox& b = IH[i].bb;
          u.bb.y0 = std::min( u.bb.y0, b.y0 ); u.bb.y1 = std::max( u.bb.y1, b.y1 ); u.bb.c0 = std::min( u.bb.c0, b.c0 ); u.bb.c1 = std::max( u.bb.c1, b.c1 );
          if( k && ( IT[i].flags & IT_F_CSCALE ) ) u.hasCs = true;
        }
        u.i1 = (uint32_t) i; u.iA = u.i0;
        cnt[u.ctu + 1] += u.i1 - u.i0;
        units.push_back( std::move( u ) );
      }
      for( int a = 0; a < numCtu; a++ ) cnt[a + 1] += cnt[a];
      continue;
    }
    const std::vector<uint32_t>& pool = prodPool[k];
    parent.resize( n );
    for( size_t i = 0; i < n; i++ ) parent[i] = (uint32_t) i;
    auto find = [&]( uint32_t a ) { while( parent[a] != a ) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
    auto unite = [&]( uint32_t a, uint32_t b ) { a = find( a ); b = find( b ); if( a != b ) parent[std::max( a, b )] = std::min( a, b ); };     // root = first block
    int64_t bulk = -1; uint32_t bulkCtu = 0;
    for( size_t i = 0; i < n; i++ )
    {
      const bool ra = intra[k][i].mode == IT_MODE_RESI_ADD && k;
      if( ra ) { if( bulk >= 0 && bulkCtu == itemH[k][i].ctu ) unite( (uint32_t) bulk, (uint32_t) i ); else { bulk = (int64_t) i; bulkCtu = itemH[k][i].ctu; } continue; }
      const ItemH& ih = itemH[k][i];
      // an all-intra CTU: one unit per component (every block but the first reads from a block of its CTU that precedes it)
      if( fastCtu[ih.ctu] ) { if( i > 0 && itemH[k][i - 1].ctu == ih.ctu ) unite( (uint32_t) i, (uint32_t) i - 1 ); continue; }
      for( uint32_t q = ih.p0; q < ih.p0 + ih.pn; q++ )
      {
        const uint32_t key = pool[q], pk = key >> 28, pi = key & 0x0fffffff;
        if( (int) pk == k && itemH[k][pi].ctu == ih.ctu && !( intra[k][pi].mode == IT_MODE_RESI_ADD && k ) ) unite( (uint32_t) i, pi );
      }
    }
    // units in the order of their first block; blocks of a unit contiguous and in coding order
    unitOfRoot.assign( n, -1 );
    size_t numMembers = 0;
    for( size_t i = 0; i < n; i++ )
    {
      const uint32_t r = find( (uint32_t) i );
      if( unitOfRoot[r] < 0 ) { unitOfRoot[r] = (int32_t) numMembers; if( members.size() <= numMembers ) members.emplace_back(); members[numMembers].clear(); numMembers++; }
      members[unitOfRoot[r]].push_back( (uint32_t) i );
    }
    std::vector<IntraItem>& sorted = intraTmp[k]; sorted.clear(); sorted.reserve( n );
    std::vector<ItemH>& sortedH = itemHTmp; sortedH.clear(); sortedH.reserve( n );
    newIdx.resize( n );
    for( size_t mi = 0; mi < numMembers; mi++ )
    {
      const std::vector<uint32_t>& all = members[mi];
      // Developer option (VVR_INTRA_CHUNK in developer builds): a long cluster (an intra CTU: every block reads from the one before it) cut into
      // pieces of at most intraChunk blocks in coding order, each a unit of its own, so that the CTU to the right could start when the piece that
      // holds its left neighbours is done.  Measured on 4K intra pictures with mean CU sizes 32 / 20 / 16: 0 .. -9 % at best, the below-left
      // reference samples of the first blocks of a CTU reach far down the left CTU's last column.
      const bool cut = all.size() > intraChunk && !( k && intra[k][all[0]].mode == IT_MODE_RESI_ADD );
      for( size_t a = 0; a < all.size(); )
      {
      size_t b = all.size();
      if( cut )
      {
        b = std::min( all.size(), a + intraChunk );
        if( all.size() - b < intraChunk / 2 ) b = all.size();                   // (no tiny last piece)
        // the partitions of an ISP coding unit share the reference line fetched with the first one: they stay together
        while( b < all.size() && !k && ( intra[k][all[b]].flags & IT_F_ISP ) == IT_F_ISP && !( intra[k][all[b]].flags & IT_F_MIP ) && ( intra[k][all[b]].tu & 0xfff ) ) b++;
      }
      UnitH u; u.comp = (uint32_t) k; u.ctu = itemH[k][all[a]].ctu; u.i0 = (uint32_t) sorted.size();
      for( ; a < b; a++ )
      {
        const uint32_t i = all[a];
        newIdx[i] = (uint32_t) sorted.size();
        sorted.push_back( intra[k][i] ); sortedH.push_back( itemH[k][i] );
        const BBox& b = sortedH.back().bb;
        u.bb.y0 = std::min( u.bb.y0, b.y0 ); u.bb.y1 = std::max( u.bb.y1, b.y1 ); u.bb.c0 = std::min( u.bb.c0, b.c0 ); u.bb.c1 = std::max( u.bb.c1, b.c1 );
        if( k && ( intra[k][i].flags & IT_F_CSCALE ) ) u.hasCs = true;
      }
      u.i1 = (uint32_t) sorted.size();
      u.iA = ( k && sorted[u.i0].mode == IT_MODE_RESI_ADD ) ? u.i1 : u.i0;
      units.push_back( std::move( u ) );
      }
    }
    intra[k].swap( sorted ); itemH[k].swap( sortedH );
    // block index -> its new place, in every producer list that names a block of component k (chroma never is a producer for luma, and
    // components are processed in ascending order, so every reference to component k is fixed here)
    for( int k2 = k; k2 < ncomp; k2++ ) for( uint32_t& key : prodPool[k2] ) if( (int) ( key >> 28 ) == k ) key = ( (uint32_t) k << 28 ) | newIdx[key & 0x0fffffff];
    // the per-CTU offsets follow the new order (units, hence blocks, stay grouped by CTU)
    {
      uint32_t* cnt = &ctuStartV[(size_t) k * ( numCtu + 1 )];
      std::fill( cnt, cnt + numCtu + 1, 0u );
      for( auto& ih : itemH[k] ) cnt[ih.ctu + 1]++;
      for( int a = 0; a < numCtu; a++ ) cnt[a + 1] += cnt[a];
    }
  }
  // dependencies between units
  {
    // (a picture whose CUs are all intra CUs collected no producers per block - every CTU is a "fast" one, below: nothing to map, nothing to walk)
    if( !allIntraCus )
    {
    for( int k = 0; k < ncomp; k++ ) unitOfItem[k].assign( intra[k].size(), 0 );
    for( size_t u = 0; u < units.size(); u++ ) for( uint32_t i = units[u].i0; i < units[u].i1; i++ ) unitOfItem[units[u].comp][i] = (uint32_t) u;
    }
    for( size_t u = 0; !allIntraCus && u < units.size(); u++ )
    {
      UnitH& U = units[u];
      const std::vector<uint32_t>& pool = prodPool[U.comp];
      uint32_t last = 0xffffffffu;
      for( uint32_t i = U.i0; i < U.i1; i++ )
      {
        const ItemH& ih = itemH[U.comp][i];
        for( uint32_t q = ih.p0; q < ih.p0 + ih.pn; q++ )
        {
          const uint32_t d = unitOfItem[pool[q] >> 28][pool[q] & 0x0fffffff];
          if( d == u || d == last ) continue;
          last = d;
          if( std::find( U.deps.begin(), U.deps.end(), d ) == U.deps.end() ) U.deps.push_back( d );
        }
      }
    }
    // units of all-intra CTUs (no per-block producers were collected): everything such a unit can read outside itself lies in the CTUs left,
    // above-left, above and above-right of its own - reference lines incl. the above-right extension, multi-reference lines, the top-left sample -
    // as far as they belong to its slice and tile; a chroma unit also reads luma there and in its own CTU (CCLM templates, the neighbourhood of
    // the chroma-scaling factor).  It waits for every unit of these (component, CTU) pairs: a superset of what its blocks read, of CTUs that
    // precede it in the wavefront anyway.
    {
      std::vector<uint32_t>& uFirst = unitCount;                       // (reused scratch) first unit / number of units per (component, CTU); units of a pair are contiguous
      uFirst.assign( 2 * 3 * (size_t) numCtu, 0 );
      uint32_t* uNum = uFirst.data() + 3 * (size_t) numCtu;
      for( size_t u = units.size(); u-- > 0; ) { const size_t key = (size_t) units[u].comp * numCtu + units[u].ctu; uFirst[key] = (uint32_t) u; uNum[key]++; }
      for( size_t u = 0; u < units.size(); u++ )
      {
        UnitH& U = units[u];
        if( !fastCtu[U.ctu] ) continue;
        const int cx = (int) ( U.ctu % ctusX ), cy = (int) ( U.ctu / ctusX );
        auto addAll = [&]( uint32_t comp, uint32_t ctuIdx )
        {
          const size_t key = (size_t) comp * numCtu + ctuIdx;
          for( uint32_t q = uFirst[key]; q < uFirst[key] + uNum[key]; q++ ) if( q != u && std::find( U.deps.begin(), U.deps.end(), q ) == U.deps.end() ) U.deps.push_back( q );
        };
        if( U.comp ) addAll( 0, U.ctu );
        const int nb[4][2] = { { -1, 0 }, { -1, -1 }, { 0, -1 }, { 1, -1 } };
        for( int n = 0; n < 4; n++ )
        {
          const int nx = cx + nb[n][0], ny = cy + nb[n][1];
          if( nx < 0 || ny < 0 || nx >= ctusX ) continue;
          const uint32_t nc = (uint32_t) ( ny * ctusX + nx );
          if( !sameSliceAndTile( nc, U.ctu ) ) continue;
          addAll( U.comp, nc );
          if( U.comp ) addAll( 0, nc );
        }
      }
    }
    foldLongDepLists();
    rankUnits();
  }
  return VVR_OK;
}

int PrepScratch::groupUnits()
{
  // ---- group the clusters of one (component, CTU) that sit at the same depth of the dependency graph into one unit: they cannot depend
  // on each other, a workgroup start costs more than a few small blocks, and waiting for the union of their producers delays nothing
  // that matters (all of them are less deep).  Residual-add units keep their own (HBM to HBM) workgroup.
  if( units.empty() ) return VVR_OK;
  // a picture whose CUs are all intra CUs: one unit per (component, CTU) and the join units of long dependency lists - nothing shares a
  // (component, CTU, depth), every group would be a unit by itself, in the order they are in
  if( allIntraCus ) return VVR_OK;
  target.assign( units.size(), -1 );            // original unit -> group
  size_t numGroups = 0;                         // groups: original units in creation order
  auto newGroup = [&]() { if( groups.size() <= numGroups ) groups.emplace_back(); groups[numGroups].clear(); return numGroups++; };
  {
    std::vector<std::pair<uint64_t, uint32_t>> keyed; keyed.reserve( units.size() );
    for( size_t u = 0; u < units.size(); u++ )
    {
      const bool own = units[u].iA == units[u].i1;             // residual-add unit (or empty): not grouped
      keyed.emplace_back( own ? ( ( (uint64_t) 1 << 63 ) | u ) : ( ( (uint64_t) units[u].comp << 56 ) | ( (uint64_t) units[u].ctu << 24 ) | (uint64_t) std::min( units[u].rank, 0xffffff ) ), (uint32_t) u );
    }
    std::stable_sort( keyed.begin(), keyed.end(), []( const std::pair<uint64_t, uint32_t>& x, const std::pair<uint64_t, uint32_t>& y ) { return x.first < y.first; } );
    const uint32_t groupMax = 12;               // blocks per grouped unit: a serial workgroup should stay short
    for( size_t i = 0; i < keyed.size(); )
    {
      size_t j = i; const size_t g = newGroup();
      uint32_t blocks = 0;
      while( j < keyed.size() && keyed[j].first == keyed[i].first )
      {
        const uint32_t nb = units[keyed[j].second].i1 - units[keyed[j].second].i0;
        if( blocks && blocks + nb > groupMax ) break;                   // start another one
        blocks += nb;
        groups[g].push_back( keyed[j].second ); target[keyed[j].second] = (int32_t) g; j++;
      }
      i = j;
    }
  }
  // groups in the order of their first original unit (keeps the blocks grouped by CTU)
  std::vector<uint32_t> orderM( numGroups );
  for( size_t m = 0; m < numGroups; m++ ) orderM[m] = (uint32_t) m;
  std::stable_sort( orderM.begin(), orderM.end(), [&]( uint32_t x, uint32_t y ) { return groups[x][0] < groups[y][0]; } );
  for( int k = 0; k < 3; k++ ) intraTmp[k].clear();
  std::vector<UnitH>& merged = unitsTmp; merged.clear();
  std::vector<uint32_t>& newIndexOfGroup = newIdx; newIndexOfGroup.assign( numGroups, 0 );
  for( uint32_t m : orderM )
  {
    const UnitH& f = units[groups[m][0]];
    UnitH U; U.comp = f.comp; U.ctu = f.ctu; U.i0 = (uint32_t) intraTmp[f.comp].size();
    for( uint32_t u : groups[m] )
    {
      const UnitH& o = units[u];
      intraTmp[o.comp].insert( intraTmp[o.comp].end(), intra[o.comp].begin() + o.i0, intra[o.comp].begin() + o.i1 );
      U.bb.y0 = std::min( U.bb.y0, o.bb.y0 ); U.bb.y1 = std::max( U.bb.y1, o.bb.y1 ); U.bb.c0 = std::min( U.bb.c0, o.bb.c0 ); U.bb.c1 = std::max( U.bb.c1, o.bb.c1 );
      U.hasCs = U.hasCs || o.hasCs;
    }
    U.i1 = (uint32_t) intraTmp[f.comp].size();
    U.iA = f.iA == f.i1 ? U.i1 : U.i0;
    newIndexOfGroup[m] = (uint32_t) merged.size();
    merged.push_back( std::move( U ) );
  }
  // ---- which of the blocks directly before it in its unit a block does NOT read from (IntraItem::comp bits 2..7, in blocks here; emitUnitTable
  // turns it into items): the kernel predicts a unit's blocks with several wavefronts and starts a block when all blocks up to the last one
  // it reads from are done.  The clusters that share a unit cannot depend on each other, and inside a cluster a block mostly reads from the
  // one before it - but not always (the first block of the lower half of a split reads from the upper half's first blocks only).
  // Blocks of all-intra CTUs carry theirs already (buildWorkLists: the CTU-local cell map; no producer lists were collected for them).
  {
    for( int k = 0; k < ncomp; k++ ) posAfterGrouping[k].assign( intra[k].size(), 0xffffffffu );
    for( int k = 0; k < 3; k++ ) groupFill[k] = 0;
    for( uint32_t m : orderM ) for( uint32_t u : groups[m] )
    {
      const UnitH& o = units[u];
      for( uint32_t i = o.i0; i < o.i1; i++ ) posAfterGrouping[o.comp][i] = groupFill[o.comp]++;
    }
    for( uint32_t m : orderM )
    {
      const UnitH& U = merged[newIndexOfGroup[m]];
      if( U.iA == U.i1 || fastCtu[U.ctu] ) continue;
      const uint32_t c = U.comp;
      for( uint32_t u : groups[m] )
      {
        const UnitH& o = units[u];
        for( uint32_t i = o.i0; i < o.i1; i++ )
        {
          const uint32_t local = posAfterGrouping[c][i] - U.i0;
          int64_t last = -1;                       // the last block of the unit this one reads from
          const ItemH& ih = itemH[c][i];
          for( uint32_t q = ih.p0; q < ih.p0 + ih.pn; q++ )
          {
            const uint32_t key = prodPool[c][q], pk = key >> 28, pi = key & 0x0fffffffu;
            if( pk != c || target[unitOfItem[pk][pi]] != (int32_t) m ) continue;
            last = std::max<int64_t>( last, (int64_t) posAfterGrouping[c][pi] - (int64_t) U.i0 );
          }
          if( last >= (int64_t) local ) last = (int64_t) local - 1;      // (cannot happen: producers precede their readers in coding order)
          const uint32_t indep = std::min<uint32_t>( 63, (uint32_t) ( (int64_t) local - 1 - last ) );
          IntraItem& dst = intraTmp[c][posAfterGrouping[c][i]];
          dst.comp = (uint8_t) ( ( dst.comp & 3 ) | ( indep << 2 ) );
        }
      }
    }
  }
  for( size_t m = 0; m < numGroups; m++ )
  {
    UnitH& U = merged[newIndexOfGroup[m]];
    for( uint32_t u : groups[m] ) for( uint32_t d : units[u].deps )
    {
      const uint32_t nd = newIndexOfGroup[target[d]];
      if( nd != newIndexOfGroup[m] && std::find( U.deps.begin(), U.deps.end(), nd ) == U.deps.end() ) U.deps.push_back( nd );
    }
  }
  for( int k = 0; k < ncomp; k++ ) intra[k].swap( intraTmp[k] );
  units.swap( merged );
  foldLongDepLists();
  rankUnits();
  return VVR_OK;
}

int PrepScratch::emitUnitTable( std::string& err )
{
  // one item array for the three components; active (component, CTU) pairs in raster order.  A block of more than IT_SPLIT_SAMPLES samples
  // (ordinary prediction modes and CIIP, luma and chroma) becomes 2, 4 or 8 items, one band of rows each: the kernel predicts an item
  // with one wavefront, the bands of a block with several at once (they read the same reference samples and write disjoint rows, so
  // band p is independent of the p items before it) - a band of 256 samples is one round of four samples per lane
  // (two passes: where every block's items start, then the items - written in place, no growing vector on the serial tail of an I picture's host stage)
  auto bandsLog2 = [&]( const IntraItem& src, int k )
  {
    const int samples = 1 << ( src.lw + src.lh );
    const bool split = samples > IT_SPLIT_SAMPLES && src.mode <= 66 && ( k || ( !( src.flags & IT_F_MIP ) && ( src.flags & IT_F_ISP ) != IT_F_ISP ) );
    int lp = 0;
    if( split ) while( lp < IT_MAX_LPARTS && ( samples >> lp ) > IT_SPLIT_SAMPLES ) lp++;
    return lp;
  };
  {
    uint32_t at = (uint32_t) intraAll.size();
    for( int k = 0; k < 3; k++ )
    {
      const size_t n = intra[k].size();
      itemMap[k].resize( n + 1 );
      uint32_t* im = itemMap[k].data(); const IntraItem* src = intra[k].data();
      for( size_t bi = 0; bi < n; bi++ ) { im[bi] = at; at += 1u << bandsLog2( src[bi], k ); }
      im[n] = at;
    }
    intraAll.resize( at );
  }
  for( int k = 0; k < 3; k++ )
  {
    const size_t n = intra[k].size();
    const uint32_t* im = itemMap[k].data(); const IntraItem* srcs = intra[k].data(); IntraItem* out = intraAll.data();
    for( size_t bi = 0; bi < n; bi++ )
    {
      const IntraItem& src = srcs[bi];
      // the blocks before this one that it does not read from (groupUnits), counted in items
      const uint32_t indepBlocks = std::min<uint32_t>( src.comp >> 2, (uint32_t) bi );
      const uint32_t indepItems = im[bi] - im[bi - indepBlocks];
      const int lp = ilog2i( (int) ( im[bi + 1] - im[bi] ) );
      for( int part = 0; part < ( 1 << lp ); part++ )
      {
        IntraItem it = src;
        it.nTL = (uint8_t) ( ( src.nTL & 1 ) | ( part << 1 ) | ( lp << 4 ) );
        it.comp = (uint8_t) ( k | ( std::min<uint32_t>( 63, indepItems + part ) << 2 ) );
        out[im[bi] + part] = it;
      }
    }
  }
  // device unit table: units that wait for nothing first (they can never block a resident workgroup slot), then the others by depth of the
  // dependency graph and along the CTU wavefront; a unit only ever waits for units that hold a lower ticket
  {
    perm.clear(); inv.assign( units.size(), 0 );
    for( size_t t = 0; t < units.size(); t++ ) if( units[t].deps.empty() ) perm.push_back( (uint32_t) t );
    {
      // dependent units by depth first (every producer is less deep, hence holds a lower ticket; units of one depth start together, so few of
      // them find a producer that has not even started), then in WAVEFRONT order (key = ctuX + 2 * ctuY): the workgroups that are resident at
      // any time are the ones on or near the current front.  Measured 7 % faster on B pictures than wavefront-major order, the same on intra
      // pictures where depth and wavefront coincide.
      const size_t first = perm.size();
      for( size_t t = 0; t < units.size(); t++ ) if( !units[t].deps.empty() ) perm.push_back( (uint32_t) t );
      // (the key of a unit once, not two divisions per comparison: the sort was the largest single piece of an I picture's serial tail)
      std::vector<uint64_t>& key = sortKey; key.resize( units.size() );
      for( size_t t = 0; t < units.size(); t++ ) key[t] = ( (uint64_t) (uint32_t) units[t].rank << 32 ) | (uint32_t) ( (int) ( units[t].ctu % ctusX ) + 2 * (int) ( units[t].ctu / ctusX ) );
      std::stable_sort( perm.begin() + first, perm.end(), [&]( uint32_t a, uint32_t b ) { return key[a] < key[b]; } );
    }
    // with residual-add blocks in the picture the stage runs in two launches, luma units then chroma units (k_resi_add between them): the luma
    // units take the first tickets; the order inside both parts stays (luma units never wait for chroma units)
    numLumaUnits = 0;
    const bool twoLaunches = !resiAdd.empty();
    if( twoLaunches ) numLumaUnits = (int) ( std::stable_partition( perm.begin(), perm.end(), [&]( uint32_t u ) { return units[u].comp == 0; } ) - perm.begin() );
    for( size_t t = 0; t < perm.size(); t++ ) inv[perm[t]] = (uint32_t) t;
    for( auto& u : units ) for( uint32_t d : u.deps ) units[d].waited = true;
    {
      // Workgroups the stage is launched with: about twice the AVERAGE parallelism of the dependency graph = total work / critical path, work
      // counted in blocks.  More would only add workgroups that spin on their producers while holding 50 KB of LDS each (an intra picture: 1530
      // units, a 62-CTU-deep wavefront, about 15 of them busy at any time), and those are taken away from the other pictures in flight; a
      // picture of isolated intra blocks (B picture: chains a few units deep) gets hundreds.
      std::vector<uint32_t>& path = unitCount;            // (reused below) longest chain ending in the unit, in blocks; tickets are a topological order
      path.assign( units.size(), 0 );
      uint64_t mult = h.slice_type == 2 ? 4 : 2;          // (an intra picture, workgroups of eight wavefronts: 4803 against 5001 us at 4K with 4 instead of 2; inter pictures: no difference from 2 to 8)
#if defined( VVR_WATCHDOG ) || defined( VVR_DEV_ENV )
      if( const char* e = getenv( "VVR_INTRA_WG_MULT" ) ) mult = (uint64_t) atoi( e );      // developer build: sweep
#endif
      // (two launches: each part on its own - the luma units are finished when the chroma units start)
      auto workgroupsOf = [&]( size_t t0, size_t t1 ) -> int
      {
        if( t1 <= t0 ) return 0;
        uint64_t work = 0; uint32_t critical = 1;
        for( size_t t = t0; t < t1; t++ )
        {
          const UnitH& u = units[perm[t]];
          const uint32_t cost = 1 + ( u.i1 - u.i0 );
          uint32_t before = 0;
          for( uint32_t d : u.deps ) if( inv[d] >= t0 ) before = std::max( before, path[inv[d]] );
          path[t] = before + cost;
          work += cost; critical = std::max( critical, path[t] );
        }
        return (int) std::min<uint64_t>( t1 - t0, std::max<uint64_t>( 32, mult * ( ( work + critical - 1 ) / critical ) ) );
      };
      intraWorkgroups = workgroupsOf( 0, twoLaunches ? (size_t) numLumaUnits : perm.size() );
      intraWorkgroupsChroma = twoLaunches ? workgroupsOf( (size_t) numLumaUnits, perm.size() ) : 0;
    }
    unitCount.assign( 3 * (size_t) numCtu, 0 );
    for( auto& u : units ) unitCount[(size_t) u.comp * numCtu + u.ctu]++;
    unitsDev.resize( units.size() );
    // a picture whose units are all whole CTUs of intra CUs (an I picture without IBC): its CTU wavefront can be resolved block by block (k_intra<.., FINE>)
    intraFine = allIntraCus && !twoLaunches && !( h.tool_flags & VVR_TOOL_IBC );
    for( size_t t = 0; t < perm.size(); t++ )
    {
      const UnitH& u = units[perm[t]];
      IntraUnit& d = unitsDev[t]; memset( &d, 0, sizeof( d ) );
      // bit 31: the unit is the whole (component, CTU) and every sample of the CTU is intra, so the kernel only stages the reference
      // border and writes the CTU back with 16-byte stores
      const bool all = unitCount[(size_t) u.comp * numCtu + u.ctu] == 1 && fastCtu[u.ctu];      // (fastCtu: every CU of the CTU is an intra CU)
      d.ent = ( u.comp << 24 ) | u.ctu | ( u.hasCs ? 0x20000000u : 0 ) | ( u.waited ? 0x40000000u : 0 ) | ( all ? 0x80000000u : 0 );
      if( !all && u.i1 > u.i0 ) intraFine = false;
      d.i0 = itemMap[u.comp][u.i0]; d.i1 = itemMap[u.comp][u.i1]; d.iA = itemMap[u.comp][u.iA];
      d.bbox = (uint32_t) u.bb.y0 | ( (uint32_t) u.bb.y1 << 8 ) | ( (uint32_t) u.bb.c0 << 16 ) | ( (uint32_t) u.bb.c1 << 24 );
      d.ndeps = (uint32_t) std::min<size_t>( u.deps.size(), VVR_INTRA_MAX_DEPS );
      if( u.deps.size() > VVR_INTRA_MAX_DEPS ) FAIL( VVR_ERR_UNSPECIFIED, "internal: intra unit with too many dependencies" );
      for( uint32_t k = 0; k < d.ndeps; k++ ) d.deps[k] = inv[u.deps[k]];
    }
  }
  return VVR_OK;
}

// (leaf) the item list of k_intra_leaf: luma blocks, the chroma-scaling factors of the VPDUs that need one, Cb blocks, Cr blocks - each list in decoding order; a
// block of more than IT_SPLIT_SAMPLES samples (ordinary prediction modes and CIIP) as 2, 4 or 8 bands of rows, like the other path's
int PrepScratch::emitLeafItems( std::string& err )
{
  (void) err;
  std::vector<uint16_t>& levAll = levAllV; levAll.clear();
  for( int k = 0; k < 3; k++ )
  {
    for( const IntraItem& src : intra[k] )
    {
      const int samples = 1 << ( src.lw + src.lh );
      const bool split = samples > IT_SPLIT_SAMPLES && src.mode <= 66 && ( k || ( !( src.flags & IT_F_MIP ) && ( src.flags & IT_F_ISP ) != IT_F_ISP ) );
      int lp = 0;
      if( split ) while( lp < IT_MAX_LPARTS && ( samples >> lp ) > IT_SPLIT_SAMPLES ) lp++;
      for( int part = 0; part < ( 1 << lp ); part++ )
      {
        IntraItem it = src;
        it.nTL = (uint8_t) ( ( src.nTL & 1 ) | ( part << 1 ) | ( lp << 4 ) );
        it.comp = (uint8_t) k;
        intraAll.push_back( it );
        if( leafSort ) levAll.push_back( lev[k][&src - intra[k].data()] );
      }
    }
    if( k == 0 && cscale )
    {
      // the residual-add blocks grouped by VPDU (decoding order inside a VPDU), one IT_MODE_CSFAC item per VPDU whose factor somebody needs: x | y << 16 = its
      // first block in the list, lw | lh << 8 = how many
      std::vector<uint32_t>& first = unitCount; first.assign( csNeeded.size() + 1, 0 );
      for( const IntraItem& r : resiAdd ) first[r.tu + 1]++;
      for( size_t vp = 0; vp < csNeeded.size(); vp++ ) first[vp + 1] += first[vp];
      std::vector<IntraItem>& sorted = intraTmp[0]; sorted.resize( resiAdd.size() );
      { std::vector<uint32_t>& fill = perm; fill.assign( first.begin(), first.end() - 1 ); for( const IntraItem& r : resiAdd ) sorted[fill[r.tu]++] = r; }
      resiAdd.swap( sorted );
      for( size_t vp = 0; vp < csNeeded.size(); vp++ ) if( csNeeded[vp] )
      {
        const uint32_t f = first[vp], n = first[vp + 1] - first[vp];
        IntraItem it; memset( &it, 0, sizeof( it ) );
        it.mode = IT_MODE_CSFAC; it.tu = (uint32_t) vp; it.comp = 1;
        it.x = (uint16_t) ( f & 0xffff ); it.y = (uint16_t) ( f >> 16 ); it.lw = (uint8_t) ( n & 0xff ); it.lh = (uint8_t) ( n >> 8 );
        intraAll.push_back( it );
        if( leafSort ) levAll.push_back( csLevelOf( vp ) );
      }
    }
  }
  if( leafSort && !intraAll.empty() )
  {
    // by level, decoding order within a level (counting sort; the bands of a block and the partitions of an ISP coding unit stay together: same level, neighbours)
    uint32_t top = 0;
    for( uint16_t l : levAll ) top = std::max<uint32_t>( top, l );
    std::vector<uint32_t>& at = unitCount; at.assign( (size_t) top + 2, 0 );
    for( uint16_t l : levAll ) at[(size_t) l + 1]++;
    for( uint32_t l = 0; l <= top; l++ ) at[l + 1] += at[l];
    std::vector<IntraItem>& sorted = intraTmp[1]; sorted.resize( intraAll.size() );
    for( size_t i = 0; i < intraAll.size(); i++ ) sorted[at[levAll[i]]++] = intraAll[i];
    intraAll.swap( sorted );
  }
  numLumaUnits = 0; intraWorkgroups = intraWorkgroupsChroma = 0;
  return VVR_OK;
}

void PrepScratch::layout( PinnedRanges* pinned )
{
  const double samples = (double) h.width * h.height * ( ncomp == 3 ? 1.5 : 1.0 );
  bytes[K_DEBLOCK_V] = bytes[K_DEBLOCK_H] = samples * 4 + (double) w4 * h4 * sizeof( vvr_lfp );
  bytes[K_SAO] = samples * 4; bytes[K_ALF] = samples * 4; bytes[K_COPY] = samples * 4;
  // LMCS, per launch: the inverse pass reads and writes every luma sample; the forward pass those of the inter CUs (upper bound: all)
  if( lmcs ) bytes[K_LMCS] = (double) h.width * h.height * 4;
  parts.clear(); total = 0;
  auto add = [&]( const void* src, size_t n ) { Part q{ src, n, total, false }; parts.push_back( q ); total += alignUp( std::max<size_t>( n, 16 ), 256 ); return (int) parts.size() - 1; };
  // the large arrays of the description come first: those that lie in pinned memory of the context (vvr_host_alloc) are copied to HBM from
  // where they are, everything behind them is staged (one contiguous range of the image)
  auto addCaller = [&]( const void* src, size_t n ) { const int i = add( src, n ); parts[i].direct = pinned && n >= 65536 && (size_t) i == numDirect && pinned->contains( src, n ); if( parts[i].direct ) numDirect++; return i; };
  numDirect = 0;
  iL0 = iL1 = -1;
  if( !( h.tool_flags & VVR_TOOL_LFP_ON_DEVICE ) )
  {
    iL0 = addCaller( p->lfp[0], sizeof( vvr_lfp ) * (size_t) w4 * h4 );
    iL1 = addCaller( p->lfp[1], sizeof( vvr_lfp ) * (size_t) w4 * h4 );
  }
  iCu = addCaller( p->cu, sizeof( vvr_cu ) * p->num_cu );
  iCoef = addCaller( p->coef, sizeof( int16_t ) * (size_t) p->num_coef );
  iTu = addCaller( p->tu, sizeof( vvr_tu ) * p->num_tu );
  iAffMv = add( affMv.data(), sizeof( vvr_motion ) * affMv.size() );
  iSao = p->sao ? add( p->sao, sizeof( vvr_sao_ctu ) * numCtu ) : -1;
  iAlf = p->alf ? add( p->alf, sizeof( vvr_alf_ctu ) * numCtu ) : -1;
  iAlfP = p->alf_params ? add( p->alf_params, sizeof( vvr_alf_params ) * std::max<uint32_t>( 1, p->num_alf_sets ) ) : -1;
  iSlices = ( p->slices && p->num_slices ) ? add( p->slices, sizeof( vvr_slice_header ) * p->num_slices ) : -1;
  // (LMCS: the forward mapping of inter predictions happens where the motion-compensation kernels store them: no per-cell map of inter CUs any more)
  iLmcs = lmcs ? add( p->lmcs, sizeof( vvr_lmcs_params ) ) : -1;
  iSl = ( h.tool_flags & VVR_TOOL_SCALING_LIST ) ? add( p->scaling, sizeof( vvr_scaling_list ) ) : -1;
  iCtuSlice = p->ctu_slice ? add( p->ctu_slice, sizeof( uint16_t ) * numCtu ) : -1;
  iSubpics = iCtuSubpic = -1;
  if( p->subpics && p->num_subpics > 1 )
  {
    ctuSubpicV.assign( (size_t) numCtu, 0 );
    for( uint32_t k = 0; k < p->num_subpics; k++ )
    {
      const vvr_subpic& sp = p->subpics[k];
      for( int y = sp.y0 >> h.log2_ctu; y <= sp.y1 >> h.log2_ctu; y++ ) for( int x = sp.x0 >> h.log2_ctu; x <= sp.x1 >> h.log2_ctu; x++ ) ctuSubpicV[(size_t) y * ctusX + x] = (uint16_t) k;
    }
    iSubpics = add( p->subpics, sizeof( vvr_subpic ) * p->num_subpics );
    iCtuSubpic = add( ctuSubpicV.data(), sizeof( uint16_t ) * ctuSubpicV.size() );
  }
  iCtuTile = p->ctu_tile ? add( p->ctu_tile, sizeof( uint16_t ) * numCtu ) : -1;
  iWp = wpOn ? add( p->wp, sizeof( vvr_wp_params ) * std::max<uint32_t>( 1, p->num_wp_sets ) ) : -1;
  iCsVpdu = cscale ? add( csVpduV.data(), sizeof( uint32_t ) * csVpduV.size() ) : -1;
  iMc = add( mc.data(), sizeof( McItem ) * mc.size() );
  iMcB = add( mcBdof.data(), sizeof( McItem ) * mcBdof.size() );
  iMcD = add( mcDmvr.data(), sizeof( McItem ) * mcDmvr.size() );
  iMcA = add( mcAff.data(), sizeof( McItem ) * mcAff.size() );
  iMcR = add( mcRpr.data(), sizeof( McItem ) * mcRpr.size() );
  iRpr = p->rpr && p->hdr.slice_type != 2 ? add( p->rpr, sizeof( vvr_rpr_params ) ) : -1;
  for( int k = 0; k < 3; k++ ) iTb[k] = add( tb[k].data(), sizeof( TbItem ) * tb[k].size() );
  iMcCus = add( mcCus.data(), sizeof( McCuRef ) * mcCus.size() );
  iIntra = add( intraAll.data(), sizeof( IntraItem ) * intraAll.size() );
  iResi = add( resiAdd.data(), sizeof( IntraItem ) * resiAdd.size() );
  iUnits = add( unitsDev.data(), sizeof( IntraUnit ) * unitsDev.size() );
  iLfSb = lfpOnDevice ? add( lfSb.data(), sizeof( LfSbCell ) * lfSb.size() ) : -1;
  // behind everything that is uploaded: room for what the device writes itself (k_expand_mc, k_lf_init)
  stagedEndOff = total;
  for( int k = 0; k < 3; k++ ) iMcDev[k] = add( nullptr, sizeof( McItem ) * devTiles[k] );
  iLfTu = iLfTuC = iLfMotion = -1;
  if( lfpOnDevice )
  {
    const size_t cells = (size_t) w4 * h4;
    iL0 = add( nullptr, sizeof( vvr_lfp ) * cells ); iL1 = add( nullptr, sizeof( vvr_lfp ) * cells );
    iLfTu = add( nullptr, sizeof( LfCell ) * cells ); iLfTuC = ncomp == 3 ? add( nullptr, sizeof( LfCell ) * cells ) : -1;
    iLfMotion = add( nullptr, ( sizeof( LfMv ) + sizeof( uint32_t ) ) * cells );
    // the per-cell records written and read (chroma tree: as far as there is one - not counted), the motion of inter cells written (read at edges the motion decides:
    // not counted), the two tables written, the CU / TU records read once
    bytes[K_LF_INIT] = (double) cells * ( 2 * sizeof( LfCell ) + ( h.slice_type != 2 ? sizeof( vvr_motion ) : 0 ) + 2 * sizeof( vvr_lfp ) ) + (double) sizeof( vvr_cu ) * p->num_cu + (double) sizeof( vvr_tu ) * p->num_tu + (double) lfSb.size() * sizeof( LfSbCell );
  }
}

// The work lists of a picture whose CUs are all intra CUs, built in parts (bands of CTU rows) by several threads: such CTUs are analysed without looking
// at any other CTU (mapCtu, the fast path of buildWorkLists), so the parts are independent; every part is built in the scratch of the thread that
// runs it and appended to this one's lists in band order.  The records of a part are checked by the thread that builds it.
int PrepScratch::buildInParts( const vvr_config& cfg, HostHelpers& helpers, bool validate, std::string& err )
{
  (void) cfg;
  const int n = std::min( helpers.width(), ctusY );
  struct Shared { std::mutex mu; std::condition_variable cv; int turn = 0; int rc = VVR_OK; std::string err; uint64_t area[2] = { 0, 0 }; } sh;
  const vvr_picture* pic = p;
  PrepScratch* owner = this;
  const bool analysed = !allIntraCus && !leaf;   // blocks name their producers: what a band reads of the band above is resolved when the bands are joined
  if( !allIntraCus ) g_bandPictures++;
  if( analysed ) for( int k = 0; k < ncomp; k++ ) edgeRow[k].assign( (size_t) w4, -1 );
  helpers.run( n, *this, [&]( int part, PrepScratch& R )
  {
    const int row0 = (int) ( (int64_t) owner->ctusY * part / n ), row1 = (int) ( (int64_t) owner->ctusY * ( part + 1 ) / n );
    const uint32_t cu0 = pic->ctu_first_cu[(size_t) row0 * owner->ctusX], cu1 = pic->ctu_first_cu[(size_t) row1 * owner->ctusX];
    int rc = VVR_OK; std::string e; uint64_t area[2] = { 0, 0 };
    if( cu0 > cu1 || cu1 > pic->num_cu ) { rc = VVR_ERR_PARAMETER; e = "ctu_first_cu is not ascending"; }
    if( rc == VVR_OK && validate ) rc = validate_records_range( pic, cu0, cu1, area, e );
    if( rc == VVR_OK )
    {
      if( &R != owner ) { R.begin( pic ); rc = R.beginMaps( owner ); }
      R.partCtu0 = (uint32_t) row0 * owner->ctusX; R.partCtu1 = (uint32_t) row1 * owner->ctusX;
      R.partTopY = analysed ? row0 << owner->h.log2_ctu : 0; R.partBad = false; R.pending.clear();
      if( rc == VVR_OK ) rc = R.buildWorkLists( e, cu0, cu1 );
      if( rc == VVR_OK && R.partBad ) { rc = VVR_ERR_UNSUPPORTED; e = "a block reads further up than the CTU row above it"; }
      R.partCtu0 = 0; R.partCtu1 = 0xffffffffu; R.partTopY = 0;
    }
    // append in band order (the owner's own part is the first and already in place)
    std::unique_lock<std::mutex> lk( sh.mu );
    sh.cv.wait( lk, [&]{ return sh.turn == part; } );
    if( rc != VVR_OK && sh.rc == VVR_OK ) { sh.rc = rc; sh.err = e; }
    if( rc == VVR_OK && sh.rc == VVR_OK )
    {
      sh.area[0] += area[0]; sh.area[1] += area[1];
      PrepScratch& O = *owner;
      uint32_t off[3] = { 0, 0, 0 };
      if( &R != owner )
      {
        for( int k = 0; k < O.ncomp; k++ ) off[k] = (uint32_t) O.intra[k].size();
        for( int k = 0; k < O.ncomp; k++ )
        {
          O.intra[k].insert( O.intra[k].end(), R.intra[k].begin(), R.intra[k].end() );
          if( O.leaf ) continue;                                                                               // (a list of blocks in decoding order is all there is)
          if( !analysed ) O.itemH[k].insert( O.itemH[k].end(), R.itemH[k].begin(), R.itemH[k].end() );        // (no producer lists in all-intra CTUs: p0 / pn stay 0)
          else
          {
            // the producer lists in picture-wide numbering; pending producers from the edge row the band above left behind; no producer twice
            std::vector<uint32_t>& pool = O.prodPool[k];
            for( ItemH IH : R.itemH[k] )
            {
              const uint32_t p0 = (uint32_t) pool.size();
              for( uint32_t q = IH.p0; q < IH.p0 + IH.pn; q++ )
              {
                uint32_t key = R.prodPool[k][q];
                if( key & 0x80000000u )
                {
                  const uint32_t pe = R.pending[key & 0x7fffffffu], kk = pe >> 16;
                  const int32_t id = O.edgeRow[kk][pe & 0xffff];
                  if( id < 0 ) continue;
                  key = ( kk << 28 ) | (uint32_t) id;
                }
                else key = ( key & 0xf0000000u ) | ( ( key & 0x0fffffffu ) + off[key >> 28] );
                if( std::find( pool.begin() + p0, pool.end(), key ) == pool.end() ) pool.push_back( key );
              }
              IH.p0 = p0; IH.pn = (uint32_t) pool.size() - p0;
              O.itemH[k].push_back( IH );
            }
          }
        }
        for( int k = 0; k < 3; k++ ) O.tb[k].insert( O.tb[k].end(), R.tb[k].begin(), R.tb[k].end() );
        for( int k = 0; k < K_NUM; k++ ) O.bytes[k] += R.bytes[k];
        O.bytesIntraLuma += R.bytesIntraLuma; O.bytesBdof += R.bytesBdof; for( int k = 0; k < 3; k++ ) O.bytesTb[k] += R.bytesTb[k];
        for( uint32_t c = (uint32_t) row0 * O.ctusX; c < (uint32_t) row1 * O.ctusX; c++ ) O.fastCtu[c] = R.fastCtu[c];
        if( O.cscale )
        {
          const int nv = 1 << ( O.h.log2_ctu - O.vpduLog2 );
          const size_t v0 = (size_t) row0 * nv * O.vpdusX, v1 = std::min( (size_t) row1 * nv, (size_t) O.vpdusY ) * O.vpdusX;
          if( v1 > v0 ) memcpy( &O.csVpduV[v0], &R.csVpduV[v0], sizeof( uint32_t ) * ( v1 - v0 ) );
          if( v1 > v0 && O.leaf ) memcpy( &O.csNeeded[v0], &R.csNeeded[v0], v1 - v0 );
        }
        // the inter stage: tiles the host writes (the sub-block motion of affine tiles sits behind what is there already), CUs whose tiles the device writes
        {
          const int32_t affOff = (int32_t) O.affMv.size();
          const size_t a0 = O.mcAff.size();
          O.mc.insert( O.mc.end(), R.mc.begin(), R.mc.end() );
          O.mcBdof.insert( O.mcBdof.end(), R.mcBdof.begin(), R.mcBdof.end() );
          O.mcDmvr.insert( O.mcDmvr.end(), R.mcDmvr.begin(), R.mcDmvr.end() );
          O.mcAff.insert( O.mcAff.end(), R.mcAff.begin(), R.mcAff.end() );
          if( !( O.h.tool_flags & VVR_TOOL_AFFINE_MV_ON_DEVICE ) ) for( size_t i = a0; i < O.mcAff.size(); i++ ) O.mcAff[i].mv[0][0] += affOff;
          O.affMv.insert( O.affMv.end(), R.affMv.begin(), R.affMv.end() );
          O.lfSb.insert( O.lfSb.end(), R.lfSb.begin(), R.lfSb.end() );
          for( McCuRef m : R.mcCus ) { m.first += O.devTiles[m.first >> 30]; O.mcCus.push_back( m ); }
          for( int k = 0; k < 3; k++ ) O.devTiles[k] += R.devTiles[k];
          O.numDmvr = std::max( O.numDmvr, R.numDmvr );
          O.resiAdd.insert( O.resiAdd.end(), R.resiAdd.begin(), R.resiAdd.end() );
        }
      }
      if( analysed && part + 1 < n )
      {
        // what the next band reads of this one: the blocks of its last cell row
        const int cy = ( std::min( row1 << O.h.log2_ctu, (int) O.h.height ) - 1 ) >> 2;
        for( int k = 0; k < O.ncomp; k++ ) for( int cx = 0; cx < O.w4; cx++ ) { const int32_t d = R.itemAtGet( k, R.cellIdx( cx, cy ) ); O.edgeRow[k][cx] = d < 0 ? -1 : d + (int32_t) off[k]; }
      }
    }
    sh.turn = part + 1;
    sh.cv.notify_all();
  } );
  if( sh.rc != VVR_OK ) { err = sh.err; return sh.rc; }
  if( validate ) return validate_cover( p, sh.area, err );
  return VVR_OK;
}

int vvr_host_build( const vvr_picture* p, PrepScratch& S, size_t* totalBytes, std::string& err, PinnedRanges* pinned, HostHelpers* helpers, bool validateRecords )
{
  S.begin( p );
  int rc;
  // an I picture without intra block copy whose CUs are all intra CUs (beginMaps looks): in parts, when there is somebody to share the work with
  // ... and, when the owner of the scratch asks for it (partsForAll: the device is waiting for work), any other picture: bands of CTU rows, what a band reads of
  // the band above resolved when the bands are joined (buildInParts)
  const bool anyInParts = S.partsForAll && !p->rpr;
  if( helpers && helpers->width() > 1 && !( p->hdr.tool_flags & VVR_TOOL_IBC ) && p->ctu_first_cu && p->num_cu >= 512 && S.ctusY >= 2 && ( p->hdr.slice_type == 2 || anyInParts ) )
  {
    if( ( rc = S.beginMaps() ) != VVR_OK ) return rc;
    if( S.allIntraCus || anyInParts )
    {
      if( ( rc = S.buildInParts( vvr_config(), *helpers, validateRecords, err ) ) != VVR_OK ) return rc;
      if( S.leaf ) { if( ( rc = S.emitLeafItems( err ) ) != VVR_OK ) return rc; }
      else if( ( rc = S.formUnits() ) != VVR_OK || ( rc = S.groupUnits() ) != VVR_OK || ( rc = S.emitUnitTable( err ) ) != VVR_OK ) return rc;
      S.layout( pinned );
      *totalBytes = S.total;
      return VVR_OK;
    }
  }
  if( helpers ) helpers->notInParts();      // (e.g. an I picture with intra block copy: the other workers need not wait for parts that will not come)
  if( validateRecords && ( rc = vvr_host_validate_records( vvr_config(), p, err ) ) != VVR_OK ) return rc;
#ifdef VVR_DEV_ENV
  if( getenv( "VVR_PHASES" ) )
  {
    // developer build: time per phase of the work-list builder
    auto now = []{ return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count(); };
    double t[7]; t[0] = now();
    rc = S.beginMaps(); S.beginLevels(); t[1] = now();
    if( rc == VVR_OK ) rc = S.buildWorkLists( err ); t[2] = now();
    if( rc == VVR_OK && !S.leaf ) rc = S.formUnits(); t[3] = now();
    if( rc == VVR_OK && !S.leaf ) rc = S.groupUnits(); t[4] = now();
    if( rc == VVR_OK ) rc = S.leaf ? S.emitLeafItems( err ) : S.emitUnitTable( err ); t[5] = now();
    if( rc != VVR_OK ) return rc;
    S.layout( pinned ); t[6] = now();
    fprintf( stderr, "[vvr] phases (ms): maps %.2f, work lists %.2f, units %.2f, groups %.2f, unit table %.2f, layout %.2f\n", t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5] );
    *totalBytes = S.total;
    return VVR_OK;
  }
#endif
  if( ( rc = S.beginMaps() ) != VVR_OK ) return rc;
  S.beginLevels();
  if( ( rc = S.buildWorkLists( err ) ) != VVR_OK ) return rc;
  if( S.leaf ) { if( ( rc = S.emitLeafItems( err ) ) != VVR_OK ) return rc; }
  else if( ( rc = S.formUnits() ) != VVR_OK || ( rc = S.groupUnits() ) != VVR_OK || ( rc = S.emitUnitTable( err ) ) != VVR_OK ) return rc;
  S.layout( pinned );
  *totalBytes = S.total;
  return VVR_OK;
}

size_t vvr_host_num_col( const vvr_picture* p )
{
  const size_t w4 = ( p->hdr.width + 3 ) >> 2, h4 = ( p->hdr.height + 3 ) >> 2;
  return ( ( w4 + 1 ) >> 1 ) * ( ( h4 + 1 ) >> 1 );
}

void vvr_host_gather_col( const vvr_picture* p, vvr_motion* dst )
{
  const size_t w4 = ( p->hdr.width + 3 ) >> 2, h4 = ( p->hdr.height + 3 ) >> 2, w8 = ( w4 + 1 ) >> 1;
  for( size_t y = 0; y < h4; y += 2 )
  {
    const vvr_motion* src = p->motion + y * w4;
    vvr_motion* d = dst + ( y >> 1 ) * w8;
    for( size_t x = 0; x < w4; x += 2 ) d[x >> 1] = src[x];
  }
}

size_t vvr_host_staged_bytes( const PrepScratch& S ) { return S.stagedEndOff; }

void vvr_host_pack( const PrepScratch& S, char* host )
{
  // (the alignment gap behind a part is cleared: what goes to the device is a function of the picture alone, not of what the ring entry held before)
  for( size_t i = S.numDirect; i < S.parts.size(); i++ )
  {
    const Part& pt = S.parts[i];
    if( pt.off >= S.stagedEndOff ) break;        // (what the device writes itself has no host image)
    if( pt.n && pt.src ) memcpy( host + pt.off, pt.src, pt.n );
    const size_t end = pt.off + ( pt.src ? pt.n : 0 ), next = i + 1 < S.parts.size() ? S.parts[i + 1].off : S.total;
    if( next > end && next - end < 4096 ) memset( host + end, 0, next - end );
  }
}

void vvr_host_upload_plan( const PrepScratch& S, std::vector<DirectCopy>& direct, size_t* stagedBegin, size_t* stagedEnd )
{
  direct.clear();
  for( size_t i = 0; i < S.numDirect; i++ ) direct.push_back( DirectCopy{ S.parts[i].src, S.parts[i].n, S.parts[i].off } );
  *stagedBegin = S.parts[S.numDirect].off;
  *stagedEnd = S.stagedEndOff;
}


void vvr_host_bind( const PrepScratch& S, vvr_prepared& q, char* base )
{
  const vvr_picture* p = S.p;
  auto at = [&]( int i ) -> char* { return i >= 0 ? base + S.parts[i].off : nullptr; };
  q.hdr = S.h;
  PicDev& d = q.pic; memset( &d, 0, sizeof( d ) );
  d.hdr = S.h; d.w4 = S.w4; d.h4 = S.h4; d.ctus_x = S.ctusX; d.ctus_y = S.ctusY;
  d.cu = (const vvr_cu*) at( S.iCu ); d.tu = (const vvr_tu*) at( S.iTu ); d.coef = (const int16_t*) at( S.iCoef );
  d.affMotion = (const vvr_motion*) at( S.iAffMv );
  d.lfp[0] = (const vvr_lfp*) at( S.iL0 ); d.lfp[1] = (const vvr_lfp*) at( S.iL1 );
  q.lfpOnDevice = S.lfpOnDevice; q.lfpDev[0] = (vvr_lfp*) at( S.iL0 ); q.lfpDev[1] = (vvr_lfp*) at( S.iL1 );
  q.lfCell = (LfCell*) at( S.iLfTu ); q.lfCellC = (LfCell*) at( S.iLfTuC ); q.lfMv = (LfMv*) at( S.iLfMotion ); q.lfRef = q.lfMv ? (uint32_t*) ( q.lfMv + (size_t) S.w4 * S.h4 ) : nullptr;      // (vectors, then reference indices)
  q.lfSb = (const LfSbCell*) at( S.iLfSb ); q.numLfSb = (int) S.lfSb.size(); q.numCu = p->num_cu; q.numTu = p->num_tu;
  d.sao = (const vvr_sao_ctu*) at( S.iSao ); d.alf = (const vvr_alf_ctu*) at( S.iAlf ); d.alf_params = (const vvr_alf_params*) at( S.iAlfP );
  d.lmcs = (const vvr_lmcs_params*) at( S.iLmcs ); d.scaling = (const vvr_scaling_list*) at( S.iSl ); d.wp = (const vvr_wp_params*) at( S.iWp ); d.rpr = (const vvr_rpr_params*) at( S.iRpr );
  d.ctuSlice = (const uint16_t*) at( S.iCtuSlice ); d.ctuTile = (const uint16_t*) at( S.iCtuTile );
  d.slices = (const vvr_slice_header*) at( S.iSlices ); d.numAlfSets = (int) std::max<uint32_t>( 1, p->num_alf_sets ); d.numWpSets = (int) std::max<uint32_t>( 1, p->num_wp_sets );
  d.subpics = (const vvr_subpic*) at( S.iSubpics ); d.ctuSubpic = (const uint16_t*) at( S.iCtuSubpic );
  d.csVpdu = (const uint32_t*) at( S.iCsVpdu ); d.vpdusX = S.vpdusX; d.vpduLog2 = S.vpduLog2;
  q.mcItems = (McItem*) at( S.iMc ); q.numMc = (int) S.mc.size();
  q.mcDev = (McItem*) at( S.iMcDev[0] ); q.numMcDev = (int) S.devTiles[0];
  q.bdofItems = (McItem*) at( S.iMcDev[1] ); q.numBdofItems = (int) S.devTiles[1];
  q.dmvrItems = (McItem*) at( S.iMcDev[2] ); q.numDmvrItems = (int) S.devTiles[2];
  q.mcCus = (McCuRef*) at( S.iMcCus ); q.numMcCus = (int) S.mcCus.size();
  q.affItems = (McItem*) at( S.iMcA ); q.numAffItems = (int) S.mcAff.size();
  q.rprItems = (McItem*) at( S.iMcR ); q.numRprItems = (int) S.mcRpr.size();
  q.numDmvr = S.numDmvr;
  for( int k = 0; k < 3; k++ ) { q.tbItems[k] = (TbItem*) at( S.iTb[k] ); q.numTb[k] = (int) S.tb[k].size(); }
  q.intraItems = (IntraItem*) at( S.iIntra ); q.numIntra = (int) S.intraAll.size(); q.intraLeaf = S.leaf; q.intraFine = !S.leaf && S.intraFine && !S.unitsDev.empty();
  q.units = (IntraUnit*) at( S.iUnits ); q.numActive = (int) S.unitsDev.size(); q.intraWorkgroups = S.intraWorkgroups;
  q.resiItems = (IntraItem*) at( S.iResi ); q.numResi = (int) S.resiAdd.size(); q.numLumaUnits = S.numLumaUnits; q.intraWorkgroupsChroma = S.intraWorkgroupsChroma;
  memcpy( q.bytes, S.bytes, sizeof( q.bytes ) );
  q.bytesBdof = S.bytesBdof; q.bytesIntraLuma = S.bytesIntraLuma; for( int k = 0; k < 3; k++ ) q.bytesTb[k] = S.bytesTb[k];
}
