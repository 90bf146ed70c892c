// tools/synth.cpp — synthetic PRE-PARSED picture generator (measurement / test infrastructure, not the product).
//
// Emits what the host side of a decoder (CABAC parse + MV derivation + boundary-strength derivation) would hand to the
// reconstruction stage for one picture: vvr_cu / vvr_tu records, packed levels, the motion field, deblocking edge
// parameters, SAO / ALF controls (include/vvr.h).  Content statistics follow SURVEY.md §8(d) "configs as concrete
// synthetic inputs".  Everything is drawn from a counter-based RNG seeded per picture, so a (seed, parameters) pair
// fully determines the picture — the GPU box regenerates the same pictures the CPU oracle saw.
//
// All values are "conformant-stream-like": partitions are legal power-of-two CUs inside the picture, MVs obey
// clipMv (Mv.cpp:64-82), deblocking filter lengths obey the non-overlap rules of LoopFilter.cpp:910-922, so that the
// reference's sequential edge loop and a parallel edge filter are equivalent.
#include "../include/vvr.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct Rng {   // splitmix64
  uint64_t s;
  explicit Rng( uint64_t seed ) : s( seed * 0x9E3779B97F4A7C15ull + 0x1234567ull ) {}
  uint64_t next() { uint64_t z = ( s += 0x9E3779B97F4A7C15ull ); z = ( z ^ ( z >> 30 ) ) * 0xBF58476D1CE4E5B9ull; z = ( z ^ ( z >> 27 ) ) * 0x94D049BB133111EBull; return z ^ ( z >> 31 ); }
  uint32_t u( uint32_t n ) { return (uint32_t) ( next() % n ); }                 // [0,n)
  double   f() { return ( next() >> 11 ) * ( 1.0 / 9007199254740992.0 ); }       // [0,1)
  bool     p( double prob ) { return f() < prob; }
  int      laplace( double b ) { double x = f() - 0.5; double v = -b * ( x < 0 ? -1 : 1 ) * std::log( 1 - 2 * std::fabs( x ) + 1e-12 ); return (int) std::lround( v ); }
};

} // namespace

extern "C" {

typedef struct vvs_params {
  uint64_t seed;
  uint16_t width, height;
  uint8_t  bit_depth, log2_ctu, chroma_format, slice_type;
  uint32_t tool_flags;          // VVR_TOOL_* to put in the header (and to draw tools from)
  int8_t   num_ref[2];
  int32_t  poc;
  int32_t  ref_poc[2][VVR_MAX_REFS];
  int16_t  ref_slot[2][VVR_MAX_REFS];
  int16_t  out_slot;
  int8_t   base_qp;             // 32 for the RA QP32 config
  uint8_t  min_cu_log2;         // 3: min CU 8x8
  float    p_intra;             // fraction of CUs coded intra (I slices: 1)
  float    p_bi;                // of inter CUs
  float    p_coded;             // fraction of TUs with a coded luma block
  float    p_coded_chroma;
  float    p_small_corner;      // fraction of coded blocks whose levels sit in a <= 8x8 corner
  float    p_mts;               // explicit MTS (luma, size <= 32)
  float    p_ts;                // transform skip
  float    p_lfnst;             // of intra CUs
  float    p_split_scale;       // scales the split probabilities (1 = mean CU ~32x32)
  float    mv_sigma;            // luma samples
  float    p_sao, p_alf_luma, p_alf_chroma, p_ccalf;
  float    p_imv_hpel;
  float    p_jccr;
  float    p_mrl, p_bdpcm;
  float    p_affine;            // of inter CUs >= 8x8 (half of them 6-parameter)
  float    p_geo;               // of inter CUs that can use the geometric partitioning mode
  float    p_ciip;              // of inter CUs that can combine inter and intra prediction
  float    p_sbtmvp;            // of inter CUs >= 8x8: sub-block temporal merge (per-8x8 motion)
  float    p_bcw;               // of bi-predicted CUs with at least 256 samples: unequal CU-level weights
  float    p_cclm;              // of intra CUs: chroma predicted from the reconstructed luma (CCLM, MDLM_L, MDLM_T)
  float    p_mip;               // of intra CUs: matrix-based luma prediction
  float    p_sbt;               // of inter CUs (not CIIP, at most 64x64): sub-block transform (residual in one half / quarter of the CU)
  float    p_isp;               // of intra CUs (no MRL / BDPCM / MIP): intra sub-partitions, four luma partitions predicted one after the other
  float    dual_tree;           // > 0: I pictures use separate luma and chroma coding trees below 64x64 (qtbtt_dual_tree_intra_flag); 2: luma CUs down to 4x4; 3: and ISP on 4xN / Nx4 CUs (1xN, Nx1, 2xN, Nx2 partitions)
  float    p_ibc;               // with VVR_TOOL_IBC: of the CUs that would be intra (luma at most 64x64, not in a chroma tree): intra block copy,
                                // where a block vector into the valid part of the IBC virtual buffer is found
  uint8_t  num_slices;          // > 1: the picture is cut into that many slices (one tile: bands of CTU rows; several tiles: runs of tiles in raster order)
  uint8_t  tile_cols, tile_rows;// > 1: uniform tile grid.  Whether the loop filters cross these boundaries is in tool_flags (VVR_TOOL_NO_LF_ACROSS_*)
  uint16_t wrap_offset;         // > 0: horizontal reference wrap-around with this period (luma samples; header field of the same name), and inter CUs close
                                // to the left / right picture edge point across it more often
  uint8_t  subpics;             // bit 0: one sub-picture per tile (needs a tile grid; slices are then one per tile as well); bits 1-2: treated as a picture:
                                // 0 none, 1 all, 2 some; bits 3-4: loop filters across sub-picture boundaries: 0 everywhere, 1 nowhere, 2 for some
  uint8_t  intra_slices;        // P / B pictures with several slices: bit ( k & 7 ) set = slice k holds intra (and IBC) CUs only - an I slice in a picture of another kind
  uint8_t  virtual_boundaries;  // bits 0-1: number of vertical, bits 2-3: of horizontal virtual boundaries of the in-loop filters (picture header); bit 4: the first
                                // of each direction lies on a CTU boundary
  uint16_t scaled_refs[2];      // reference picture resampling: bit i of [l] = reference picture i of list l is a scaled one (vvr_rpr_ref.scaled; the caller attaches
                                // the table): CUs predicting from it take no BDOF / DMVR (InterPrediction.cpp:1431-1435) and their MVs are left as drawn
  uint16_t mv_window;           // with wrap_offset: how many luma samples an MV may point beyond the window clipMv leaves it in (0: two CTUs + 64, within one wrap period of
                                // the margins); a parsed stream may carry vectors several periods out - e.g. 2000 to generate those
} vvs_params;

typedef struct vvs_buffers {     // caller-allocated, sized with vvs_bounds()
  vvr_cu*      cu;      uint32_t max_cu;
  vvr_tu*      tu;      uint32_t max_tu;
  int16_t*     coef;    uint64_t max_coef;
  uint32_t*    ctu_first_cu;
  vvr_motion*  motion;
  vvr_lfp*     lfp[2];
  vvr_sao_ctu* sao;
  vvr_alf_ctu* alf;
  vvr_alf_params* alf_params;
  vvr_lmcs_params* lmcs;          // filled when VVR_TOOL_LMCS is in tool_flags
  vvr_wp_params*   wp;            // filled when VVR_TOOL_WP is in tool_flags (P / B pictures)
  vvr_scaling_list* scaling;      // filled when VVR_TOOL_SCALING_LIST is in tool_flags
  uint16_t*    ctu_slice;         // [num_ctu], filled when the parameters ask for more than one slice (else left alone)
  uint16_t*    ctu_tile;          // [num_ctu], likewise for tiles
  vvr_subpic*  subpics;           // [up to 255], filled when the parameters ask for sub-pictures
  // outputs
  uint32_t     num_cu, num_tu; uint64_t num_coef; uint32_t num_dmvr; uint32_t num_subpics;
  vvr_pic_header hdr;
} vvs_buffers;

__attribute__((visibility("default")))
void vvs_bounds( const vvs_params* P, uint32_t* max_cu, uint32_t* max_tu, uint64_t* max_coef )
{
  const uint32_t m = 1u << P->min_cu_log2;
  const uint32_t n = ( ( P->width + m - 1 ) / m ) * ( ( P->height + m - 1 ) / m );
  const uint32_t trees = P->dual_tree >= 2.0f ? 5 : ( P->dual_tree > 0 || P->min_cu_log2 == 2 ) ? 2 : 1;                 // dual tree: luma and chroma CUs (luma down to 4x4: four times as many)
  *max_cu = n * trees; *max_tu = ( P->p_isp > 0 ? 4 * n : n + n / 4 ) * trees + 16;      // ISP: four TUs per CU
  *max_coef = (uint64_t) P->width * P->height * 3 / 2 + 4096;
}

__attribute__((visibility("default")))
void vvs_default_params( vvs_params* P )
{
  memset( P, 0, sizeof( *P ) );
  P->seed = 1234; P->width = 3840; P->height = 2160; P->bit_depth = 10; P->log2_ctu = 7; P->chroma_format = 1; P->slice_type = 0;
  P->base_qp = 32; P->min_cu_log2 = 3;
  P->p_intra = 0.15f; P->p_bi = 0.6f; P->p_coded = 0.35f; P->p_coded_chroma = 0.2f; P->p_small_corner = 0.8f; P->p_mts = 0.15f; P->p_ts = 0.03f; P->p_lfnst = 0.2f;
  P->p_split_scale = 1.0f; P->mv_sigma = 8.0f; P->p_sao = 0.4f; P->p_alf_luma = 0.8f; P->p_alf_chroma = 0.5f; P->p_ccalf = 0.3f; P->p_imv_hpel = 0.1f; P->p_jccr = 0.1f; P->p_mrl = 0.15f; P->p_bdpcm = 0.03f;
  P->p_affine = 0.0f; P->p_geo = 0.0f; P->p_ciip = 0.0f; P->p_sbtmvp = 0.0f; P->p_bcw = 0.05f; P->p_cclm = 0.0f; P->p_mip = 0.0f; P->p_sbt = 0.0f; P->p_isp = 0.0f; P->dual_tree = 0.0f; P->p_ibc = 0.0f;
}

namespace {

struct Gen {
  const vvs_params& P; vvs_buffers& B; Rng rng;
  int W, H, w4, h4, ctu, bd;
  std::vector<int32_t> cuOf4;      // per 4x4: CU index
  std::vector<int32_t> tuOf4;      // per 4x4: TU index
  std::vector<int32_t> cuOf4C, tuOf4C;   // dual tree: the same maps of the chroma tree
  int curTree = VVR_TREE_JOINT;    // tree the CUs being added belong to
  bool curI = false;               // the CTU being generated lies in an I slice (or the picture is an I picture)
  int modeType = 0;                // mode constraint of the current sub-tree (SCIPU): 0 all, 1 inter only, 2 intra only (local dual tree)
  bool cclmOk = true;              // CCLM allowed for the chroma CUs being added (CU::checkCCLMAllowed, UnitTools.cpp:3439)
  Gen( const vvs_params& p, vvs_buffers& b ) : P( p ), B( b ), rng( p.seed ) {}

  // slices and tiles: index of every CTU; two positions see each other (intra availability, CIIP neighbours: CodingStructure::getCURestricted)
  // only inside one slice and one tile; the loop filters additionally obey the pps_loop_filter_across_* flags
  std::vector<uint16_t> sliceOfCtu, tileOfCtu;
  int ctusX = 0, ctusY = 0;
  int ctuOfPos( int x, int y ) const { return ( y >> P.log2_ctu ) * ctusX + ( x >> P.log2_ctu ); }
  bool sameSliceTile( int a, int b ) const { return sliceOfCtu[a] == sliceOfCtu[b] && tileOfCtu[a] == tileOfCtu[b]; }
  // is the edge left of / above the 4x4 unit (x4, y4) on a virtual boundary of the picture header?  (LoopFilter::xDeriveEdgefilterParam, LoopFilter.cpp:669)
  bool onVirtualBoundary( int d, int x4, int y4 ) const
  {
    const vvr_pic_header& h = B.hdr;
    if( d == 0 ) { for( int i = 0; i < h.num_ver_vb; i++ ) if( h.vb_pos_x[i] == ( x4 << 2 ) ) return true; }
    else         { for( int i = 0; i < h.num_hor_vb; i++ ) if( h.vb_pos_y[i] == ( y4 << 2 ) ) return true; }
    return false;
  }
  std::vector<uint16_t> subpicOfCtu; std::vector<vvr_subpic> subpicV;
  bool lfMayCross( int a, int b ) const
  {
    if( ( P.tool_flags & VVR_TOOL_NO_LF_ACROSS_SLICES ) && sliceOfCtu[a] != sliceOfCtu[b] ) return false;
    if( ( P.tool_flags & VVR_TOOL_NO_LF_ACROSS_TILES ) && tileOfCtu[a] != tileOfCtu[b] ) return false;
    // deblocking between two sub-pictures needs the flag of both (LoopFilter.cpp:1074-1088)
    if( !subpicV.empty() && subpicOfCtu[a] != subpicOfCtu[b] && !( subpicV[subpicOfCtu[a]].lf_across && subpicV[subpicOfCtu[b]].lf_across ) ) return false;
    return true;
  }
  void layoutSlicesAndTiles()
  {
    ctusX = ( W + ctu - 1 ) / ctu; ctusY = ( H + ctu - 1 ) / ctu;
    const int n = ctusX * ctusY;
    sliceOfCtu.assign( n, 0 ); tileOfCtu.assign( n, 0 );
    const int tc = std::max<int>( 1, std::min<int>( P.tile_cols, ctusX ) ), tr = std::max<int>( 1, std::min<int>( P.tile_rows, ctusY ) );
    for( int cy = 0; cy < ctusY; cy++ ) for( int cx = 0; cx < ctusX; cx++ ) tileOfCtu[cy * ctusX + cx] = (uint16_t) ( ( cy * tr / ctusY ) * tc + cx * tc / ctusX );
    const int numTiles = tc * tr, ns = std::max<int>( 1, P.num_slices );
    if( ns > 1 )
    {
      if( numTiles > 1 )
      {
        // raster-scan slices: runs of complete tiles
        const int k = std::min( ns, numTiles );
        for( int a = 0; a < n; a++ ) sliceOfCtu[a] = (uint16_t) ( tileOfCtu[a] * k / numTiles );
      }
      else
      {
        // one tile, rectangular slices: bands of complete CTU rows
        const int k = std::min( ns, ctusY );
        for( int a = 0; a < n; a++ ) sliceOfCtu[a] = (uint16_t) ( ( a / ctusX ) * k / ctusY );
      }
    }
    subpicV.clear(); B.num_subpics = 0;
    if( ( P.subpics & 1 ) && numTiles > 1 && B.subpics )
    {
      // one sub-picture per tile, one (rectangular) slice per sub-picture (pps_single_slice_per_subpic_flag)
      subpicOfCtu = tileOfCtu; sliceOfCtu = tileOfCtu;
      subpicV.resize( numTiles );
      for( int k = 0; k < numTiles; k++ ) { vvr_subpic& sp = subpicV[k]; memset( &sp, 0, sizeof( sp ) ); sp.x0 = sp.y0 = 0xffff; }
      for( int a = 0; a < n; a++ )
      {
        vvr_subpic& sp = subpicV[tileOfCtu[a]];
        const int x0 = ( a % ctusX ) * ctu, y0 = ( a / ctusX ) * ctu, x1 = std::min( W, x0 + ctu ) - 1, y1 = std::min( H, y0 + ctu ) - 1;
        sp.x0 = (uint16_t) std::min<int>( sp.x0, x0 ); sp.y0 = (uint16_t) std::min<int>( sp.y0, y0 ); sp.x1 = (uint16_t) std::max<int>( sp.x1, x1 ); sp.y1 = (uint16_t) std::max<int>( sp.y1, y1 );
      }
      const int tm = ( P.subpics >> 1 ) & 3, lm = ( P.subpics >> 3 ) & 3;
      for( int k = 0; k < numTiles; k++ )
      {
        subpicV[k].treated_as_pic = (uint8_t) ( tm == 1 || ( tm == 2 && rng.p( 0.6 ) ) );
        subpicV[k].lf_across = (uint8_t) ( lm == 0 || ( lm == 2 && rng.p( 0.5 ) ) );
      }
      memcpy( B.subpics, subpicV.data(), sizeof( vvr_subpic ) * numTiles ); B.num_subpics = (uint32_t) numTiles;
      if( B.ctu_slice ) memcpy( B.ctu_slice, sliceOfCtu.data(), sizeof( uint16_t ) * n );
    }
    if( B.ctu_slice && ns > 1 && subpicV.empty() ) memcpy( B.ctu_slice, sliceOfCtu.data(), sizeof( uint16_t ) * n );
    if( B.ctu_tile && numTiles > 1 ) memcpy( B.ctu_tile, tileOfCtu.data(), sizeof( uint16_t ) * n );
  }

  // intra block copy: what the IBC virtual buffer of every CTU row holds (CodingStructure::fillIBCbuffer, CodingStructure.cpp:550): per
  // 4x4 cell of the buffer the picture column (in 4-sample units) of the samples stored there, -1 = nothing valid; luma and chroma
  // are filled by different CUs in separate trees.  The buffer is 256 * 128 / CtbSize luma samples wide and wraps.
  bool ibcOn = false;
  int ibcW4 = 0;
  std::vector<int32_t> vbL, vbC;
  bool ibcCellsValid( const std::vector<int32_t>& vb, int rx, int ry, int w, int h, int ctuY0 ) const
  {
    if( rx < 0 || ry < ctuY0 || rx + w > W || ry + h > std::min( H, ctuY0 + ctu ) ) return false;
    for( int cy = ry >> 2; cy <= ( ry + h - 1 ) >> 2; cy++ ) for( int cx = rx >> 2; cx <= ( rx + w - 1 ) >> 2; cx++ )
      if( vb[(size_t) cy * ibcW4 + ( cx % ibcW4 )] != cx ) return false;
    return true;
  }
  // a block vector whose reference block is completely valid in the virtual buffer (luma, and chroma at the halved vector for CUs
  // that carry chroma); tries a few candidates left of / above the CU inside the CTU row
  bool findBv( int x, int y, int w, int h, bool withChroma, int& bvx, int& bvy )
  {
    const int ctuY0 = y & ~( ctu - 1 );
    for( int t = 0; t < 24; t++ )
    {
      int rx, ry;
      const int kind = rng.u( 3 );
      if( kind == 0 ) { rx = x - w - (int) rng.u( 48 ); ry = y + (int) rng.u( 17 ) - 8; }                  // left of the CU
      else if( kind == 1 ) { rx = x + (int) rng.u( 17 ) - 8; ry = y - h - (int) rng.u( 32 ); }             // above it
      else { rx = x - (int) rng.u( std::min( x, 4 * ibcW4 - ctu ) + 1 ); ry = ctuY0 + (int) rng.u( ctu - h + 1 ); }   // anywhere in the buffer
      if( rng.p( 0.3 ) ) { rx &= ~3; ry &= ~3; }
      const int vx = rx - x, vy = ry - y;
      if( !ibcCellsValid( vbL, rx, ry, w, h, ctuY0 ) ) continue;
      if( withChroma && !ibcCellsValid( vbC, x + 2 * ( vx >> 1 ), y + 2 * ( vy >> 1 ), w, h, ctuY0 ) ) continue;
      bvx = vx; bvy = vy;
      return true;
    }
    return false;
  }
  void ibcTrack( const vvr_cu& cu )
  {
    const int vS = std::min( ctu, 64 );
    if( cu.tree != VVR_TREE_CHROMA && ( cu.x % vS ) == 0 && ( cu.y % vS ) == 0 )
    {
      // start of a VPDU: the area half a buffer away is given up (DecCu.cpp:84-91, the virtual buffer reset of the decoding process)
      const int rw = std::max<int>( vS, cu.w ), rh = std::max<int>( vS, cu.h );
      for( int yy = 0; yy < rh && cu.y + yy < H; yy += 4 ) for( int xx = 0; xx < rw; xx += 4 )
      {
        const size_t k = (size_t) ( ( cu.y + yy ) >> 2 ) * ibcW4 + ( ( ( cu.x + xx ) >> 2 ) + ibcW4 / 2 ) % ibcW4;
        vbL[k] = -1; vbC[k] = -1;
      }
    }
    for( int yy = 0; yy < cu.h; yy += 4 ) for( int xx = 0; xx < cu.w; xx += 4 )
    {
      const int cx = ( cu.x + xx ) >> 2;
      const size_t k = (size_t) ( ( cu.y + yy ) >> 2 ) * ibcW4 + cx % ibcW4;
      if( cu.tree != VVR_TREE_CHROMA ) vbL[k] = cx;
      if( cu.tree != VVR_TREE_LUMA ) vbC[k] = cx;
    }
  }

  bool wpOn = false;
  bool wpPresent( int l, int r ) const { return wpOn && r >= 0 && ( B.wp->e[l][r][0].present || B.wp->e[l][r][1].present || B.wp->e[l][r][2].present ); }
  // pred_weight_table(): denominators 0..7, weights 1 << denom + delta (delta in -128..127), offsets in -128..127 (8-bit units);
  // one flag for luma and one for both chroma components per reference picture; own random stream
  void genWp()
  {
    Rng r( P.seed * 0x9E3779B97F4A7C15ull + 77 );
    vvr_wp_params& w = *B.wp; memset( &w, 0, sizeof( w ) );
    w.log2_denom[0] = (uint8_t) r.u( 8 );
    w.log2_denom[1] = (uint8_t) std::min( 7, std::max( 0, (int) w.log2_denom[0] + (int) r.u( 5 ) - 2 ) );
    for( int l = 0; l < 2; l++ ) for( int i = 0; i < VVR_MAX_REFS; i++ )
    {
      const bool fl = i < P.num_ref[l] && r.p( 0.6 ), fc = i < P.num_ref[l] && P.chroma_format && r.p( 0.5 );
      for( int c = 0; c < 3; c++ )
      {
        vvr_wp_entry& e = w.e[l][i][c];
        const int den = w.log2_denom[c ? 1 : 0];
        e.weight = (int16_t) ( 1 << den ); e.offset = 0; e.present = (uint8_t) ( c ? fc : fl );
        if( !e.present ) continue;
        const int span = std::max( 2, ( 1 << den ) / 2 );
        e.weight = (int16_t) ( ( 1 << den ) + std::min( 127, std::max( -128, r.laplace( span / 3.0 ) ) ) );
        e.offset = (int16_t) ( r.p( 0.1 ) ? ( r.p( 0.5 ) ? 127 : -128 ) : std::min( 127, std::max( -128, r.laplace( 6.0 ) ) ) );
      }
    }
  }

  // scaling_list_data(): 28 matrices (2 of 2x2, 6 of 4x4, 20 of 8x8) with entries 1..255 around the neutral 16, smoothly growing
  // with the frequency like real quantisation matrices, plus the DC entries of the up-sampled ones; own random stream
  void genScalingList()
  {
    Rng r( P.seed * 0xD1B54A32D192ED03ull + 99 );
    vvr_scaling_list& L = *B.scaling; memset( &L, 0, sizeof( L ) );
    for( int id = 0; id < 28; id++ )
    {
      const int n = id < 2 ? 2 : id < 8 ? 4 : 8;
      const int base = 8 + (int) r.u( 17 ), slope = (int) r.u( 4 * 8 / n + 1 );
      for( int y = 0; y < n; y++ ) for( int x = 0; x < n; x++ )
        L.coef[id][y * n + x] = (uint8_t) std::min( 255, std::max( 1, base + slope * ( x + y ) + r.laplace( 2.0 ) ) );
      if( r.p( 0.1 ) ) L.coef[id][r.u( n * n )] = (uint8_t) ( r.p( 0.5 ) ? 255 : 1 );
      L.dc[id] = (uint8_t) ( id >= 14 ? std::min( 255, std::max( 1, base + r.laplace( 2.0 ) ) ) : 16 );
    }
  }

  // reference wrap-around: inter CUs near the left / right picture edge point across it often, some of them further than the wrap copy's margin
  void wrapBias( int32_t mv[2], int x, int w )
  {
    if( !P.wrap_offset || !rng.p( 0.4 ) ) return;
    const int reach = ( ctu + 96 ) * 16;
    if( x < 2 * ctu ) mv[0] -= (int32_t) rng.u( (uint32_t) reach );
    else if( x + w > W - 2 * ctu ) mv[0] += (int32_t) rng.u( (uint32_t) reach );
  }
  void clipMv( int32_t mv[2], int x, int y ) const   // clipMvInPic, Mv.cpp:64 (with wrap-around the stream may carry MVs beyond it: kept inside a wider window)
  {
    const int off = 8;
    const int beyond = P.wrap_offset ? ( P.mv_window ? P.mv_window : 2 * ctu + 64 ) : 0;
    const int horMax = ( W + off - x - 1 + beyond ) * 16, horMin = ( -ctu - off - x + 1 - beyond ) * 16;
    const int verMax = ( H + off - y - 1 ) * 16, verMin = ( -ctu - off - y + 1 ) * 16;
    mv[0] = std::min( horMax, std::max( horMin, mv[0] ) );
    mv[1] = std::min( verMax, std::max( verMin, mv[1] ) );
  }

  void genLevels( vvr_tu& tu, int c, int bw, int bh, bool ts, bool bdpcm, bool lfnst )
  {
    tu.coef_off[c] = (uint32_t) B.num_coef;
    int16_t* dst = B.coef + B.num_coef;
    if( bdpcm )
    {   // BDPCM: the full block of (DPCM-coded) levels is transmitted (Quant.cpp:239); keep them sparse and small
      tu.max_scan_x[c] = (uint8_t) ( bw - 1 ); tu.max_scan_y[c] = (uint8_t) ( bh - 1 );
      for( int i = 0; i < bw * bh; i++ ) dst[i] = (int16_t) ( rng.p( 0.12 ) ? rng.laplace( 1.0 ) : 0 );
      dst[0] = dst[0] ? dst[0] : 1;
      B.num_coef += (uint64_t) bw * bh;
      return;
    }
    if( lfnst )
    {   // LFNST: only the first 8 (4x4 / 8x8 blocks) or 16 positions of the top-left 4x4 diagonal scan may be non-zero
      static const uint8_t scanXY[16][2] = { {0,0},{0,1},{1,0},{0,2},{1,1},{2,0},{0,3},{1,2},{2,1},{3,0},{1,3},{2,2},{3,1},{2,3},{3,2},{3,3} };
      const int maxNz = ( ( bw == 4 && bh == 4 ) || ( bw == 8 && bh == 8 ) ) ? 8 : 16;
      const int nz = 1 + rng.u( maxNz );
      int mx = 0, my = 0;
      for( int i = 0; i < nz; i++ ) { mx = std::max<int>( mx, scanXY[i][0] ); my = std::max<int>( my, scanXY[i][1] ); }
      tu.max_scan_x[c] = (uint8_t) mx; tu.max_scan_y[c] = (uint8_t) my;
      const int cw = mx + 1, ch = my + 1;
      for( int i = 0; i < cw * ch; i++ ) dst[i] = 0;
      for( int i = 0; i < nz; i++ ) { int v = rng.laplace( 2.0 ); if( i == 0 ) v *= 3; if( i == nz - 1 && v == 0 ) v = 1; dst[scanXY[i][1] * cw + scanXY[i][0]] = (int16_t) v; }
      B.num_coef += (uint64_t) cw * ch;
      return;
    }
    // corner extents
    int mx, my;
    const int capW = std::min( bw, 32 ), capH = std::min( bh, 32 );
    if( ts ) { mx = bw - 1; my = bh - 1; }
    else if( rng.p( P.p_small_corner ) ) { mx = rng.u( std::min( capW, 8 ) ); my = rng.u( std::min( capH, 8 ) ); if( rng.p( 0.15 ) ) mx = my = 0; }
    else { mx = rng.u( capW ); my = rng.u( capH ); }
    const int mts = tu.mts_idx[c];
    if( mts > 1 ) { mx = std::min( mx, 15 ); my = std::min( my, 15 ); }   // MTS zero-out: only the 16x16 corner can be non-zero
    tu.max_scan_x[c] = (uint8_t) mx; tu.max_scan_y[c] = (uint8_t) my;
    const int cw = mx + 1, ch = my + 1;
    bool any = false;
    for( int y = 0; y < ch; y++ ) for( int x = 0; x < cw; x++ )
    {
      int v = 0;
      const double dens = ( x == mx && y == my ) ? 1.0 : 0.6 / ( 1.0 + 0.15 * ( x + y ) );
      if( rng.p( dens ) ) { v = rng.laplace( 1.5 ); if( x == 0 && y == 0 ) v *= 4; if( v == 0 ) v = rng.p( 0.5 ) ? 1 : -1; }
      dst[y * cw + x] = (int16_t) v; any |= v != 0;
    }
    if( !any ) dst[my * cw + mx] = 1;
    B.num_coef += (uint64_t) cw * ch;
  }

  void addCu( int x, int y, int w, int h )
  {
    vvr_cu& cu = B.cu[B.num_cu]; memset( &cu, 0, sizeof( cu ) );
    const uint32_t cuIdx = B.num_cu++;
    cu.x = x; cu.y = y; cu.w = w; cu.h = h; cu.tree = (uint8_t) curTree;
    const bool treeL = curTree == VVR_TREE_LUMA, treeC = curTree == VVR_TREE_CHROMA;
    cu.qp = (int8_t) std::min( 63, std::max( 0, P.base_qp + (int) rng.u( 7 ) - 3 ) );
    cu.bcw_idx = 2; cu.ref_idx[0] = cu.ref_idx[1] = -1;
    const bool isI = curI;
    // (a 4x4 CU is never inter predicted: pred_mode is inferred; it only gets here in 4:0:0 pictures, where no chroma constraint forces a mode)
    const bool intraCand = isI || modeType == 2 || ( w == 4 && h == 4 ) || ( modeType != 1 && std::max( w, h ) <= 64 && rng.p( P.p_intra ) );
    // intra block copy instead of intra prediction (IBC CUs take the place of intra CUs in the coding tree: same size limits)
    bool ibc = false;
    if( intraCand && ibcOn && P.p_ibc > 0 && !treeC && w <= 64 && h <= 64 && rng.p( P.p_ibc ) )
    {
      int bvx = 0, bvy = 0;
      ibc = findBv( x, y, w, h, !treeL && P.chroma_format, bvx, bvy );
      if( ibc ) { cu.mv[0][0][0] = bvx * 16; cu.mv[0][0][1] = bvy * 16; cu.inter_dir = 1; cu.intra_dir[0] = 1; /* DC: what a co-located chroma CU derives */ }
    }
    const bool intra = intraCand && !ibc;
    cu.pred_mode = ibc ? VVR_PRED_IBC : intra ? VVR_PRED_INTRA : VVR_PRED_INTER;
    if( ibc ) {}
    else if( intra )
    {
      const int r = rng.u( 100 );
      cu.intra_dir[0] = r < 20 ? 0 : r < 35 ? 1 : 2 + rng.u( 65 );
      const int rc = rng.u( 100 );
      cu.intra_dir[1] = rc < 40 ? cu.intra_dir[0] : rc < 55 ? 0 : rc < 65 ? 1 : rc < 75 ? 18 : rc < 85 ? 50 : 2 + rng.u( 65 );   // DM / planar / DC / hor / ver / any (CCLM: not generated yet)
      if( P.chroma_format && !treeL && cclmOk && rng.p( P.p_cclm ) ) cu.intra_dir[1] = (uint8_t) ( 67 + rng.u( 3 ) );     // LM_CHROMA_IDX, MDLM_L_IDX, MDLM_T_IDX
      cu.lfnst_intra_mode = cu.intra_dir[0];
      if( treeC )
      {
        // chroma CU of a dual tree: the derived mode is the luma mode at the centre of the co-located luma area
        // (PU::getCoLocatedIntraLumaMode: MIP -> planar); it also selects the LFNST set of CCLM blocks
        const vvr_cu& lc = B.cu[cuOf4[( ( y + h / 2 ) >> 2 ) * w4 + ( ( x + w / 2 ) >> 2 )]];
        const uint8_t colMode = ( lc.flags & VVR_CU_MIP ) ? 0 : lc.intra_dir[0];
        if( rc < 40 ) cu.intra_dir[1] = colMode;
        cu.intra_dir[0] = colMode; cu.lfnst_intra_mode = colMode;
        // LFNST of a chroma tree CU applies to both chroma blocks (at least 4x4 each)
        if( ( P.tool_flags & VVR_TOOL_LFNST ) && w >= 8 && h >= 8 && rng.p( P.p_lfnst ) ) cu.lfnst_idx = (uint8_t) ( 1 + rng.u( 2 ) );      // (lfnst_idx needs min( width, height ) >= 4 of the chroma blocks)
      }
      // multiple reference lines: luma only, never on the first row of a CTU, not with planar (intra_luma_ref_idx semantics)
      if( !treeC && ( y & ( ctu - 1 ) ) != 0 && cu.intra_dir[0] != 0 && rng.p( P.p_mrl ) ) cu.multi_ref_idx = (uint8_t) ( 1 + rng.u( 2 ) );
      // BDPCM (implies transform skip of the luma block)
      if( !treeC && w <= 32 && h <= 32 && !cu.multi_ref_idx && rng.p( P.p_bdpcm ) ) { cu.bdpcm[0] = (uint8_t) ( 1 + rng.u( 2 ) ); cu.intra_dir[0] = cu.bdpcm[0] == 1 ? 18 : 50; cu.lfnst_intra_mode = cu.intra_dir[0]; }
      // chroma BDPCM (intra_bdpcm_chroma_flag: chroma blocks of at most 32x32; implies transform skip of both chroma blocks, horizontal or vertical prediction; no LFNST
      // on a CU that has a transform-skip block)
      if( P.chroma_format && !treeL && ( w >> 1 ) <= 32 && ( h >> 1 ) <= 32 && ( w >> 1 ) >= 4 && ( h >> 1 ) >= 4 && rng.p( P.p_bdpcm ) )
      {
        cu.bdpcm[1] = (uint8_t) ( 1 + rng.u( 2 ) ); cu.intra_dir[1] = cu.bdpcm[1] == 1 ? 18 : 50; cu.lfnst_idx = 0;
      }
      // MIP: luma mode index into the matrix set of the block size class (16 / 8 / 6 modes), optional transposition; the chroma
      // derived mode of a MIP CU is planar; no MRL / BDPCM; LFNST only for blocks of at least 16x16 (allowLfnstWithMip)
      if( !treeC && !cu.bdpcm[0] && !cu.multi_ref_idx && w <= 64 && h <= 64 && rng.p( P.p_mip ) )
      {
        const int sizeId = ( w == 4 && h == 4 ) ? 0 : ( w == 4 || h == 4 || ( w == 8 && h == 8 ) ) ? 1 : 2;
        cu.flags |= VVR_CU_MIP | ( rng.p( 0.5 ) ? VVR_CU_MIP_TRANSP : 0 );
        cu.intra_dir[0] = (uint8_t) rng.u( sizeId == 0 ? 16 : sizeId == 1 ? 8 : 6 );
        if( cu.intra_dir[1] < 67 && !cu.bdpcm[1] ) cu.intra_dir[1] = 0;
        cu.lfnst_intra_mode = 0;
      }
      // intra sub-partitions: horizontal (1) or vertical (2) split of the luma block in four (CU::canUseISP: more than 16 samples,
      // at most the maximum transform size); LFNST only while the partitions are at least 4x4 (CU::canUseLfnstWithISP)
      if( P.p_isp > 0 && !treeC && w * h > 16 && ( ( w >= 8 && h >= 8 ) || P.dual_tree >= 3.0f || P.min_cu_log2 == 2 ) && !( treeL && w == 64 && h == 64 ) && !cu.bdpcm[0] && !cu.multi_ref_idx && !( cu.flags & VVR_CU_MIP ) && w <= 64 && h <= 64 && rng.p( P.p_isp ) ) cu.isp_mode = (uint8_t) ( 1 + rng.u( 2 ) );
      const int ispParts = ( ( w == 4 && h == 8 ) || ( w == 8 && h == 4 ) ) ? 2 : 4;      // 4x8 / 8x4 CUs are split in two
      const bool ispNoLfnst = cu.isp_mode && ( cu.isp_mode == 1 ? h / ispParts < 4 : w / ispParts < 4 );
      // LFNST index (luma of single-tree CUs): needs DCT2 and a residual confined to the first 8/16 scan positions, see genLevels
      if( !treeC && ( P.tool_flags & VVR_TOOL_LFNST ) && !cu.bdpcm[0] && !cu.bdpcm[1] && ( !( cu.flags & VVR_CU_MIP ) || ( w >= 16 && h >= 16 ) ) && !ispNoLfnst && rng.p( P.p_lfnst ) ) cu.lfnst_idx = (uint8_t) ( 1 + rng.u( 2 ) );
    }
    else
    {
      const bool canBi = P.slice_type == 0 && P.num_ref[1] > 0 && ( w + h > 12 );
      const bool bi = canBi && rng.p( P.p_bi );
      const int list = bi ? 2 : ( P.slice_type == 0 && P.num_ref[1] > 0 && rng.p( 0.5 ) ) ? 1 : 0;
      cu.inter_dir = bi ? 3 : ( list + 1 );
      for( int l = 0; l < 2; l++ )
      {
        if( !( cu.inter_dir & ( 1 << l ) ) ) continue;
        cu.ref_idx[l] = (int8_t) rng.u( P.num_ref[l] );
        int32_t mv[2] = { rng.laplace( P.mv_sigma * 16 / 1.414 ), rng.laplace( P.mv_sigma * 16 / 1.414 ) };
        const int r = rng.u( 100 );
        if( r < 15 ) { mv[0] &= ~15; mv[1] &= ~15; }          // integer-pel
        else if( r < 25 ) mv[0] &= ~15;
        else if( r < 35 ) mv[1] &= ~15;
        if( rng.p( P.p_imv_hpel ) ) { cu.imv = 3; mv[0] &= ~7; mv[1] &= ~7; }
        wrapBias( mv, x, w );
        clipMv( mv, x, y );
        cu.mv[l][0][0] = mv[0]; cu.mv[l][0][1] = mv[1];
      }
      if( cu.imv == 3 ) for( int l = 0; l < 2; l++ ) { cu.mv[l][0][0] &= ~7; cu.mv[l][0][1] &= ~7; }
      cu.flags |= rng.p( 0.5 ) ? VVR_CU_MERGE : 0;
      // BCW: bi-prediction with CU-level weights {-2,3,5,10}/8 instead of 4/8 (index into g_BcwWeights, 2 = equal weights)
      if( bi && w * h >= 256 && rng.p( P.p_bcw ) ) { static const uint8_t idx[4] = { 0, 1, 3, 4 }; cu.bcw_idx = idx[rng.u( 4 )]; }
      // affine: control-point MVs = the translational MV plus small corner deltas (affine never uses the half-pel AMVR filter)
      if( w >= 8 && h >= 8 && rng.p( P.p_affine ) )
      {
        cu.flags |= VVR_CU_AFFINE | ( rng.p( 0.5 ) ? VVR_CU_AFFINE_6P : 0 );
        cu.imv = 0;
        const int spread = rng.p( 0.1 ) ? 200 : 24;           // a few with a large spread: the fallback to one MV (isSubblockVectorSpreadOverLimit)
        for( int l = 0; l < 2; l++ )
        {
          if( cu.ref_idx[l] < 0 ) continue;
          for( int k = 1; k < 3; k++ ) { cu.mv[l][k][0] = cu.mv[l][0][0] + rng.laplace( spread ); cu.mv[l][k][1] = cu.mv[l][0][1] + rng.laplace( spread ); }
          if( rng.p( 0.05 ) ) for( int k = 1; k < 3; k++ ) { cu.mv[l][k][0] = cu.mv[l][0][0]; cu.mv[l][k][1] = cu.mv[l][0][1]; }    // all equal: PROF off
        }
      }
      // branch taken by InterPrediction::motionCompensation (InterPrediction.cpp:1372-1459)
      bool identical = false;
      if( bi && P.ref_poc[0][cu.ref_idx[0]] == P.ref_poc[1][cu.ref_idx[1]] && cu.mv[0][0][0] == cu.mv[1][0][0] && cu.mv[0][0][1] == cu.mv[1][0][1] ) identical = true;
      if( wpOn ) identical = false;                       // xCheckIdenticalMotion (:408): never with weighted bi-prediction
      // PU::isBiPredFromDifferentDirEqDistPoc (UnitTools.cpp:3094): one reference before, one after, same POC distance
      bool eqDist = false;
      if( bi ) { const int d0 = P.poc - P.ref_poc[0][cu.ref_idx[0]], d1 = P.poc - P.ref_poc[1][cu.ref_idx[1]]; eqDist = d0 * d1 < 0 && d0 == -d1; }
      const bool sizeOk = w >= 8 && h >= 8 && w * h >= 128;
      // geometric partitioning (merge mode of B slices, 8..64, aspect ratio < 8): two uni-predictions blended along a line
      if( !( cu.flags & VVR_CU_AFFINE ) && P.slice_type == 0 && P.num_ref[1] > 0 && w >= 8 && h >= 8 && w <= 64 && h <= 64 && w < 8 * h && h < 8 * w && rng.p( P.p_geo ) )
      {
        cu.flags |= VVR_CU_GEO | VVR_CU_MERGE;
        cu.imv = 0; cu.bcw_idx = 2;
        cu.geo_split_dir = (uint8_t) rng.u( 64 );
        for( int k = 0; k < 2; k++ )
        {
          const int l = rng.u( 2 ), r = rng.u( P.num_ref[l] );
          cu.geo_dir_ref[k] = (uint8_t) ( ( ( l + 1 ) << 4 ) | r );
          int32_t mv[2] = { rng.laplace( P.mv_sigma * 16 / 1.414 ), rng.laplace( P.mv_sigma * 16 / 1.414 ) };
          if( rng.p( 0.2 ) ) { mv[0] &= ~15; mv[1] &= ~15; }
          wrapBias( mv, x, w );
          clipMv( mv, x, y );
          cu.geo_mv[k][0] = mv[0]; cu.geo_mv[k][1] = mv[1];
        }
        // what the parser leaves in the CU / motion field for later stages: the stored motion is per 4x4 (GPM motion storage); for
        // this generator the field simply carries partition 0 (uni) -- it only feeds the edge-parameter derivation, which is an input
        const int l0g = ( cu.geo_dir_ref[0] >> 4 ) - 1;
        cu.inter_dir = (uint8_t) ( l0g + 1 );
        cu.ref_idx[0] = cu.ref_idx[1] = -1; cu.ref_idx[l0g] = (int8_t) ( cu.geo_dir_ref[0] & 15 );
        memset( cu.mv, 0, sizeof( cu.mv ) );
        cu.mv[l0g][0][0] = cu.geo_mv[0][0]; cu.mv[l0g][0][1] = cu.geo_mv[0][1];
      }
      // SbTMVP: merge CU whose 8x8 sub-blocks carry their own motion (filled into the motion field below)
      if( !( cu.flags & ( VVR_CU_AFFINE | VVR_CU_GEO ) ) && w >= 8 && h >= 8 && rng.p( P.p_sbtmvp ) ) { cu.flags |= VVR_CU_SBTMVP | VVR_CU_MERGE; cu.imv = 0; cu.bcw_idx = 2; }
      // CIIP: regular merge CU (no affine/GPM/MMVD), 64 <= area, sides < 128 (8..64 here); the intra part is planar
      if( !( cu.flags & ( VVR_CU_AFFINE | VVR_CU_GEO | VVR_CU_SBTMVP ) ) && w * h >= 64 && w <= 64 && h <= 64 && rng.p( P.p_ciip ) )
      {
        cu.flags |= VVR_CU_CIIP | VVR_CU_MERGE;
        cu.imv = 0; cu.bcw_idx = 2;
        cu.intra_dir[0] = cu.intra_dir[1] = 0;
        // neighbours the blend weights look at: the CU left of the bottom-left sample and the CU above the top-right sample
        auto isIntraAt = [&]( int px, int py ) { if( px < 0 || py < 0 || !sameSliceTile( ctuOfPos( px, py ), ctuOfPos( x, y ) ) ) return false; const int32_t k = cuOf4[( py >> 2 ) * w4 + ( px >> 2 )]; return k >= 0 && B.cu[k].pred_mode == VVR_PRED_INTRA; };
        cu.ciip_neigh_intra = (uint8_t) ( ( isIntraAt( x - 1, y + h - 1 ) ? 1 : 0 ) | ( isIntraAt( x + w - 1, y - 1 ) ? 2 : 0 ) );
      }
      const bool aff = ( cu.flags & VVR_CU_AFFINE ) != 0;
      const bool wpAny = bi && ( wpPresent( 0, cu.ref_idx[0] ) || wpPresent( 1, cu.ref_idx[1] ) );       // BDOF / DMVR only with default weights (:1420, UnitTools.cpp:1297-1302)
      const bool anyScaled = ( cu.ref_idx[0] >= 0 && ( ( P.scaled_refs[0] >> cu.ref_idx[0] ) & 1 ) ) || ( cu.ref_idx[1] >= 0 && ( ( P.scaled_refs[1] >> cu.ref_idx[1] ) & 1 ) );
      const bool bio = !anyScaled && ( P.tool_flags & VVR_TOOL_BDOF ) && !wpAny && bi && eqDist && sizeOk && cu.bcw_idx == 2 && !aff && !( cu.flags & ( VVR_CU_CIIP | VVR_CU_SBTMVP ) );      // (:1407-1427), no SMVD/WP here
      const bool dmvr = !anyScaled && ( P.tool_flags & VVR_TOOL_DMVR ) && !wpAny && ( cu.flags & VVR_CU_MERGE ) && bi && eqDist && sizeOk && cu.bcw_idx == 2 && !aff && !( cu.flags & ( VVR_CU_CIIP | VVR_CU_SBTMVP ) );   // PU::checkDMVRCondition (UnitTools.cpp:1277)
      // xCheckIdenticalMotion (:404) is false for affine CUs: they go through xPredInterBi -> xPredAffineBlk per list
      // affine CUs with the same reference picture and the same control points in both lists take the uni-directional path (:424-429)
      if( aff && bi && P.ref_poc[0][cu.ref_idx[0]] == P.ref_poc[1][cu.ref_idx[1]] && rng.p( 0.3 ) ) memcpy( cu.mv[1], cu.mv[0], sizeof( cu.mv[0] ) );
      cu.mc_mode = ( cu.flags & VVR_CU_SBTMVP ) ? VVR_MC_SBTMVP : ( cu.flags & VVR_CU_GEO ) ? VVR_MC_GEO : aff ? VVR_MC_AFFINE : dmvr ? ( bio ? VVR_MC_DMVR_BDOF : VVR_MC_DMVR ) : bio ? VVR_MC_BDOF : ( !bi || identical ) ? VVR_MC_UNI : VVR_MC_BI;
      if( cu.mc_mode == VVR_MC_DMVR || cu.mc_mode == VVR_MC_DMVR_BDOF ) { cu.dmvr_off = B.num_dmvr; B.num_dmvr += ( ( w + 15 ) / 16 ) * ( ( h + 15 ) / 16 ); }     // one delta MV per 16x16 sub-block (m_dmvrMvCache)
    }
    // sub-block transform (cu_sbt_flag): the CU is split in two TUs, half/half or quarter/three quarters, and only one carries a residual
    int sbtIdx = 0, sbtPos = 0;
    if( !intra && !ibc && P.p_sbt > 0 && !( cu.flags & VVR_CU_CIIP ) && w <= 64 && h <= 64 && rng.p( P.p_sbt ) )
    {
      int cand[4], n = 0;
      if( w >= 8 ) cand[n++] = 1;      // SBT_VER_HALF
      if( h >= 8 ) cand[n++] = 2;      // SBT_HOR_HALF
      if( w >= 16 ) cand[n++] = 3;     // SBT_VER_QUAD
      if( h >= 16 ) cand[n++] = 4;     // SBT_HOR_QUAD
      sbtIdx = cand[rng.u( n )]; sbtPos = rng.u( 2 );
      cu.sbt_info = (uint8_t) ( sbtIdx | ( sbtPos << 4 ) );
    }
    // transform units: split at 64 (max TB size), cbf per block
    cu.first_tu = B.num_tu;
    bool rootCbf = false;
    const int tw = std::min( w, 64 ), th = std::min( h, 64 );
    struct Tb { int x, y, w, h; bool resi; } tbs[4]; int ntb = 0;
    if( sbtIdx )
    {
      const bool ver = sbtIdx == 1 || sbtIdx == 3, quad = sbtIdx >= 3;
      const int full = ver ? w : h, a = quad ? ( sbtPos == 0 ? full / 4 : 3 * full / 4 ) : full / 2;
      tbs[0] = ver ? Tb{ 0, 0, a, h, sbtPos == 0 } : Tb{ 0, 0, w, a, sbtPos == 0 };
      tbs[1] = ver ? Tb{ a, 0, w - a, h, sbtPos == 1 } : Tb{ 0, a, w, h - a, sbtPos == 1 };
      ntb = 2;
    }
    else if( cu.isp_mode )
    {
      const int np = ( ( w == 4 && h == 8 ) || ( w == 8 && h == 4 ) ) ? 2 : 4;
      for( int k = 0; k < np; k++ ) tbs[ntb++] = cu.isp_mode == 1 ? Tb{ 0, k * ( h / np ), w, h / np, true } : Tb{ k * ( w / np ), 0, w / np, h, true };
    }
    else for( int ty = 0; ty < h; ty += th ) for( int tx = 0; tx < w; tx += tw ) tbs[ntb++] = Tb{ tx, ty, tw, th, true };
    bool ispAnyLuma = false;
    for( int ti = 0; ti < ntb; ti++ )
    {
      const int tx = tbs[ti].x, ty = tbs[ti].y, tw = tbs[ti].w, th = tbs[ti].h;
      vvr_tu& tu = B.tu[B.num_tu]; memset( &tu, 0, sizeof( tu ) );
      const uint32_t tuIdx = B.num_tu++;
      tu.x = x + tx; tu.y = y + ty; tu.w = tw; tu.h = th; tu.cu = cuIdx;
      tu.comp_mask = treeL ? 1 : treeC ? 6 : P.chroma_format ? 7 : 1;
      if( cu.isp_mode && ti != ntb - 1 ) tu.comp_mask = 1;          // the (unsplit) chroma blocks of an ISP CU belong to the last TU
      const int qpBd = 6 * ( bd - 8 );
      tu.qp[0] = (int8_t) ( cu.qp + qpBd );
      tu.qp[1] = tu.qp[2] = (int8_t) ( std::min( 63, std::max( -qpBd, (int) cu.qp ) ) + qpBd );   // identity chroma QP mapping, zero offsets
      // joint coding of the chroma residuals (tu_joint_cbcr_residual_flag): one coded block, mode = ( cbfCb << 1 ) | cbfCr,
      // levels in Cb for modes 2 / 3 and in Cr for mode 1 (TrQuant::invTransformICT, TrQuant.cpp:320)
      int jccr = 0;
      if( sbtIdx && !tbs[ti].resi ) goto tu_done;       // the other part of an SBT CU: no residual at all
      if( P.chroma_format && P.p_jccr > 0 && rng.p( P.p_coded_chroma ) && rng.p( P.p_jccr ) ) jccr = 1 + rng.u( 3 );
      tu.joint_cbcr = (uint8_t) jccr;
      for( int c = 0; c < ( P.chroma_format ? 3 : 1 ); c++ )
      {
        if( !( tu.comp_mask & ( 1 << c ) ) ) continue;
        const int bw = c ? ( cu.isp_mode ? w >> 1 : tw >> 1 ) : tw, bh = c ? ( cu.isp_mode ? h >> 1 : th >> 1 ) : th;
        const bool ispLast = c == 0 && cu.isp_mode && ti == ntb - 1 && !ispAnyLuma;      // at least one partition is coded (the last cbf is inferred)
        const bool force = ispLast || ( c == 0 && ( ( intra && ( cu.bdpcm[0] || cu.lfnst_idx ) ) || ( cu.flags & VVR_CU_CIIP ) || sbtIdx ) );     // these modes are only signalled with a coded luma block (CIIP: merge, never skip => cu_coded_flag = 1)
        if( c && jccr )
        {
          if( ( jccr >> ( 2 - c ) ) & 1 ) tu.cbf |= 1 << c;
          if( c != ( ( jccr >> 1 ) ? 1 : 2 ) ) continue;                              // only the coded component carries levels
        }
        else
        {
          if( !force && !rng.p( c ? P.p_coded_chroma : P.p_coded ) ) continue;
          tu.cbf |= 1 << c;
        }
        if( c == 0 && cu.isp_mode ) ispAnyLuma = true;
        bool ts = bw <= 32 && bh <= 32 && !sbtIdx && !( c == 0 && cu.isp_mode ) && rng.p( P.p_ts );
        if( c == 0 && intra && cu.bdpcm[0] ) ts = true;
        if( c > 0 && intra && cu.bdpcm[1] ) ts = true;
        if( ( c == 0 || treeC ) && intra && cu.lfnst_idx ) ts = false;
        tu.mts_idx[c] = ts ? VVR_MTS_SKIP : VVR_MTS_DCT2;
        const bool implicitMts = intra && ( P.tool_flags & VVR_TOOL_IMPLICIT_MTS );
        if( !ts && c == 0 && bw <= 32 && bh <= 32 && !sbtIdx && !cu.isp_mode && !implicitMts && !ibc && !( intra && cu.lfnst_idx ) && rng.p( P.p_mts ) ) tu.mts_idx[c] = (uint8_t) ( 2 + rng.u( 4 ) );
        // getTrTypes (TrQuant.cpp:330): explicit MTS -> hor = (idx-2)&1 ? DCT8 : DST7 ; ver = (idx-2)>>1 ? DCT8 : DST7
        int hor = 0, ver = 0;
        if( tu.mts_idx[c] > 1 ) { hor = ( ( tu.mts_idx[c] - 2 ) & 1 ) ? 1 : 2; ver = ( ( tu.mts_idx[c] - 2 ) >> 1 ) ? 1 : 2; }
        if( implicitMts && c == 0 && !ts && !cu.isp_mode && !cu.lfnst_idx && !( cu.flags & VVR_CU_MIP ) )
        {   // implicit MTS (getTrTypes, TrQuant.cpp:336,349-360): DST-7 along every dimension of 4..16 samples
          hor = ( bw >= 4 && bw <= 16 ) ? 2 : 0; ver = ( bh >= 4 && bh <= 16 ) ? 2 : 0;
        }
        if( cu.isp_mode && c == 0 && !cu.lfnst_idx )
        {   // ISP: DST-7 along every dimension of 4..16 samples, DCT-2 otherwise (getTrTypes, TrQuant.cpp:349-360)
          hor = ( bw >= 4 && bw <= 16 ) ? 2 : 0; ver = ( bh >= 4 && bh <= 16 ) ? 2 : 0;
        }
        if( sbtIdx && c == 0 )
        {   // the transform pair follows from the position of the residual part (getTrTypes, TrQuant.cpp:366-398); 1 = DCT8, 2 = DST7
          if( sbtIdx == 1 || sbtIdx == 3 ) { if( bh > 32 ) hor = ver = 0; else { hor = sbtPos == 0 ? 1 : 2; ver = 2; } }
          else                             { if( bw > 32 ) hor = ver = 0; else { hor = 2; ver = sbtPos == 0 ? 1 : 2; } }
        }
        tu.tr_type[c] = (uint8_t) ( ( ver << 2 ) | hor );
        genLevels( tu, c, bw, bh, ts, intra && ( c == 0 ? cu.bdpcm[0] : cu.bdpcm[1] ), ( c == 0 || treeC ) && intra && cu.lfnst_idx );
        rootCbf = true;
      }
      tu_done:
      for( int yy = 0; yy < th; yy += 4 ) for( int xx = 0; xx < tw; xx += 4 )
        if( tu.x + xx < W && tu.y + yy < H ) ( treeC ? tuOf4C : tuOf4 )[( ( tu.y + yy ) >> 2 ) * w4 + ( ( tu.x + xx ) >> 2 )] = (int32_t) tuIdx;
    }
    cu.num_tu = B.num_tu - cu.first_tu;
    if( rootCbf ) cu.flags |= VVR_CU_ROOT_CBF;
    if( !intra && !rootCbf && ( cu.flags & VVR_CU_MERGE ) ) cu.flags |= VVR_CU_SKIP;
    // motion field + CU map
    for( int yy = 0; yy < h; yy += 4 ) for( int xx = 0; xx < w; xx += 4 )
    {
      const int i4 = ( ( y + yy ) >> 2 ) * w4 + ( ( x + xx ) >> 2 );
      if( treeC ) { cuOf4C[i4] = (int32_t) cuIdx; continue; }
      cuOf4[i4] = (int32_t) cuIdx;
      vvr_motion& m = B.motion[i4];
      m.ref_idx[0] = cu.ref_idx[0]; m.ref_idx[1] = cu.ref_idx[1];
      for( int l = 0; l < 2; l++ ) { m.mv[l][0] = cu.ref_idx[l] >= 0 ? cu.mv[l][0][0] : 0; m.mv[l][1] = cu.ref_idx[l] >= 0 ? cu.mv[l][0][1] : 0; }
      if( ibc ) { m.mv[0][0] = cu.mv[0][0][0]; m.mv[0][1] = cu.mv[0][0][1]; }      // block vector, no reference index (PU::spanMotionInfo, UnitTools.cpp:3018)
    }
    if( ibcOn ) ibcTrack( cu );
    if( cu.flags & VVR_CU_AFFINE ) for( int l = 0; l < 2; l++ ) if( cu.ref_idx[l] >= 0 ) setAllAffineMv( cu, l );
    if( cu.flags & VVR_CU_SBTMVP )
    {
      // per 8x8: uni or bi, own reference indices and MVs; neighbours often share their motion (the reference joins those)
      vvr_motion prev; memset( &prev, 0, sizeof( prev ) ); bool havePrev = false;
      for( int yy = 0; yy < h; yy += 8 ) for( int xx = 0; xx < w; xx += 8 )
      {
        vvr_motion m; memset( &m, 0, sizeof( m ) ); m.ref_idx[0] = m.ref_idx[1] = -1;
        if( havePrev && rng.p( 0.35 ) ) m = prev;
        else
        {
          const int dir = ( P.slice_type == 0 && P.num_ref[1] > 0 ) ? 1 + rng.u( 3 ) : 1;
          for( int l = 0; l < 2; l++ )
          {
            if( !( dir & ( 1 << l ) ) ) continue;
            m.ref_idx[l] = (int8_t) rng.u( P.num_ref[l] );
            int32_t mv[2] = { cu.mv[cu.ref_idx[0] >= 0 ? 0 : 1][0][0] + rng.laplace( 24 ), cu.mv[cu.ref_idx[0] >= 0 ? 0 : 1][0][1] + rng.laplace( 24 ) };
            clipMv( mv, x + xx, y + yy );
            m.mv[l][0] = mv[0]; m.mv[l][1] = mv[1];
          }
          if( dir == 3 && P.ref_poc[0][m.ref_idx[0]] == P.ref_poc[1][m.ref_idx[1]] && rng.p( 0.3 ) ) { m.mv[1][0] = m.mv[0][0]; m.mv[1][1] = m.mv[0][1]; }   // identical motion
        }
        prev = m; havePrev = true;
        for( int sy = 0; sy < 8; sy += 4 ) for( int sx = 0; sx < 8; sx += 4 ) B.motion[(size_t) ( ( y + yy + sy ) >> 2 ) * w4 + ( ( x + xx + sx ) >> 2 )] = m;
      }
    }
  }

  // PU::setAllAffineMv (UnitTools.cpp:2689): per-4x4 sub-block MVs from the control points, or one fallback MV when the
  // sub-block vectors spread too far (InterPrediction::isSubblockVectorSpreadOverLimit, InterPrediction.cpp:892)
  static bool spreadOverLimit( int a, int b, int c, int d, int predType )
  {
    const int s4 = 4 << 11, filterTap = 6;
    if( predType == 3 )
    {
      int rw = std::max( std::max( 0, 4 * a + s4 ), std::max( 4 * c, 4 * a + 4 * c + s4 ) ) - std::min( std::min( 0, 4 * a + s4 ), std::min( 4 * c, 4 * a + 4 * c + s4 ) );
      int rh = std::max( std::max( 0, 4 * b ), std::max( 4 * d + s4, 4 * b + 4 * d + s4 ) ) - std::min( std::min( 0, 4 * b ), std::min( 4 * d + s4, 4 * b + 4 * d + s4 ) );
      rw = ( rw >> 11 ) + filterTap + 3; rh = ( rh >> 11 ) + filterTap + 3;
      return rw * rh > ( filterTap + 9 ) * ( filterTap + 9 );
    }
    int rw = std::max( 0, 4 * a + s4 ) - std::min( 0, 4 * a + s4 ), rh = std::max( 0, 4 * b ) - std::min( 0, 4 * b );
    rw = ( rw >> 11 ) + filterTap + 3; rh = ( rh >> 11 ) + filterTap + 3;
    if( rw * rh > ( filterTap + 9 ) * ( filterTap + 5 ) ) return true;
    rw = std::max( 0, 4 * c ) - std::min( 0, 4 * c ); rh = std::max( 0, 4 * d + s4 ) - std::min( 0, 4 * d + s4 );
    rw = ( rw >> 11 ) + filterTap + 3; rh = ( rh >> 11 ) + filterTap + 3;
    return rw * rh > ( filterTap + 5 ) * ( filterTap + 9 );
  }
  static void roundAffineMv( int& mx, int& my, int sh ) { const int o = 1 << ( sh - 1 ); mx = ( mx + o - ( mx >= 0 ) ) >> sh; my = ( my + o - ( my >= 0 ) ) >> sh; }
  static int ilog2( int v ) { int l = 0; while( ( 1 << l ) < v ) l++; return l; }
  void setAllAffineMv( const vvr_cu& cu, int l )
  {
    const int shift = 7;
    const int dHX = ( cu.mv[l][1][0] - cu.mv[l][0][0] ) * ( 1 << ( shift - ilog2( cu.w ) ) ), dHY = ( cu.mv[l][1][1] - cu.mv[l][0][1] ) * ( 1 << ( shift - ilog2( cu.w ) ) );
    int dVX, dVY;
    if( cu.flags & VVR_CU_AFFINE_6P ) { dVX = ( cu.mv[l][2][0] - cu.mv[l][0][0] ) * ( 1 << ( shift - ilog2( cu.h ) ) ); dVY = ( cu.mv[l][2][1] - cu.mv[l][0][1] ) * ( 1 << ( shift - ilog2( cu.h ) ) ); }
    else { dVX = -dHY; dVY = dHX; }
    const int baseX = cu.mv[l][0][0] * ( 1 << shift ), baseY = cu.mv[l][0][1] * ( 1 << shift );
    const bool over = spreadOverLimit( dHX, dHY, dVX, dVY, cu.inter_dir );
    int fx = 0, fy = 0;
    if( over ) { fx = baseX + dHX * ( cu.w >> 1 ) + dVX * ( cu.h >> 1 ); fy = baseY + dHY * ( cu.w >> 1 ) + dVY * ( cu.h >> 1 ); roundAffineMv( fx, fy, shift ); fx = std::min( ( 1 << 17 ) - 1, std::max( -( 1 << 17 ), fx ) ); fy = std::min( ( 1 << 17 ) - 1, std::max( -( 1 << 17 ), fy ) ); }
    for( int hy = 0; hy < cu.h / 4; hy++ ) for( int wx = 0; wx < cu.w / 4; wx++ )
    {
      int mx = fx, my = fy;
      if( !over )
      {
        mx = baseX + dHX * ( 2 + 4 * wx ) + dVX * ( 2 + 4 * hy ); my = baseY + dHY * ( 2 + 4 * wx ) + dVY * ( 2 + 4 * hy );
        roundAffineMv( mx, my, shift );
        mx = std::min( ( 1 << 17 ) - 1, std::max( -( 1 << 17 ), mx ) ); my = std::min( ( 1 << 17 ) - 1, std::max( -( 1 << 17 ), my ) );
      }
      vvr_motion& m = B.motion[(size_t) ( cu.y / 4 + hy ) * w4 + cu.x / 4 + wx];
      m.mv[l][0] = mx; m.mv[l][1] = my;
    }
  }

  void split( int x, int y, int w, int h )
  {
    if( x >= W || y >= H ) return;
    // smallest CU side: 4 in the luma tree of dual-tree pictures (dual_tree >= 2) and, with min_cu_log2 = 2, everywhere a 4-wide CU is
    // legal (luma-tree CUs of a local dual tree, inter-only sub-trees); chroma-tree nodes stay at 8 luma samples (4 chroma)
    const int minS = curTree == VVR_TREE_CHROMA ? 8 : ( curTree == VVR_TREE_LUMA && P.dual_tree >= 2.0f ) ? 4 : 1 << P.min_cu_log2;
    const bool crossX = x + w > W, crossY = y + h > H;
    if( crossX || crossY )
    {
      const int minS = std::max( 8, 1 << P.min_cu_log2 );       // (implicit splits at the picture boundary stop at 8x8: picture sizes are multiples of 8)
      // implicit boundary split: quad when possible, else binary in the crossing direction
      if( w > minS && h > minS && ( ( crossX && crossY ) || w == h ) ) { const int hw = w >> 1, hh = h >> 1; split( x, y, hw, hh ); split( x + hw, y, hw, hh ); split( x, y + hh, hw, hh ); split( x + hw, y + hh, hw, hh ); }
      else if( w <= 64 && h > 64 ) { split( x, y, w, h >> 1 ); split( x, y + ( h >> 1 ), w, h >> 1 ); }                 // (VPDU rule first)
      else if( w > 64 && h <= 64 ) { split( x, y, w >> 1, h ); split( x + ( w >> 1 ), y, w >> 1, h ); }
      else if( crossX && w > minS ) { split( x, y, w >> 1, h ); split( x + ( w >> 1 ), y, w >> 1, h ); }
      else if( crossY && h > minS ) { split( x, y, w, h >> 1 ); split( x, y + ( h >> 1 ), w, h >> 1 ); }
      return;
    }
    const int s = std::max( w, h );
    double ps = s >= 128 ? 0.92 : s >= 64 ? 0.70 : s >= 32 ? 0.45 : s >= 16 ? 0.25 : s >= 8 && minS < 8 ? 0.2 : 0.0;
    ps = std::min( 0.98, ps * P.p_split_scale );
    if( w != h && std::min( w, h ) <= minS ) ps *= 0.5;
    if( curI && s > 64 ) ps = 1.0;   // intra pictures / slices: CUs <= 64
    enum { S_NONE, S_QT, S_BV, S_BH, S_TV, S_TH };
    int type = S_NONE;
    if( rng.p( ps ) )
    {
      const int r = rng.u( 100 );
      // VPDU rules of the standard (every 64x64 region is decoded completely before the next one, or a CU covers whole regions):
      // no vertical binary split of a block that is at most 64 wide but taller than 64, no horizontal one of a block wider than 64 but
      // at most 64 tall, ternary splits only inside 64x64.  Inter-only sub-trees (SCIPU) keep at least 32 luma samples per CU.
      const bool interOnly = modeType == 1;
      const bool canQ = w == h && w > minS && !( interOnly && w * h <= 64 ), canH = h > minS && !( w > 64 && h <= 64 ) && !( interOnly && w * h <= 32 ), canV = w > minS && !( w <= 64 && h > 64 ) && !( interOnly && w * h <= 32 );
      const bool canTH = h >= 4 * minS && h <= 64 && w <= 64 && !( interOnly && w * h <= 64 ), canTV = w >= 4 * minS && w <= 64 && h <= 64 && !( interOnly && w * h <= 64 );
      if( canQ && r < 50 ) type = S_QT;
      else if( canV && ( r < 68 || !canH ) ) type = S_BV;
      else if( canH && r < 86 ) type = S_BH;
      else if( canTV && r < 93 ) type = S_TV;
      else if( canTH ) type = S_TH;
      else if( canH ) type = S_BH;
      else if( canV ) type = S_BV;
      else if( canQ ) type = S_QT;
    }
    if( type == S_NONE ) { addCu( x, y, w, h ); return; }
    // smallest chroma intra prediction unit (mode_constraint, modeTypeCondition of the coding_tree semantics): a split of a single-tree
    // node that would create chroma blocks of fewer than 16 samples or 2-wide intra chroma blocks makes the sub-tree either inter-only
    // (chroma follows the split) or intra-only with a LOCAL DUAL TREE: the luma children are luma-tree CUs, then one chroma-tree CU
    // covers the node
    int cond = 0;
    if( curTree == VVR_TREE_JOINT && modeType == 0 && P.chroma_format )
    {
      const int area = w * h; const bool bt = type == S_BV || type == S_BH, tt = type == S_TV || type == S_TH;
      if( ( area == 64 && ( type == S_QT || tt ) ) || ( area == 32 && bt ) ) cond = 1;
      else if( ( area == 64 && bt ) || ( area == 128 && tt ) || ( w == 8 && type == S_BV ) || ( w == 16 && type == S_TV ) ) cond = 1 + ( !curI ? 1 : 0 );
    }
    const int oldMode = modeType, oldTree = curTree;
    bool localDual = false;
    if( cond == 1 ) { modeType = 2; localDual = true; }
    else if( cond == 2 ) { if( rng.p( 0.5 ) ) { modeType = 2; localDual = true; } else modeType = 1; }
    if( localDual ) curTree = VVR_TREE_LUMA;
    switch( type )
    {
      case S_QT: { const int hw = w >> 1, hh = h >> 1; split( x, y, hw, hh ); split( x + hw, y, hw, hh ); split( x, y + hh, hw, hh ); split( x + hw, y + hh, hw, hh ); break; }
      case S_BV: split( x, y, w >> 1, h ); split( x + ( w >> 1 ), y, w >> 1, h ); break;
      case S_BH: split( x, y, w, h >> 1 ); split( x, y + ( h >> 1 ), w, h >> 1 ); break;
      case S_TV: split( x, y, w >> 2, h ); split( x + ( w >> 2 ), y, w >> 1, h ); split( x + 3 * ( w >> 2 ), y, w >> 2, h ); break;
      default:   split( x, y, w, h >> 2 ); split( x, y + ( h >> 2 ), w, h >> 1 ); split( x, y + 3 * ( h >> 2 ), w, h >> 2 ); break;
    }
    if( localDual ) { curTree = VVR_TREE_CHROMA; addCu( x, y, w, h ); }
    modeType = oldMode; curTree = oldTree;
  }

  // ---------------------------------------------------------------------------------------------------------------
  // deblocking edge parameters: the table LoopFilter::calcFilterStrengthsCTU (LoopFilter.cpp:495-1360) fills.
  // ---------------------------------------------------------------------------------------------------------------
  // dual tree (intra pictures): luma edges from the luma tree, chroma edges (8x8 chroma-sample grid) from the chroma tree; every
  // edge separates intra blocks (boundary strength 2)
  void deriveLfpDual()
  {
    const int qpBd = 6 * ( bd - 8 );
    for( int d = 0; d < 2; d++ )
    {
      memset( B.lfp[d], 0, sizeof( vvr_lfp ) * (size_t) w4 * h4 );
      for( int y4 = 0; y4 < h4; y4++ ) for( int x4 = 0; x4 < w4; x4++ )
      {
        const int px4 = d == 0 ? x4 - 1 : x4, py4 = d == 0 ? y4 : y4 - 1;
        if( px4 < 0 || py4 < 0 ) continue;
        const int iq = y4 * w4 + x4, ip = py4 * w4 + px4;
        vvr_lfp& L = B.lfp[d][iq];
        int bsY = 0, bsC = 0;
        if( tuOf4[iq] != tuOf4[ip] )
        {
          const vvr_tu& TQ = B.tu[tuOf4[iq]]; const vvr_tu& TP = B.tu[tuOf4[ip]];
          const int sizeQ = d == 0 ? TQ.w : TQ.h, sizeP = d == 0 ? TP.w : TP.h;
          int lenP, lenQ;
          if( sizeP <= 4 || sizeQ <= 4 ) lenP = lenQ = 1;
          else { lenP = sizeP >= 32 ? 7 : 3; lenQ = sizeQ >= 32 ? 7 : 3; }
          L.side_max_filt_length = (uint8_t) ( 0x80 | ( lenP << 4 ) | lenQ );
          L.flags |= 1; bsY = ( B.cu[TQ.cu].bdpcm[0] && B.cu[TP.cu].bdpcm[0] ) ? 0 : 2;      // no filtering between two BDPCM blocks (LoopFilter.cpp:1146)
          if( B.cu[TQ.cu].pred_mode == VVR_PRED_IBC && B.cu[TP.cu].pred_mode == VVR_PRED_IBC )
          {   // two IBC blocks: coded residual, else the block vectors (the current picture is the reference of both, LoopFilter.cpp:1346-1360)
            const vvr_motion& mq = B.motion[iq]; const vvr_motion& mp = B.motion[ip];
            bsY = ( ( TQ.cbf & 1 ) || ( TP.cbf & 1 ) ) ? 1 : ( TQ.cu != TP.cu && ( std::abs( mq.mv[0][0] - mp.mv[0][0] ) >= 8 || std::abs( mq.mv[0][1] - mp.mv[0][1] ) >= 8 ) ) ? 1 : 0;
          }
          L.qp[0] = (int8_t) ( ( B.cu[TQ.cu].qp + B.cu[TP.cu].qp + 1 ) >> 1 );
        }
        const int posAlong = d == 0 ? ( x4 << 2 ) : ( y4 << 2 );
        if( posAlong % 16 == 0 && tuOf4C[iq] != tuOf4C[ip] )
        {
          const vvr_tu& TQ = B.tu[tuOf4C[iq]]; const vvr_tu& TP = B.tu[tuOf4C[ip]];
          const int sizeQ = ( d == 0 ? TQ.w : TQ.h ) >> 1, sizeP = ( d == 0 ? TP.w : TP.h ) >> 1;
          L.flags |= 2 | ( ( sizeP >= 8 && sizeQ >= 8 ) ? 0x20 : 0 ); bsC = ( B.cu[TQ.cu].bdpcm[1] && B.cu[TP.cu].bdpcm[1] ) ? 0 : 2;      // no chroma filtering between two chroma BDPCM blocks (LoopFilter.cpp:1132)
          L.qp[1] = (int8_t) ( ( TQ.qp[1] + TP.qp[1] - 2 * qpBd + 1 ) >> 1 );
          L.qp[2] = (int8_t) ( ( TQ.qp[2] + TP.qp[2] - 2 * qpBd + 1 ) >> 1 );
        }
        L.bs = (uint8_t) ( bsY | ( bsC << 2 ) | ( bsC << 4 ) );
        if( !lfMayCross( ctuOfPos( x4 << 2, y4 << 2 ), ctuOfPos( px4 << 2, py4 << 2 ) ) || onVirtualBoundary( d, x4, y4 ) ) { L.bs = 0; L.flags &= (uint8_t) ~3; }
      }
    }
  }

  void deriveLfp()
  {
    const int qpBd = 6 * ( bd - 8 );
    // boundary strength from the motion of the two 4x4 units next to the edge (LoopFilter.cpp:1222-1360)
    auto motionBs = [&]( int iq, int ip ) -> int
    {
      const vvr_motion& mq = B.motion[iq]; const vvr_motion& mp = B.motion[ip];
      auto refPoc = [&]( const vvr_motion& m, int l ) { return m.ref_idx[l] >= 0 ? P.ref_poc[l][m.ref_idx[l]] : INT32_MIN; };
      const int nq = ( mq.ref_idx[0] >= 0 ) + ( mq.ref_idx[1] >= 0 ), np = ( mp.ref_idx[0] >= 0 ) + ( mp.ref_idx[1] >= 0 );
      auto far = [&]( const int32_t a[2], const int32_t b[2] ) { return std::abs( a[0] - b[0] ) >= 8 || std::abs( a[1] - b[1] ) >= 8; };
      if( nq != np ) return 1;
      if( nq == 1 )
      {
        const int lq = mq.ref_idx[0] >= 0 ? 0 : 1, lp = mp.ref_idx[0] >= 0 ? 0 : 1;
        return ( refPoc( mq, lq ) != refPoc( mp, lp ) || far( mq.mv[lq], mp.mv[lp] ) ) ? 1 : 0;
      }
      const int q0 = refPoc( mq, 0 ), q1 = refPoc( mq, 1 ), p0 = refPoc( mp, 0 ), p1 = refPoc( mp, 1 );
      if( !( ( q0 == p0 && q1 == p1 ) || ( q0 == p1 && q1 == p0 ) ) ) return 1;
      if( p0 != p1 )
      {
        if( q0 == p0 ) return ( far( mq.mv[0], mp.mv[0] ) || far( mq.mv[1], mp.mv[1] ) ) ? 1 : 0;
        return ( far( mq.mv[0], mp.mv[1] ) || far( mq.mv[1], mp.mv[0] ) ) ? 1 : 0;
      }
      return ( ( far( mq.mv[0], mp.mv[0] ) || far( mq.mv[1], mp.mv[1] ) ) && ( far( mq.mv[0], mp.mv[1] ) || far( mq.mv[1], mp.mv[0] ) ) ) ? 1 : 0;
    };
    for( int d = 0; d < 2; d++ )
    {
      memset( B.lfp[d], 0, sizeof( vvr_lfp ) * (size_t) w4 * h4 );
      for( int y4 = 0; y4 < h4; y4++ ) for( int x4 = 0; x4 < w4; x4++ )
      {
        const int px4 = d == 0 ? x4 - 1 : x4, py4 = d == 0 ? y4 : y4 - 1;
        if( px4 < 0 || py4 < 0 ) continue;                    // picture boundary: never filtered
        const int iq = y4 * w4 + x4, ip = py4 * w4 + px4;
        const int tq = tuOf4[iq], tp = tuOf4[ip];
        if( tq == tp ) continue;                               // not a transform (or CU) edge
        const vvr_tu& TQ = B.tu[tq]; const vvr_tu& TP = B.tu[tp];
        const vvr_cu& CQ = B.cu[TQ.cu]; const vvr_cu& CP = B.cu[TP.cu];
        // the blocks that own the chroma on either side: the chroma-tree CU of a local dual tree where there is one, else the same block
        const bool haveC = !tuOf4C.empty();
        const int tqc = haveC && tuOf4C[iq] >= 0 ? tuOf4C[iq] : tq, tpc = haveC && tuOf4C[ip] >= 0 ? tuOf4C[ip] : tp;
        const vvr_tu& TQc = B.tu[tqc]; const vvr_tu& TPc = B.tu[tpc];
        const vvr_cu& CQc = B.cu[TQc.cu]; const vvr_cu& CPc = B.cu[TPc.cu];
        vvr_lfp& L = B.lfp[d][iq];
        // maximum filter lengths from the transform sizes orthogonal to the edge (LoopFilter.cpp:910-922)
        const int sizeQ = d == 0 ? TQ.w : TQ.h, sizeP = d == 0 ? TP.w : TP.h;
        int lenP, lenQ;
        if( sizeP <= 4 || sizeQ <= 4 ) lenP = lenQ = 1;
        else { lenP = sizeP >= 32 ? ( ( CP.flags & ( VVR_CU_AFFINE | VVR_CU_SBTMVP ) ) ? 5 : 7 ) : 3; lenQ = sizeQ >= 32 ? 7 : 3; }      // (:911 cuP->affineFlag(): set for every sub-block merge CU, the SbTMVP ones too)
        L.side_max_filt_length = (uint8_t) ( 0x80 | ( lenP << 4 ) | lenQ );
        L.flags = 1;                                           // filterEdge luma
        // chroma edges live on the 8x8 chroma-sample grid = 16 luma samples
        const int posAlong = d == 0 ? ( x4 << 2 ) : ( y4 << 2 );
        const bool chromaEdge = P.chroma_format && ( posAlong % 16 == 0 ) && tqc != tpc && !( &CQc == &CPc && CQc.isp_mode );     // the chroma block of an ISP CU is not split
        if( chromaEdge )
        {
          // (the chroma block of an ISP CU is not split: its size is the CU's)
          const int sizeQc = ( CQc.isp_mode ? ( d == 0 ? CQc.w : CQc.h ) : ( d == 0 ? TQc.w : TQc.h ) ) >> 1, sizePc = ( CPc.isp_mode ? ( d == 0 ? CPc.w : CPc.h ) : ( d == 0 ? TPc.w : TPc.h ) ) >> 1;
          if( sizePc >= 8 && sizeQc >= 8 ) L.flags |= 0x20;   // both sides >= 8 chroma samples: long chroma filter allowed
        }
        // boundary strength (LoopFilter.cpp:1094-1360)
        int bsY = 0, bsCb = 0, bsCr = 0;
        const bool intra = CQ.pred_mode == VVR_PRED_INTRA || CP.pred_mode == VVR_PRED_INTRA || ( ( CQ.flags | CP.flags ) & VVR_CU_CIIP );   // CIIP counts as intra for the BS
        // (the chroma of a local dual tree is always intra, also next to IBC luma CUs)
        const bool intraC = CQc.pred_mode == VVR_PRED_INTRA || CPc.pred_mode == VVR_PRED_INTRA || ( ( CQc.flags | CPc.flags ) & VVR_CU_CIIP );
        if( chromaEdge )
        {
          const bool jointChr = TQc.joint_cbcr || TPc.joint_cbcr;     // (LoopFilter.cpp:1180-1184)
          if( intraC ) bsCb = bsCr = ( CQc.pred_mode == VVR_PRED_INTRA && CQc.bdpcm[1] && CPc.pred_mode == VVR_PRED_INTRA && CPc.bdpcm[1] ) ? 0 : 2;      // (LoopFilter.cpp:1132)
          else { if( ( TQc.cbf & 2 ) || ( TPc.cbf & 2 ) || jointChr ) bsCb = 1; if( ( TQc.cbf & 4 ) || ( TPc.cbf & 4 ) || jointChr ) bsCr = 1; }
        }
        if( intra ) bsY = ( CQ.bdpcm[0] && CP.bdpcm[0] ) ? 0 : 2;      // no luma filtering between two BDPCM blocks (LoopFilter.cpp:1146)
        else
        {
          if( ( TQ.cbf & 1 ) || ( TP.cbf & 1 ) ) bsY = 1;
          if( !bsY && &CQ != &CP )
          {
            // IBC: a different prediction mode on the other side always filters (:1218); two IBC blocks compare their block vectors,
            // the current picture being the reference of both (:1237-1240,1346-1360)
            if( CQ.pred_mode != CP.pred_mode ) bsY = 1;
            else if( CQ.pred_mode == VVR_PRED_IBC ) bsY = ( std::abs( B.motion[iq].mv[0][0] - B.motion[ip].mv[0][0] ) >= 8 || std::abs( B.motion[iq].mv[0][1] - B.motion[ip].mv[0][1] ) >= 8 ) ? 1 : 0;
            else bsY = motionBs( iq, ip );
          }
        }
        L.bs = (uint8_t) ( bsY | ( bsCb << 2 ) | ( bsCr << 4 ) );
        // an edge on a slice / tile boundary the loop filters must not cross is not filtered (m_stLFCUParam.leftEdge / topEdge, LoopFilter.cpp:1078,1088)
        if( !lfMayCross( ctuOfPos( x4 << 2, y4 << 2 ), ctuOfPos( px4 << 2, py4 << 2 ) ) || onVirtualBoundary( d, x4, y4 ) ) { L.bs = 0; L.flags &= (uint8_t) ~3; }
        L.qp[0] = (int8_t) ( ( CQ.qp + CP.qp + 1 ) >> 1 );
        L.qp[1] = (int8_t) ( ( TQc.qp[1] + TPc.qp[1] - 2 * qpBd + 1 ) >> 1 );
        L.qp[2] = (int8_t) ( ( TQc.qp[2] + TPc.qp[2] - 2 * qpBd + 1 ) >> 1 );
        if( !L.bs ) { /* edge without any filtering keeps its length info, like the reference table */ }
      }
      // sub-block edges of affine and SbTMVP CUs (8x8 grid; xSetEdgeFilterInsidePu :1032, xSetMaxFilterLengthPQForCodingSubBlocks :707):
      // luma only, strength from the motion of the sub-blocks, filter lengths limited by the distance to the next transform edge
      for( uint32_t ci = 0; ci < B.num_cu; ci++ )
      {
        const vvr_cu& cu = B.cu[ci];
        if( cu.pred_mode != VVR_PRED_INTER || !( cu.flags & ( VVR_CU_AFFINE | VVR_CU_SBTMVP ) ) || cu.tree == VVR_TREE_CHROMA ) continue;
        const int perp = d == 0 ? cu.w : cu.h, parl = d == 0 ? cu.h : cu.w;
        auto cell = [&]( int pp, int pl ) { const int x = d == 0 ? cu.x + pp : cu.x + pl, y = d == 0 ? cu.y + pl : cu.y + pp; return ( y >> 2 ) * w4 + ( x >> 2 ); };
        // (in a CU with sub-block edges the marker of a transform edge is only written where that edge may be filtered, LoopFilter.cpp:960-981 under bValue)
        auto isTe = [&]( int pp, int pl )
        {
          if( pp < 0 || pp >= perp || !( B.lfp[d][cell( pp, pl )].side_max_filt_length & 0x80 ) ) return false;
          const int cx = ( d == 0 ? cu.x + pp : cu.x + pl ) >> 2, cy = ( d == 0 ? cu.y + pl : cu.y + pp ) >> 2;
          if( onVirtualBoundary( d, cx, cy ) ) return false;
          return pp > 0 || lfMayCross( ctuOfPos( cx << 2, cy << 2 ), ctuOfPos( ( d == 0 ? cx - 1 : cx ) << 2, ( d == 0 ? cy : cy - 1 ) << 2 ) );
        };
        for( int pl = 0; pl < parl; pl += 4 ) for( int pp = 0; pp < perp; pp += 8 )
        {
          if( ( d == 0 ? cu.x : cu.y ) + pp == 0 ) continue;                       // picture boundary
          vvr_lfp& L = B.lfp[d][cell( pp, pl )];
          const bool onVb = onVirtualBoundary( d, ( d == 0 ? cu.x + pp : cu.x + pl ) >> 2, ( d == 0 ? cu.y + pl : cu.y + pp ) >> 2 );      // an edge on a virtual boundary is not filtered (:575-600)
          int lenP, lenQ, te = 0;
          if( L.side_max_filt_length & 0x80 )
          {
            te = 0x80; lenQ = std::min( L.side_max_filt_length & 7, 5 ); lenP = ( L.side_max_filt_length >> 4 ) & 7;
            if( pp > 0 )
            {
              lenP = std::min( lenP, 5 );
              // a transform edge inside the CU that is also a sub-block edge: without a coded block on either side the motion decides
              // (xSetEdgeFilterInsidePu :1046 turns the edge marker into 3, so xGetBoundaryStrengthSingle goes on to the motion test)
              if( onVb ) { L.flags &= (uint8_t) ~1; }       // (xSetEdgeFilterInsidePu with bValue = false, :1053)
              else if( !( L.bs & 3 ) ) L.bs = (uint8_t) ( L.bs | ( ( cu.flags & VVR_CU_CIIP ) ? 1 : motionBs( cell( pp, pl ), cell( pp - 4, pl ) ) ) );
            }
          }
          else
          {
            if( isTe( pp - 4, pl ) || pp + 4 >= perp || isTe( pp + 4, pl ) ) lenP = lenQ = 1;
            else if( pp == 8 || isTe( pp - 8, pl ) || pp + 8 >= perp || isTe( pp + 8, pl ) ) lenP = lenQ = 2;
            else lenP = lenQ = 3;
            // a pure sub-block edge: filtered where the motion of the two sub-blocks differs
            const int iq = cell( pp, pl ), ip = cell( pp - 4, pl );
            if( !onVb )
            {
              L.flags |= 1;
              L.bs = (uint8_t) ( ( cu.flags & VVR_CU_CIIP ) ? 1 : motionBs( iq, ip ) );
              L.qp[0] = cu.qp;
            }
          }
          L.side_max_filt_length = (uint8_t) ( te | ( lenP << 4 ) | lenQ );
        }
      }
    }
  }

  void genLoopFilterParams()
  {
    const int numCtu = ( ( W + ctu - 1 ) / ctu ) * ( ( H + ctu - 1 ) / ctu );
    const int maxOff = 7;   // (1 << (min(bd,10) - 5)) - 1
    for( int a = 0; a < numCtu; a++ )
    {
      vvr_sao_ctu& s = B.sao[a]; memset( &s, 0, sizeof( s ) );
      for( int c = 0; c < ( P.chroma_format ? 3 : 1 ); c++ )
      {
        const bool en = ( P.tool_flags & ( c ? VVR_TOOL_SAO_CHROMA : VVR_TOOL_SAO_LUMA ) ) && rng.p( P.p_sao );
        if( !en ) continue;
        s.mode[c] = 1;
        if( c == 2 ) { s.type[2] = s.type[1]; if( !s.mode[1] ) s.type[2] = (uint8_t) rng.u( 5 ); }   // Cb and Cr share the type when both are on
        else s.type[c] = (uint8_t) ( rng.p( 0.75 ) ? rng.u( 4 ) : 4 );
        if( s.type[c] == 4 ) { s.band_pos[c] = (uint8_t) rng.u( 32 ); for( int i = 0; i < 4; i++ ) s.offset[c][i] = (int8_t) ( (int) rng.u( 2 * maxOff + 1 ) - maxOff ); }
        else { s.offset[c][0] = (int8_t) rng.u( maxOff + 1 ); s.offset[c][1] = (int8_t) rng.u( maxOff + 1 ); s.offset[c][2] = (int8_t) -(int) rng.u( maxOff + 1 ); s.offset[c][3] = (int8_t) -(int) rng.u( maxOff + 1 ); }
      }
      vvr_alf_ctu& f = B.alf[a]; memset( &f, 0, sizeof( f ) );
      if( P.tool_flags & VVR_TOOL_ALF )
      {
        f.enable[0] = rng.p( P.p_alf_luma );
        f.luma_filter_idx = (int16_t) ( rng.p( 0.5 ) ? rng.u( 16 ) : 16 + rng.u( B.alf_params->num_luma_aps ) );
        if( P.chroma_format ) { f.enable[1] = rng.p( P.p_alf_chroma ); f.enable[2] = rng.p( P.p_alf_chroma ); f.alt[0] = (uint8_t) rng.u( VVR_ALF_MAX_CHR_ALT ); f.alt[1] = (uint8_t) rng.u( VVR_ALF_MAX_CHR_ALT ); }
        if( ( P.tool_flags & VVR_TOOL_CCALF ) && P.chroma_format ) { f.cc_idc[0] = rng.p( P.p_ccalf ) ? 1 + rng.u( VVR_CCALF_FILTERS ) : 0; f.cc_idc[1] = rng.p( P.p_ccalf ) ? 1 + rng.u( VVR_CCALF_FILTERS ) : 0; }
      }
    }
  }

  // LMCS model (lmcs_data()) + the tables Reshape::constructReshaper (Reshape.cpp:318-374) derives from it: this is host glue
  // (the parser side owns it), the back-end consumes the two LUTs
  void genLmcs()
  {
    vvr_lmcs_params& L = *B.lmcs; memset( &L, 0, sizeof( L ) );
    const int lutSize = 1 << bd, orgCW = lutSize / 16, l2cw = ilog2( orgCW );
    L.min_bin = (int16_t) rng.u( 2 ); L.max_bin = (int16_t) ( 14 + rng.u( 2 ) );
    int binCW[16] = { 0 };
    for( int i = L.min_bin; i <= L.max_bin; i++ ) binCW[i] = orgCW + (int) rng.u( 2 * ( orgCW / 3 ) + 1 ) - orgCW / 3;     // well inside [OrgCW >> 3, OrgCW << 3) and >= 1 << (bd - 5)
    for( ;; ) { int sum = 0; for( int i = 0; i < 16; i++ ) sum += binCW[i]; if( sum <= lutSize - 1 ) break; binCW[L.min_bin + rng.u( L.max_bin - L.min_bin + 1 )] -= 1; }
    for( int i = 0; i < 16; i++ ) L.model_delta_cw[i] = (int16_t) ( ( i >= L.min_bin && i <= L.max_bin ) ? binCW[i] - orgCW : 0 );
    L.model_delta_crs = (int16_t) ( (int) rng.u( 5 ) - 2 );
    int pivot[17] = { 0 }, inPivot[17] = { 0 }, fwdCoef[16], invCoef[16];
    for( int i = 0; i < 16; i++ )
    {
      pivot[i + 1] = pivot[i] + binCW[i]; inPivot[i + 1] = inPivot[i] + orgCW;
      fwdCoef[i] = ( binCW[i] * ( 1 << 11 ) + ( 1 << ( l2cw - 1 ) ) ) >> l2cw;
      if( binCW[i] == 0 ) { invCoef[i] = 0; L.chroma_scale[i] = 1 << 11; }
      else { invCoef[i] = orgCW * ( 1 << 11 ) / binCW[i]; L.chroma_scale[i] = (int16_t) ( orgCW * ( 1 << 11 ) / ( binCW[i] + L.model_delta_crs ) ); }
    }
    for( int i = 0; i < 17; i++ ) L.pivot[i] = (int16_t) pivot[i];
    auto idxInv = [&]( int v ) { int k = L.min_bin; for( ; k <= L.max_bin; k++ ) if( v < pivot[k + 1] ) break; return std::min( k, 15 ); };   // getPWLIdxInv (:280)
    for( int v = 0; v < lutSize; v++ )
    {
      const int ii = idxInv( v );
      const int inv = inPivot[ii] + ( ( invCoef[ii] * ( v - pivot[ii] ) + ( 1 << 10 ) ) >> 11 );
      L.inv_lut[v] = (int16_t) std::min( lutSize - 1, std::max( 0, (int) (int16_t) inv ) );
      const int fi = v >> l2cw;                                                                        // rspFwdCore (Buffer.cpp:321)
      L.fwd_lut[v] = (int16_t) std::min( lutSize - 1, std::max( 0, pivot[fi] + ( ( (int16_t) fwdCoef[fi] * ( v - inPivot[fi] ) + ( 1 << 10 ) ) >> 11 ) ) );
    }
  }

  void genAlfParams()
  {
    vvr_alf_params& A = *B.alf_params; memset( &A, 0, sizeof( A ) );
    static const int clipVals[3][4] = { { 256, 32, 8, 2 }, { 512, 64, 16, 4 }, { 1024, 128, 32, 8 } };   // AdaptiveLoopFilter.cpp:382
    const int* cv = clipVals[bd - 8];
    A.num_luma_aps = 2;
    for( int a = 0; a < VVR_MAX_ALF_APS; a++ ) for( int c = 0; c < VVR_ALF_CLASSES; c++ ) for( int k = 0; k < 12; k++ )
    {
      A.luma_coeff[a][c][k] = (int16_t) std::max( -128, std::min( 127, rng.laplace( k >= 9 ? 12.0 : 5.0 ) ) );
      A.luma_clip[a][c][k]  = (int16_t) cv[rng.u( 4 )];
    }
    for( int alt = 0; alt < VVR_ALF_MAX_CHR_ALT; alt++ ) for( int k = 0; k < 6; k++ )
    {
      A.chroma_coeff[alt][k] = (int16_t) std::max( -128, std::min( 127, rng.laplace( k >= 4 ? 12.0 : 5.0 ) ) );
      A.chroma_clip[alt][k]  = (int16_t) cv[rng.u( 4 )];
    }
    for( int c = 0; c < 2; c++ ) for( int f = 0; f < VVR_CCALF_FILTERS; f++ ) for( int k = 0; k < 7; k++ )
    {
      const int e = rng.u( 8 );          // CC-ALF coefficients are signed powers of two (or zero)
      A.ccalf_coeff[c][f][k] = (int16_t) ( e == 0 ? 0 : ( rng.p( 0.5 ) ? 1 : -1 ) * ( 1 << ( e - 1 ) ) );
    }
  }

  int run()
  {
    W = P.width; H = P.height; w4 = ( W + 3 ) >> 2; h4 = ( H + 3 ) >> 2; ctu = 1 << P.log2_ctu; bd = P.bit_depth;
    layoutSlicesAndTiles();
    cuOf4.assign( (size_t) w4 * h4, -1 ); tuOf4.assign( (size_t) w4 * h4, -1 );
    B.num_cu = B.num_tu = 0; B.num_coef = 0; B.num_dmvr = 0;
    for( size_t i = 0; i < (size_t) w4 * h4; i++ ) { memset( &B.motion[i], 0, sizeof( vvr_motion ) ); B.motion[i].ref_idx[0] = B.motion[i].ref_idx[1] = -1; }
    vvr_pic_header& h = B.hdr; memset( &h, 0, sizeof( h ) );
    h.abi_version = VVR_ABI_VERSION; h.tool_flags = P.tool_flags; h.width = W; h.height = H; h.chroma_format = P.chroma_format; h.bit_depth = bd;
    h.log2_ctu = P.log2_ctu; h.slice_type = P.slice_type; h.poc = P.poc; h.out_slot = P.out_slot; h.min_qp_ts = 4;
    for( int l = 0; l < 2; l++ ) { h.num_ref[l] = P.slice_type == 2 ? 0 : P.num_ref[l]; for( int i = 0; i < VVR_MAX_REFS; i++ ) { h.ref_slot[l][i] = P.ref_slot[l][i]; h.ref_poc[l][i] = P.ref_poc[l][i]; } }
    for( int c = 0; c < 3; c++ ) { h.deblock_beta_offset_div2[c] = (int8_t) ( (int) rng.u( 5 ) - 2 ); h.deblock_tc_offset_div2[c] = (int8_t) ( (int) rng.u( 5 ) - 2 ); }
    h.wrap_offset = P.wrap_offset;
    for( int d = 0; d < 2; d++ )
    {
      // virtual boundaries: distinct multiples of 8 inside the picture, ascending (ph_virtual_boundary_pos_x/y_minus1)
      const int n = ( P.virtual_boundaries >> ( 2 * d ) ) & 3, lim = d ? H : W;
      uint16_t* pos = d ? h.vb_pos_y : h.vb_pos_x;
      int got = 0;
      for( int tries = 0; got < n && tries < 64; tries++ )
      {
        int v = 8 * ( 1 + (int) rng.u( (uint32_t) std::max( 1, lim / 8 - 1 ) ) );
        if( got == 0 && ( P.virtual_boundaries & 16 ) && lim > ctu ) v = ctu * ( 1 + (int) rng.u( (uint32_t) ( ( lim - 1 ) / ctu ) ) );
        if( v <= 0 || v >= lim ) continue;
        bool dup = false; for( int i = 0; i < got; i++ ) dup |= pos[i] == v;
        if( !dup ) pos[got++] = (uint16_t) v;
      }
      std::sort( pos, pos + got );
      if( d ) h.num_hor_vb = (uint8_t) got; else h.num_ver_vb = (uint8_t) got;
    }
    if( h.tool_flags & VVR_TOOL_LADF )
    {
      // sps_ladf_*: 2..5 intervals with rising lower bounds (luma levels) and QP offsets in the syntax range (-63..63, kept small)
      h.ladf_num_intervals = (uint8_t) ( 2 + rng.u( 4 ) );
      int lb = 0;
      for( int k = 0; k < h.ladf_num_intervals; k++ )
      {
        h.ladf_qp_offset[k] = (int8_t) ( (int) rng.u( 13 ) - 6 );
        if( k ) { lb += 1 + (int) rng.u( ( 1u << bd ) / h.ladf_num_intervals ); h.ladf_lower_bound[k] = (int16_t) std::min( lb, ( 1 << bd ) - 2 ); }
      }
    }
    if( P.slice_type == 2 ) h.tool_flags &= ~(uint32_t) VVR_TOOL_WP;
    wpOn = ( h.tool_flags & VVR_TOOL_WP ) && B.wp;
    if( wpOn ) genWp();
    if( ( h.tool_flags & VVR_TOOL_SCALING_LIST ) && B.scaling ) genScalingList();
    ibcOn = ( h.tool_flags & VVR_TOOL_IBC ) != 0;
    if( ibcOn ) { ibcW4 = ( 256 * 128 / ctu ) >> 2; vbL.assign( (size_t) h4 * ibcW4, -1 ); vbC.assign( (size_t) h4 * ibcW4, -1 ); }
    int a = 0;
    const bool dual = P.slice_type == 2 && P.dual_tree > 0 && P.chroma_format;
    if( dual || P.min_cu_log2 == 2 ) { cuOf4C.assign( (size_t) w4 * h4, -1 ); tuOf4C.assign( (size_t) w4 * h4, -1 ); }
    for( int y = 0; y < H; y += ctu ) for( int x = 0; x < W; x += ctu, a++ )
    {
      B.ctu_first_cu[a] = B.num_cu;
      curI = P.slice_type == 2 || ( P.num_slices > 1 && ( ( P.intra_slices >> ( sliceOfCtu[a] & 7 ) ) & 1 ) );
      if( !dual ) { split( x, y, ctu, ctu ); continue; }
      // dual tree: the CTU is split down to 64x64 implicitly; every such node carries its luma tree, then its chroma tree.
      // Inside the picture the node is either not split or quad-split in both trees, which keeps CCLM legal (checkCCLMAllowed)
      const int R = std::min( ctu, 64 );
      for( int ry = y; ry < y + ctu && ry < H; ry += R ) for( int rx = x; rx < x + ctu && rx < W; rx += R )
      {
        const bool inside = rx + R <= W && ry + R <= H;
        const bool nodeQT = R == 64 && inside && rng.p( 0.85 ), nodeNS = R == 64 && inside && !nodeQT;
        for( int tree = VVR_TREE_LUMA; tree <= VVR_TREE_CHROMA; tree++ )
        {
          curTree = tree; cclmOk = R < 64 || inside;
          if( nodeNS ) addCu( rx, ry, R, R );
          else if( nodeQT ) { const int hs = R >> 1; split( rx, ry, hs, hs ); split( rx + hs, ry, hs, hs ); split( rx, ry + hs, hs, hs ); split( rx + hs, ry + hs, hs, hs ); }
          else split( rx, ry, R, R );
        }
      }
      curTree = VVR_TREE_JOINT; cclmOk = true;
    }
    B.ctu_first_cu[a] = B.num_cu;
    if( dual ) deriveLfpDual(); else deriveLfp();       // (deriveLfp handles the chroma-tree CUs of local dual trees through the chroma owner maps)
    if( ( P.tool_flags & VVR_TOOL_LMCS ) && B.lmcs ) genLmcs();
    genAlfParams();
    genLoopFilterParams();
    return 0;
  }
};

} // namespace

__attribute__((visibility("default")))
int vvs_generate( const vvs_params* P, vvs_buffers* B )
{
  if( P->min_cu_log2 < 2 || P->chroma_format > 1 || P->bit_depth < 8 || P->bit_depth > 10 ) return -1;
  Gen g( *P, *B );
  return g.run();
}

} // extern "C"
