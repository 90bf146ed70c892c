// vvdec_amd/csrc/vvr_host.h — host-side internals shared by vvr_api.cpp (context, job pipeline, launches) and vvr_prepare.cpp (validation and
// the work lists a picture's kernels iterate over).  Nothing here is part of the ABI (include/vvr.h).
#pragma once
#include "vvr_device.h"
#include <mutex>
#include <string>
#include <vector>

static inline size_t alignUp( size_t v, size_t a ) { return ( v + a - 1 ) / a * a; }

enum { K_MC, K_MC_DMVR, K_MC_AFFINE, K_LMCS, K_ITRANS, K_INTRA, K_RESI_ADD, K_DEBLOCK_V, K_DEBLOCK_H, K_SAO, K_ALF, K_COPY, K_OUTPUT, K_LF_INIT, K_INTRA_LEAF,
       K_DEBLOCK4, K_ALF_PLANES /* the chain of the pictures the fused passes do not cover: k_deblock4 in place, k_alf_luma + k_alf_chroma(_tile) */, K_NUM };

// A picture description resident in HBM together with its device work lists: every pointer is a device address inside one blob.
// Streaming submissions (vvr_submit) use the blob of a ring entry owned by the context; vvr_prepare gives the handle a blob of its own.
struct vvr_prepared {
  vvr_pic_header hdr;
  PicDev   pic;
  McItem*  mcItems = nullptr; int numMc = 0;               // plain tiles the host wrote (SbTMVP sub-blocks: their motion comes from the motion field)
  McItem*  mcDev = nullptr; int numMcDev = 0;              // plain tiles k_expand_mc writes (room reserved behind the uploaded image); BDOF and DMVR tiles are all written there
  McCuRef* mcCus = nullptr; int numMcCus = 0;              // the CUs k_expand_mc expands
  McItem*  bdofItems = nullptr; int numBdofItems = 0;      // tiles of CUs in BDOF mode (their own launch: larger LDS footprint)
  McItem*  dmvrItems = nullptr; int numDmvrItems = 0;      // sub-blocks that run decoder-side MV refinement
  McItem*  affItems = nullptr; int numAffItems = 0;        // tiles of affine CUs
  McItem*  rprItems = nullptr; int numRprItems = 0;        // tiles of CUs that predict from a scaled reference picture
  uint32_t numDmvr = 0;                                    // delta-MV entries the DMVR kernel writes (pairs of ints)
  TbItem*  tbItems[3] = { nullptr, nullptr, nullptr }; int numTb[3] = { 0, 0, 0 };   // size classes 16 / 32 / 64
  IntraItem* intraItems = nullptr; IntraUnit* units = nullptr; int numActive = 0, numIntra = 0;
  bool     intraFine = false;                              // (tile path) every unit is a whole CTU of intra CUs: the CTU wavefront may be resolved block by block (k_intra<.., FINE>)
  bool     intraLeaf = false;                              // the items are those of k_intra_leaf (one wavefront per block, no units): a picture with scattered intra blocks
  int      intraWorkgroups = 0;                            // workgroups the intra stage is launched with (the dependency front they can keep busy)
  IntraItem* resiItems = nullptr; int numResi = 0;         // scaled chroma residuals of inter blocks (k_resi_add); with them the stage runs as luma units, k_resi_add, chroma units
  int      numLumaUnits = 0, intraWorkgroupsChroma = 0;    // (the first numLumaUnits entries of `units` are the luma units then)
  // deblocking edge parameters derived on the device (VVR_TOOL_LFP_ON_DEVICE): room behind the uploaded image for the per-cell records of both
  // trees, the motion field as the filter sees it (from the CU records; sub-block CUs: scattered from lfSb) and the two tables; pic.lfp points at the tables
  bool     lfpOnDevice = false;
  struct LfCell* lfCell = nullptr; struct LfCell* lfCellC = nullptr; struct LfMv* lfMv = nullptr; uint32_t* lfRef = nullptr;
  const struct LfSbCell* lfSb = nullptr; int numLfSb = 0;
  vvr_lfp* lfpDev[2] = { nullptr, nullptr };
  uint32_t numCu = 0, numTu = 0;
  double   bytes[K_NUM] = { 0 };                           // algorithmic bytes per kernel
  double   bytesBdof = 0, bytesIntraLuma = 0, bytesTb[3] = { 0, 0, 0 };      // shares of bytes[K_MC] (the BDOF launch), bytes[K_INTRA] (the luma launch), bytes[K_ITRANS] (per size class)
  // ownership (vvr_prepare handles only)
  void*    blob = nullptr; size_t blobBytes = 0;
  int32_t* dmvrHost = nullptr;                             // pinned + device-mapped, 2 * numDmvr ints: the DMVR kernel writes the delta MVs here
  vvr_motion* colHost = nullptr; size_t numCol = 0;        // pinned + device-mapped: collocated motion (VVR_TOOL_COL_MOTION), filled by the host stage, patched by the DMVR kernel
  struct vvr_context* owner = nullptr;
};

// Host memory the device can read directly (vvr_host_alloc): arrays of a submitted picture that lie in it are not staged, they are copied to HBM
// from where they are.
struct PinnedRanges {
  std::mutex mu;
  std::vector<std::pair<const char*, size_t>> r;
  bool contains( const void* p, size_t n )
  {
    std::lock_guard<std::mutex> lk( mu );
    for( auto& e : r ) if( (const char*) p >= e.first && (const char*) p + n <= e.first + e.second ) return true;
    return false;
  }
};
struct DirectCopy { const void* src; size_t n, off; };

// Reusable scratch of one preparing thread: the lists are built here (no allocation in the steady state), then packed into pinned memory.
struct PrepScratch;
PrepScratch* vvr_scratch_create();
int          vvr_host_band_pictures();
void         vvr_scratch_intra_leaf( PrepScratch*, bool on, bool byLevel = false );         // pictures with scattered intra blocks take the one-wavefront-per-block path (default on)
void         vvr_scratch_parts_for_all( PrepScratch*, bool on );      // the next pictures built with this scratch: in bands of CTU rows over the helpers whatever their kind (else: I pictures only)
void         vvr_scratch_destroy( PrepScratch* );
void         vvr_scratch_warm( PrepScratch*, const vvr_config& cfg );      // allocate and touch room for an ordinary picture of this size (call from the thread that will use it)

// validation of a picture description against the context configuration (no device access)
int    vvr_host_validate( const vvr_config& cfg, const vvr_picture* p, std::string& err );
int    vvr_host_validate_header( const vvr_config& cfg, const vvr_picture* p, std::string& err );      // the O(1) part of it: header, tables, presence of the arrays
int    vvr_host_validate_records( const vvr_config& cfg, const vvr_picture* p, std::string& err );     // the per-record part
// host glue: the work lists of one picture (what DecCu::TaskTrafoCtu / TaskInterCtu / the intra task iterate over, DecCu.cpp:106-160); returns the
// number of bytes the picture needs in HBM
// Other threads that can lend a hand to the thread that prepares a picture (vvr_api.cpp: the library's worker threads).  run( n, fn ): fn( part,
// scratch ) for part = 0 .. n - 1, part 0 by the caller on `own`, the others by whoever is free - each with a scratch of its own - or by the caller
// itself; returns when all are done.
#include <functional>
struct HostHelpers
{
  virtual ~HostHelpers() {}
  virtual int  width() const = 0;       // threads that may run parts at the same time (incl. the caller)
  virtual void run( int n, PrepScratch& own, const std::function<void( int, PrepScratch& )>& fn ) = 0;
  virtual void notInParts() {}          // vvr_host_build has decided to build this picture alone (whoever held others back for its parts lets them go)
};
// validateRecords: the CU / TU records are checked on the way (vvr_host_validate_records has not been called); helpers: a picture whose work lists can be
// built in independent parts (every CU an intra CU: no analysis across CTUs) is - an I picture is what everything of the next GOP waits for
int    vvr_host_build( const vvr_picture* p, PrepScratch& S, size_t* totalBytes, std::string& err, PinnedRanges* pinned = nullptr, HostHelpers* helpers = nullptr, bool validateRecords = false );
// the H2D image: every staged part at its offset (256-byte aligned) into `host` (pinned memory of at least totalBytes); the parts that are
// copied straight from the caller's pinned arrays, and the byte range [begin, end) of the image that is staged
void   vvr_host_pack( const PrepScratch& S, char* host );
size_t vvr_host_staged_bytes( const PrepScratch& S );      // bytes of the image that have a host side (the rest is written by the device)
// collocated motion before refinement: every second 4x4 unit of the picture's motion field in both directions (DecCu.cpp:232-253); dst holds vvr_host_num_col() records
size_t vvr_host_num_col( const vvr_picture* p );
void   vvr_host_gather_col( const vvr_picture* p, vvr_motion* dst );
void   vvr_host_upload_plan( const PrepScratch& S, std::vector<DirectCopy>& direct, size_t* stagedBegin, size_t* stagedEnd );
// device pointers of a prepared picture whose image sits at devBase
void   vvr_host_bind( const PrepScratch& S, vvr_prepared& q, char* devBase );
