/*
 * oracle/vvc_oracle.h — CPU restatement of the VVC reconstruction hot path (plain C99).
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker.  The product (vvdec_amd/) never links, loads or falls back to it.
 *
 * Every function restates one piece of the reference decoder and cites the file:line it follows (paths relative to
 * /root/reference/source/Lib).  The restatement is pinned against the real reference classes (oracle/_ref, driven by
 * oracle/ref_harness.cpp) by tests/test_oracle_vs_ref.py on seeded pictures, and against the committed golden
 * fixtures under tests/golden/ which that harness produced.
 */
#ifndef VVC_ORACLE_H
#define VVC_ORACLE_H
#include "../include/vvr.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { VVO_STOP_AFTER_RECO = 4, VVO_STOP_AFTER_DBK = 8, VVO_STOP_AFTER_SAO = 16 };   /* same values as the VVREF_* flags */

/* whole picture: same calling convention as vvref_reconstruct (tight planes, ref_planes[slot*3+comp]) */
int vvo_reconstruct( const vvr_picture* pic, const uint16_t* const* ref_planes, uint16_t* const* out_planes, int flags );

/* one transform block: levels -> residual (row-major bw x bh) */
int vvo_residual( const vvr_pic_header* hdr, const vvr_cu* cu, const vvr_tu* tu, int comp, const int16_t* coef, int16_t* resi );

/* DMVR delta MVs (hor, ver in 1/16 sample) of the last vvo_reconstruct call, indexed like vvr_read_dmvr (cu.dmvr_off + sub-block);
 * returns the number of entries produced */
uint32_t vvo_get_dmvr( int32_t* dst, uint32_t max_entries );

const char* vvo_last_error( void );

#ifdef __cplusplus
}
#endif
#endif
