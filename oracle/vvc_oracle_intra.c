/* oracle/vvc_oracle_intra.c — CPU restatement (TEST INFRASTRUCTURE): intra prediction.  (filled in below) */
#include "vvc_oracle_common.h"
int vvo_intra_tu( const vvr_picture* pic, const vvr_cu* cu, const vvr_tu* tu, uint32_t tu_idx, int comp, vvo_planes* reco,
                  const int32_t* order, const int16_t* resi, int has_resi )
{
  (void) pic; (void) cu; (void) tu; (void) tu_idx; (void) comp; (void) reco; (void) order; (void) resi; (void) has_resi;
  vvo_set_error( "intra prediction not restated yet" );
  return -1;
}
