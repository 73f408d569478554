/* pmr_internal.h -- library-internal seam between pmr_chain.c and pmr_dsd.c (not installed, not part of the C-ABI).
 * `dsd_in` (reference src/dsd_in.c:167-168) runs the same dc-block + msresamp_crcf front end as the channelizer app
 * (src/sdr_pmr446.c:795-796); pmr_dsd.c borrows it from a pmr_chain created in front-end-only mode. */
#ifndef PMR_INTERNAL_H
#define PMR_INTERNAL_H

#include <stdint.h>
#include "../../include/pmr_chain.h"

/* like pmr_chain_create, but accepts num_channels == 1 (no channelizer will ever run on this handle) */
pmr_chain pmr_chain_create_frontend(const pmr_chain_cfg *cfg);
/* dc-block + resampler of one block on the front-end stream; the resampled samples land in the ring at absolute
 * indices [*xr_abs0, *xr_abs0 + *ny), dc carry fully applied.  Nothing is synchronised. */
int pmr_chain_frontend_block(pmr_chain q, const void *d_iq, unsigned n_in, unsigned *ny, uint64_t *xr_abs0);
/* resampled samples a block of n_in raw samples WILL yield (closed form; nothing is advanced) */
unsigned pmr_chain_plan_resampled(pmr_chain q, unsigned n_in);
typedef struct { void *d_xr; uint64_t xr_mask; void *stream_fe; void *d_in; unsigned res_size; int device; } pmr_fe_view;
void pmr_chain_frontend_view(pmr_chain q, pmr_fe_view *v);

#endif
