/* abz_hotmodel.h -- the by-value model argument of the kernels (device-only header: also part of the
 * translation unit hiprtc compiles for user-supplied simulators). */
#ifndef ABZ_HOTMODEL_H
#define ABZ_HOTMODEL_H

#include "abcdez_spec.h"

/* ---- the model fields the kernels touch, passed BY VALUE in the kernel arguments so they
 * arrive through scalar loads of the kernarg segment instead of a chain of dependent
 * global loads (the first build spent 80 % of its wave-cycles waiting on those).          */
struct HotModel {
  uint64_t seed;
  const abz_prior_dim* prior;   /* device, ld entries */
  const double* data;           /* device, n_data values */
  const abz_tables* tables;     /* device copy of the sampler tables */
  const double* mv;             /* NULL, or [mu | W | L] of a correlated Normal prior (abcdez_spec.h), device memory */
  const double* ext;            /* NULL, or the records of the wrapper prior families (truncated(...), MixtureModel), device memory */
  double sim_p[8];
  int32_t d, abck, n_data, n_blob;
  int32_t sim_i[2];             /* integer forms of simulator parameters: [0] = RK4 steps per observation (Lotka-Volterra, sim_p[3]) */
};

/* cumulative sweep-counter slots (abz_ctx.h, ABZ_S_CSLOT0): ABZ_CSLOTS slots of ABZ_CSTRIDE u64 (one 64-B line);
 * inside a slot: counter classes */
/* rows of at most two doubles are kept double-buffered by the packed sweeps (abz_kernels.h, smc_swarm_packed_body) */
#define ABZ_ROWS_DOUBLE_BUFFERED(ld) ((ld) <= 2)
#define ABZ_CSLOTS 256
#define ABZ_CSTRIDE 8
#define ABZ_C_NACC 0      /* sweep: accepted (smc:150)                     */
#define ABZ_C_NSIM 1      /* sweep: simulated (smc:138, mc:44)             */
#define ABZ_C_RACC 2      /* replay: accepted / simulated over all alive ranks */
#define ABZ_C_RSIM 3
#define ABZ_C_MCGT 4      /* abcdemc sweep: new distances > eps_target (mc:156) */
#define ABZ_C_MCSIM 5     /* abcdemc sweep: simulated (mc:44)                   */
#define ABZ_C_DISCARD 6   /* sweeps whose caller wants no counters         */
#define ABZ_MMSLOTS 64

#endif
