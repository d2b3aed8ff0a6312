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
  double sim_p[8];
  int32_t d, abck, n_data, n_blob;
};

#endif
