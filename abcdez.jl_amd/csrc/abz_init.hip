/*
 * abz_init.hip -- initial population.
 *
 * Replaces the prior draws / log-priors of src/abcdez_smc.jl:242-243 (mc:117-118)
 * and abcde_init! (src/abcdez_init.jl:2-22): per particle draw theta ~ prior,
 * evaluate log-prior and first distance, and redraw while either is non-finite
 * (init.jl:14-20).  retry r of particle i uses RNG epoch r.
 */
#define ABZ_PRIOR_WRAP 1        /* this translation unit's kernels evaluate the wrapper prior families too (include/abcdez_spec.h) */
#include "abz_dispatch.h"
#include "abz_kernels.h"

template <int SIM, int L, int C>
__global__ __launch_bounds__(ABZ_BLOCK) void init_kernel(const HotModel M, double* __restrict__ theta,
                                                         double* __restrict__ logpi, double* __restrict__ delta,
                                                         uint32_t i0, uint32_t n, unsigned long long* __restrict__ bad,
                                                         uint64_t* __restrict__ stamp) {
  init_kernel_body<SIM, L, C>(M, theta, logpi, delta, i0, n, bad, stamp);
}

int abz_launch_init(abcdez_ctx* ctx, double* theta, double* logpi, double* delta, int64_t i0, int64_t n) {
  if (n <= 0) return 0;
  uint64_t* stamp = ctx->stamp_cur;           /* NULL unless blobs are on (abcdez_ctx_set_stamps) */
  if (ctx->h_model.sim_id == ABZ_SIM_USER)
    return abz_jit_launch_init(ctx, theta, logpi, delta, (uint32_t)i0, (uint32_t)n, ctx->d_scal + ABZ_S_INITBAD, stamp);
  bool ok = abz_dispatch(ctx->h_model.sim_id, ctx->L, ctx->C, [&](auto S, auto LL, auto CC) {
    auto kern = init_kernel<S(), LL(), CC()>;
    hipLaunchKernelGGL(kern, dim3(abz_persistent_grid(ctx, kern, abz_grid((uint64_t)n * LL()), ABZ_BLOCK)), dim3(ABZ_BLOCK), 0,
                       ctx->stream, ctx->hot, theta, logpi, delta, (uint32_t)i0, (uint32_t)n,
                       ctx->d_scal + ABZ_S_INITBAD, stamp);
  });
  if (!ok) { abz_set_error("init: no kernel for this (simulator, ld, lanes) combination"); return -3; }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

/* ---- blobs: rebuild the simulated data of every particle from its stamp (abz_kernels.h) ---- */
template <int SIM, int L, int C>
__global__ __launch_bounds__(ABZ_BLOCK) void blob_eval_kernel(const HotModel M, const double* __restrict__ theta,
                                                              const uint64_t* __restrict__ stamp, uint32_t n,
                                                              double* __restrict__ blob, double* __restrict__ delta_out,
                                                              uint32_t nbw) {
  blob_eval_kernel_body<SIM, L, C>(M, theta, stamp, n, blob, delta_out, nbw);
}

int abz_launch_blob_eval(abcdez_ctx* ctx, const double* theta, const uint64_t* stamp, int64_t n, double* blob,
                         double* delta_out, uint32_t nbw) {
  if (n <= 0) return 0;
  if (ctx->h_model.sim_id == ABZ_SIM_USER) return abz_jit_launch_blob(ctx, theta, stamp, (uint32_t)n, blob, delta_out, nbw);
  bool ok = abz_dispatch(ctx->h_model.sim_id, ctx->L, ctx->C, [&](auto S, auto LL, auto CC) {
    auto kern = blob_eval_kernel<S(), LL(), CC()>;
    hipLaunchKernelGGL(kern, dim3(abz_persistent_grid(ctx, kern, abz_grid((uint64_t)n * LL()), ABZ_BLOCK)), dim3(ABZ_BLOCK), 0,
                       ctx->stream, ctx->hot, theta, stamp, (uint32_t)n, blob, delta_out, nbw);
  });
  if (!ok) { abz_set_error("blob_eval: no kernel for this (simulator, ld, lanes) combination"); return -3; }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}
