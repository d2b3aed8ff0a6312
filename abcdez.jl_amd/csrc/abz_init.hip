/*
 * abz_init.hip -- initial population.
 *
 * Replaces the prior draws / log-priors of src/abcdez_smc.jl:242-243 (mc:117-118)
 * and abcde_init! (src/abcdez_init.jl:2-22): per particle draw theta ~ prior,
 * evaluate log-prior and first distance, and redraw while either is non-finite
 * (init.jl:14-20).  retry r of particle i uses RNG epoch r.
 */
#include "abz_dispatch.h"

#define ABZ_MAX_RETRY 100000u

template <int SIM, int L, int C>
__global__ __launch_bounds__(ABZ_BLOCK) void init_kernel(const HotModel M, double* __restrict__ theta,
                                                         double* __restrict__ logpi, double* __restrict__ delta,
                                                         uint32_t i0, uint32_t n, unsigned long long* __restrict__ bad) {
  constexpr int LD = L * C;
  __shared__ ModelLds<LD> s_model;
  {
    ModelStage<SIM, LD> stage;
    stage.load(M);
    stage.store(s_model);
  }
  __syncthreads();
  const abz_prior_dim* pd = s_model.prior;
  const uint32_t gid = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  const uint32_t grp = gid / L;
  const int j = (int)(gid % L);
  if (grp >= n) return;                       /* whole groups leave together */
  const uint32_t i = i0 + grp;
  const uint64_t seed = M.seed;
  double th[C], pp[C];
  double lp, dl;
  uint32_t retry = 0;
  for (;;) {
    if constexpr (C == 1) {
      const abz_u64x2 w = abz_rng(seed, i, retry, 0, ABZ_RNG_INIT_PRIOR);
      double z0, z1;
      abz_normal_pair(w, &s_model.tab, &z0, &z1);
      th[0] = abz_prior_draw1(&pd[0], w.w0, z0);
      if (pd[0].family >= ABZ_PRIOR_BETA) th[0] = abz_prior_draw_ext(&pd[0], seed, i, retry, 0u, &s_model.tab);
    } else {
#pragma unroll
      for (int m = 0; m < C / 2; ++m) {
        const abz_u64x2 w = abz_rng(seed, i, retry, (uint32_t)(m * L + j), ABZ_RNG_INIT_PRIOR);
        double z0, z1;
        abz_normal_pair(w, &s_model.tab, &z0, &z1);
        const int k = Lay<L, C>::comp(j, m, 0);
        th[2 * m] = abz_prior_draw1(&pd[k], w.w0, z0);
        th[2 * m + 1] = abz_prior_draw1(&pd[k + 1], w.w1, z1);
        if (pd[k].family >= ABZ_PRIOR_BETA)
          th[2 * m] = abz_prior_draw_ext(&pd[k], seed, i, retry, (uint32_t)k, &s_model.tab);
        if (pd[k + 1].family >= ABZ_PRIOR_BETA)
          th[2 * m + 1] = abz_prior_draw_ext(&pd[k + 1], seed, i, retry, (uint32_t)(k + 1), &s_model.tab);
      }
    }
    lp = group_logprior<L, C>(pd, j, th, pp);
    dl = ABZ_NAN;
    if (abz_isfinite(lp)) dl = sim_dist<SIM, L, C>(M, &s_model.tab, j, pp, s_model.y, i, retry, ABZ_RNG_INIT_SIM);   /* init.jl:9-13,17 */
    if (abz_isfinite(dl) && abz_isfinite(lp)) break;                                          /* init.jl:14 */
    if (++retry >= ABZ_MAX_RETRY) {
      if (j == 0) atomicAdd(bad, 1ull);
      break;
    }
  }
  store_row<L, C>(theta + (size_t)i * LD, j, th);
  if (j == 0) { logpi[i] = lp; delta[i] = dl; }
}

int abz_launch_init(abcdez_ctx* ctx, double* theta, double* logpi, double* delta, int64_t i0, int64_t n) {
  if (n <= 0) return 0;
  bool ok = abz_dispatch(ctx->h_model.sim_id, ctx->L, ctx->C, [&](auto S, auto LL, auto CC) {
    hipLaunchKernelGGL((init_kernel<S(), LL(), CC()>), dim3(abz_grid((uint64_t)n * LL())), dim3(ABZ_BLOCK), 0,
                       ctx->stream, ctx->hot, theta, logpi, delta, (uint32_t)i0, (uint32_t)n,
                       ctx->d_scal + ABZ_S_INITBAD);
  });
  if (!ok) { abz_set_error("init: no kernel for this (simulator, ld, lanes) combination"); return -3; }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}
