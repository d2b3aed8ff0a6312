/*
 * abz_smc_swarm.hip -- one DE-Metropolis sweep over the alive particles.
 *
 * Replaces abcdesmc_swarm! (src/abcdez_smc.jl:106-153) and the identity. copies
 * of src/abcdez_smc.jl:337-340.  One fused kernel: donor draw -> proposal ->
 * push_p -> prior support/log-density -> simulator -> distance -> Metropolis
 * accept -> write generation t+1, with the acceptance/simulation counters reduced
 * per block.  Work items are alive RANKS, so every wave is dense regardless of how
 * many particles are dead; dead rows are carried by copy_dead_kernel only when the
 * alive set has changed since the last sweep.
 *
 * HBM roofline: per update 3 rows of 8*ld bytes read (own row, two
 * donor rows) + 16 B state + 12 B indices, one row + 16 B written.  Everything that
 * does not depend on loaded data (prior descriptors, data vector, model scalars) is
 * fetched before the first dependent load, so a wave waits on three memory round
 * trips: alive_idx[rank] -> alive_idx[donor ranks] -> rows.
 */
#include "abz_dispatch.h"

struct SmcSwarmArgs {
  HotModel hm;
  const uint32_t* alive_idx;
  const uint32_t* arank;
  const double* theta;
  const double* logpi;
  const double* delta;
  double* ntheta;
  double* nlogpi;
  double* ndelta;
  uint2* partials;              /* per-block (nacc, nsim) */
  uint8_t* row_synced;          /* per particle: both generations' theta rows are equal (may be NULL) */
  double eps, gamma0, gsig;
  uint32_t n_alive, r_lo, n_work, sweep;
  uint32_t all_alive;           /* alive_idx is the identity: skip the indirections */
};

template <int SIM, int L, int C>
__global__ __launch_bounds__(ABZ_BLOCK) void smc_swarm_kernel(const SmcSwarmArgs a) {
  constexpr int LD = L * C;
  const HotModel& M = a.hm;
  const uint32_t gid = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  const uint32_t grp = gid / L;
  const int j = (int)(gid % L);
  const bool active = grp < a.n_work;
  const uint32_t ri = a.r_lo + (active ? grp : 0u);
  const uint32_t i = a.all_alive ? ri : a.alive_idx[ri];

  __shared__ ModelLds<LD> s_model;

  /* own row + state */
  double ti[C];
  load_row<L, C>(a.theta + (size_t)i * LD, j, ti);
  const double lpi = a.logpi[i];
  const double dli = a.delta[i];
  ModelStage<SIM, LD> stage;                 /* model tables: loads in flight with the row loads */
  stage.load(M);

  /* donors a, b (smc:119-126), gamma = gamma0 (1 + randn gamma_sigma) (smc:128), log(rand) (smc:145) */
  stage.store(s_model);
  __syncthreads();                                                /* sampler + model tables staged */
  uint32_t ra, rb;
  double g, log_u;
  particle_draws<L>(&s_model.tab, M.seed, i, a.sweep, j, a.n_alive, ri, a.gamma0, a.gsig, &ra, &rb, &g, &log_u);
  const uint32_t ia = a.all_alive ? ra : a.alive_idx[ra];
  const uint32_t ib = a.all_alive ? rb : a.alive_idx[rb];
  double ta[C], tb[C];
  load_row<L, C>(a.theta + (size_t)ia * LD, j, ta);
  load_row<L, C>(a.theta + (size_t)ib * LD, j, tb);

  double tp[C], pp[C];
#pragma unroll
  for (int q = 0; q < C; ++q) tp[q] = ti[q] + (ta[q] - tb[q]) * g;

  const double lp = group_logprior<L, C>(s_model.prior, j, tp, pp);   /* smc:134 */
  const bool insupport = !(lp == ABZ_NINF);                       /* smc:135 */
  bool acc = false;
  double dp = dli;
  if (insupport) {
    dp = sim_dist<SIM, L, C>(M, &s_model.tab, j, pp, s_model.y, i, a.sweep, ABZ_RNG_SIM);   /* smc:137 */
    const double w = (lp - lpi) + (abz_kernel_logpdf(M.abck, a.eps, dp) - abz_kernel_logpdf(M.abck, a.eps, dli)); /* smc:140-141 */
    acc = (0.0 <= w) || (log_u < w);                              /* smc:145 */
  }
  if (active) {                                                   /* smc:146-150 + copies :337-340 */
    /* lazy copy: a rejected particle whose row is already identical in both generations'
     * arrays writes nothing (about half of all row writes at a 30 % acceptance rate) */
    const bool synced = a.row_synced ? a.row_synced[i] != 0 : false;
    if (acc || !synced) {
      double to[C];
#pragma unroll
      for (int q = 0; q < C; ++q) to[q] = acc ? tp[q] : ti[q];
      store_row<L, C>(a.ntheta + (size_t)i * LD, j, to);
    }
    if (j == 0) {
      a.nlogpi[i] = acc ? lp : lpi;
      a.ndelta[i] = acc ? dp : dli;
      if (a.row_synced && (acc == synced)) a.row_synced[i] = acc ? 0 : 1;
    }
  }
  block_count2(active && j == 0 && acc, active && j == 0 && insupport, a.partials);
}

/* dead rows of [i0, i0+n): carry generation t into generation t+1 (smc:337-340).
 * One thread per PARTICLE scans the flags (coalesced 4 + 1 bytes); the few rows that need
 * carrying -- dead and not yet present in both generations' arrays -- are copied by that thread. */
template <int LD>
__global__ __launch_bounds__(ABZ_BLOCK) void copy_dead_kernel(const uint32_t* __restrict__ arank,
                                                              const double* __restrict__ theta,
                                                              const double* __restrict__ logpi,
                                                              const double* __restrict__ delta,
                                                              double* __restrict__ ntheta, double* __restrict__ nlogpi,
                                                              double* __restrict__ ndelta, uint32_t i0, uint32_t n,
                                                              uint8_t* __restrict__ dead_synced) {
  const uint32_t g = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  if (g >= n) return;
  const uint32_t i = i0 + g;
  if (arank[i] != ABZ_DEAD) return;
  if (dead_synced) {
    if (dead_synced[i]) return;
    dead_synced[i] = 1;
  }
  const double* __restrict__ src = theta + (size_t)i * LD;
  double* __restrict__ dst = ntheta + (size_t)i * LD;
  if constexpr (LD == 1) {
    dst[0] = src[0];
  } else {
#pragma unroll 4
    for (int k = 0; k < LD; k += 2) *reinterpret_cast<double2*>(dst + k) = *reinterpret_cast<const double2*>(src + k);
  }
  nlogpi[i] = logpi[i];
  ndelta[i] = delta[i];
}

int abz_launch_smc_swarm(abcdez_ctx* ctx, const uint32_t* alive_idx, const uint32_t* arank, uint32_t n_alive,
                         uint32_t r_lo, uint32_t r_hi, const double* theta, const double* logpi, const double* delta,
                         double* ntheta, double* nlogpi, double* ndelta, double eps, double gamma0, double gsig,
                         uint32_t i0, uint32_t n_local, int copy_dead, uint8_t* dead_synced, uint32_t sweep,
                         uint32_t N_total) {
  SmcSwarmArgs a;
  a.hm = ctx->hot; a.alive_idx = alive_idx; a.arank = arank;
  a.theta = theta; a.logpi = logpi; a.delta = delta;
  a.ntheta = ntheta; a.nlogpi = nlogpi; a.ndelta = ndelta;
  const int L = ctx->L, C = ctx->C;
  a.n_work = r_hi - r_lo;
  const unsigned nblocks = abz_grid((uint64_t)a.n_work * (uint64_t)L);
  if (int rc = abz_cnt_reserve(ctx, nblocks ? nblocks : 1)) return rc;
  a.partials = (uint2*)ctx->cnt;
  a.row_synced = dead_synced;
  a.eps = eps; a.gamma0 = gamma0; a.gsig = gsig;
  a.n_alive = n_alive; a.r_lo = r_lo; a.n_work = r_hi - r_lo; a.sweep = sweep;
  a.all_alive = (N_total != 0 && n_alive == N_total) ? 1u : 0u;
  bool ok = abz_dispatch(ctx->h_model.sim_id, L, C, [&](auto S, auto LL, auto CC) {
    if (copy_dead && n_local > 0) {
      hipLaunchKernelGGL((copy_dead_kernel<LL() * CC()>), dim3(abz_grid((uint64_t)n_local)), dim3(ABZ_BLOCK), 0,
                         ctx->stream, arank, theta, logpi, delta, ntheta, nlogpi, ndelta, i0, n_local, dead_synced);
    }
    if (a.n_work > 0) {
      if (ctx->timing) (void)hipEventRecord(ctx->ev0, ctx->stream);
      hipLaunchKernelGGL((smc_swarm_kernel<S(), LL(), CC()>), dim3(abz_grid((uint64_t)a.n_work * LL())),
                         dim3(ABZ_BLOCK), 0, ctx->stream, a);
      if (ctx->timing) {
        (void)hipEventRecord(ctx->ev1, ctx->stream);
        ctx->ev_pending = true;
        ctx->ev_units = a.n_work;
      }
    }
  });
  if (!ok) { abz_set_error("smc_swarm: no kernel for this (simulator, ld, lanes) combination"); return -3; }
  ABZ_HIP_CHECK(hipGetLastError());
  return abz_reduce_partials(ctx, ctx->cnt, nblocks, ctx->d_scal + ABZ_S_NACC);
}
