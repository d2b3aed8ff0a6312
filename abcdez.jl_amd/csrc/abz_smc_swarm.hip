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
#include "abz_kernels.h"

template <int SIM, int L, int C>
__global__ __launch_bounds__(ABZ_BLOCK) void smc_swarm_kernel(const SmcSwarmArgs a) {
  smc_swarm_kernel_body<SIM, L, C>(a);
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
                                                              uint8_t* __restrict__ dead_synced,
                                                              const uint64_t* __restrict__ stamp,
                                                              uint64_t* __restrict__ nstamp) {
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
  if (nstamp) nstamp[i] = stamp[i];
}

int abz_launch_smc_swarm(abcdez_ctx* ctx, const uint32_t* alive_idx, const uint32_t* arank, uint32_t n_alive,
                         uint32_t r_lo, uint32_t r_hi, const double* theta, const double* logpi, const double* delta,
                         double* ntheta, double* nlogpi, double* ndelta, double eps, double gamma0, double gsig,
                         uint32_t i0, uint32_t n_local, int copy_dead, uint8_t* dead_synced, uint32_t sweep,
                         uint32_t N_total, uint32_t* alive_out, uint8_t* acc_flag, int want_counts) {
  SmcSwarmArgs a;
  a.hm = ctx->hot; a.alive_idx = alive_idx; a.arank = arank;
  a.theta = theta; a.logpi = logpi; a.delta = delta;
  a.ntheta = ntheta; a.nlogpi = nlogpi; a.ndelta = ndelta;
  const int L = ctx->L, C = ctx->C;
  a.n_work = r_hi - r_lo;
  const unsigned nblocks = abz_grid((uint64_t)a.n_work * (uint64_t)L);
  a.cslots = ctx->d_scal + ABZ_S_CSLOT0;
  a.c_cls = want_counts ? ABZ_C_NACC : ABZ_C_DISCARD;
  a.row_synced = dead_synced;
  a.eps = eps; a.gamma0 = gamma0; a.gsig = gsig;
  a.n_alive = n_alive; a.r_lo = r_lo; a.n_work = r_hi - r_lo; a.sweep = sweep;
  a.all_alive = (N_total != 0 && n_alive == N_total && !alive_out) ? 1u : 0u;
  a.rows = alive_out ? 1u : 0u;
  a.alive_out = alive_out;
  a.acc_flag = acc_flag;
  /* blob stamps: the row store updates them in place (like logpi / delta), the double buffer writes the next array */
  a.stamp = ctx->stamp_cur;
  a.nstamp = ctx->stamp_cur ? (alive_out ? ctx->stamp_cur : ctx->stamp_nxt) : nullptr;
  bool ok = true;
  if (copy_dead && n_local > 0) {
    ok = abz_dispatch_ld(ctx->h_model.ld, [&](auto LD) {
      hipLaunchKernelGGL((copy_dead_kernel<LD()>), dim3(abz_grid((uint64_t)n_local)), dim3(ABZ_BLOCK), 0, ctx->stream,
                         arank, theta, logpi, delta, ntheta, nlogpi, ndelta, i0, n_local, dead_synced,
                         (const uint64_t*)a.stamp, a.nstamp);
    });
  }
  if (ok && a.n_work > 0) {
    if (ctx->timing) (void)hipEventRecord(ctx->ev0, ctx->stream);
    if (ctx->h_model.sim_id == ABZ_SIM_USER) {
      if (int rc = abz_jit_launch_smc(ctx, &a, nblocks)) return rc;
    } else {
      ok = abz_dispatch(ctx->h_model.sim_id, L, C, [&](auto S, auto LL, auto CC) {
        hipLaunchKernelGGL((smc_swarm_kernel<S(), LL(), CC()>), dim3(nblocks), dim3(ABZ_BLOCK), 0, ctx->stream, a);
      });
    }
    if (ctx->timing) {
      (void)hipEventRecord(ctx->ev1, ctx->stream);
      ctx->ev_pending = true;
      ctx->ev_units = a.n_work;
    }
  }
  if (!ok) { abz_set_error("smc_swarm: no kernel for this (simulator, ld, lanes) combination"); return -3; }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

/* replay of the other ranks' accepted proposals on this rank's replica (abz_kernels.h) */
template <int L, int C>
__global__ __launch_bounds__(ABZ_BLOCK) void smc_replay_kernel(const SmcReplayArgs a) {
  smc_replay_kernel_body<L, C>(a);
}

int abz_launch_smc_replay(abcdez_ctx* ctx, const uint32_t* alive_row, uint32_t* alive_out, uint32_t n_alive,
                          uint32_t skip_lo, uint32_t skip_hi, double* slot0, double* slot1, const uint8_t* acc_flag,
                          double gamma0, double gsig, uint32_t sweep) {
  SmcReplayArgs a;
  a.hm = ctx->hot; a.alive_idx = alive_row; a.alive_out = alive_out; a.acc_flag = acc_flag;
  a.slot0 = slot0; a.slot1 = slot1; a.gamma0 = gamma0; a.gsig = gsig;
  a.n_alive = n_alive; a.skip_lo = skip_lo; a.skip_hi = skip_hi; a.sweep = sweep;
  const unsigned nblocks = (unsigned)(((uint64_t)n_alive + ABZ_REPLAY_CHUNK - 1) / ABZ_REPLAY_CHUNK);
  a.cslots = ctx->d_scal + ABZ_S_CSLOT0;
  bool ok = abz_dispatch_lc(ctx->L, ctx->C, [&](auto LL, auto CC) {
    hipLaunchKernelGGL((smc_replay_kernel<LL(), CC()>), dim3(nblocks), dim3(ABZ_BLOCK), 0, ctx->stream, a);
  });
  if (!ok) { abz_set_error("smc_replay: unsupported layout"); return -3; }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

/* ================================================================ packed population (abz_kernels.h) */
template <int SIM, int L, int C>
__global__ __launch_bounds__(ABZ_BLOCK) void smc_swarm_packed_kernel(const SmcPackedArgs a) {
  smc_swarm_packed_body<SIM, L, C>(a);
}
template <int L, int C>
__global__ __launch_bounds__(ABZ_BLOCK) void smc_replay_packed_kernel(const SmcReplayPackedArgs a) {
  smc_replay_packed_body<L, C>(a);
}

int abz_launch_smc_swarm_packed(abcdez_ctx* ctx, const uint32_t* bits, uint32_t* bits_out, uint32_t n_alive, uint32_t r_lo,
                                uint32_t r_hi, double* slot0, double* slot1, double* logpi, double* delta, uint8_t* flags,
                                double eps, double gamma0, double gsig, uint32_t sweep, int want_counts) {
  SmcPackedArgs a;
  a.hm = ctx->hot; a.bits = bits; a.bits_out = bits_out; a.slot0 = slot0; a.slot1 = slot1; a.logpi = logpi; a.delta = delta;
  a.cslots = ctx->d_scal + ABZ_S_CSLOT0;
  a.c_cls = want_counts ? ABZ_C_NACC : ABZ_C_DISCARD;
  a.flags = flags; a.stamp = ctx->stamp_cur;
  a.eps = eps; a.gamma0 = gamma0; a.gsig = gsig;
  a.n_alive = n_alive; a.r_lo = r_lo; a.n_work = r_hi - r_lo; a.sweep = sweep;
  if (a.n_work == 0) return 0;
  const int L = ctx->L, C = ctx->C;
  if (L > 8) { abz_set_error("smc_swarm_packed: at most 8 lanes per particle (a block must cover whole bitmap words)"); return -3; }
  const unsigned nblocks = abz_grid((uint64_t)a.n_work * (uint64_t)L);
  if (ctx->timing) (void)hipEventRecord(ctx->ev0, ctx->stream);
  bool ok = true;
  if (ctx->h_model.sim_id == ABZ_SIM_USER) {
    if (int rc = abz_jit_launch_smc_packed(ctx, &a, nblocks)) return rc;
  } else {
    ok = abz_dispatch(ctx->h_model.sim_id, L, C, [&](auto S, auto LL, auto CC) {
      if constexpr (LL() <= 8)
        hipLaunchKernelGGL((smc_swarm_packed_kernel<S(), LL(), CC()>), dim3(nblocks), dim3(ABZ_BLOCK), 0, ctx->stream, a);
    });
  }
  if (ctx->timing) {
    (void)hipEventRecord(ctx->ev1, ctx->stream);
    ctx->ev_pending = true;
    ctx->ev_units = a.n_work;
  }
  if (!ok) { abz_set_error("smc_swarm_packed: no kernel for this (simulator, ld, lanes) combination"); return -3; }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

int abz_launch_smc_replay_packed(abcdez_ctx* ctx, const uint32_t* bits, uint32_t* bits_out, uint32_t n_alive,
                                 uint32_t skip_lo, uint32_t skip_hi, double* slot0, double* slot1, double* logpi,
                                 const uint8_t* flags, double gamma0, double gsig, uint32_t sweep) {
  SmcReplayPackedArgs a;
  a.hm = ctx->hot; a.bits = bits; a.bits_out = bits_out; a.flags = flags; a.slot0 = slot0; a.slot1 = slot1; a.logpi = logpi;
  a.stamp = ctx->stamp_cur;
  a.cslots = ctx->d_scal + ABZ_S_CSLOT0;
  a.gamma0 = gamma0; a.gsig = gsig; a.n_alive = n_alive; a.skip_lo = skip_lo; a.skip_hi = skip_hi; a.sweep = sweep;
  const unsigned nblocks = (unsigned)(((uint64_t)n_alive + ABZ_REPLAY_CHUNK - 1) / ABZ_REPLAY_CHUNK);
  bool ok = abz_dispatch_lc(ctx->L, ctx->C, [&](auto LL, auto CC) {
    hipLaunchKernelGGL((smc_replay_packed_kernel<LL(), CC()>), dim3(nblocks), dim3(ABZ_BLOCK), 0, ctx->stream, a);
  });
  if (!ok) { abz_set_error("smc_replay_packed: unsupported layout"); return -3; }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}
