/*
 * abz_mc_swarm.hip -- one greedy ABC-DE-MCMC sweep over all N particles, plus the
 * resampling gathers of the packed population and push_p.
 *
 * abcdemc_swarm! (src/abcdez_mc.jl:5-61) with the copies of mc:140-143 fused in.
 * The "better particle" draw s = rand((1:N)[Ds .<= Ds[i]]) (mc:23) indexes the
 * (Ds, index)-sorted order built once per generation by abcdez_mc_rank_prepare:
 * the candidate set is order[0 .. cnt), cnt = upper_bound(sorted_delta, Ds[i]).
 */
#include "abz_dispatch.h"
#include "abz_kernels.h"

template <int SIM, int L, int C, bool PLAIN>
__global__ __launch_bounds__(ABZ_BLOCK) void mc_swarm_kernel(const McSwarmArgs a) {
  mc_swarm_kernel_body<SIM, L, C, PLAIN>(a);
}

int abz_launch_mc_swarm(abcdez_ctx* ctx, const uint32_t* order, const uint32_t* cnt, uint32_t N,
                        const double* theta, const double* logpi, const double* delta, double* ntheta, double* nlogpi,
                        double* ndelta, double eps_pop, double eps_target, double gamma0, double gsig, uint32_t i0,
                        uint32_t n_local, uint32_t sweep, const unsigned long long* eps_pop_dev, const unsigned long long* seq_dev,
                        const unsigned long long* nabove_dev, int rank_built) {
  if (n_local == 0) return 0;
  McSwarmArgs a;
  a.rank_built = (rank_built < 0 ? (order != nullptr && cnt != nullptr) : rank_built != 0) ? 1u : 0u;
  a.hm = ctx->hot; a.order = order; a.cnt = cnt;
  a.theta = theta; a.logpi = logpi; a.delta = delta;
  a.ntheta = ntheta; a.nlogpi = nlogpi; a.ndelta = ndelta;
  const unsigned ntiles = abz_grid((uint64_t)n_local * (uint64_t)ctx->L);       /* tiles of ABZ_BLOCK threads; the workgroups loop over them */
  a.cslots = ctx->d_scal + ABZ_S_CSLOT0;
  a.mm_cur = ctx->d_scal + ABZ_S_MM0 + (size_t)ctx->mm_bank * 2 * ABZ_MMSLOTS;
  a.mm_nxt = ctx->d_scal + ABZ_S_MM0 + (size_t)(1 - ctx->mm_bank) * 2 * ABZ_MMSLOTS;
  a.eps_pop = eps_pop; a.eps_target = eps_target; a.gamma0 = gamma0; a.gsig = gsig;
  a.eps_pop_dev = eps_pop_dev;
  a.seq_dev = seq_dev;
  a.nabove_dev = nabove_dev;
  a.reject_fail = ctx->d_scal + ABZ_S_MC_REJFAIL;
  a.N = N; a.i0 = i0; a.n_local = n_local; a.sweep = sweep;
  a.stamp = ctx->stamp_cur; a.nstamp = ctx->stamp_cur ? ctx->stamp_nxt : nullptr;      /* blob stamps */
  bool ok = true;
  const int tk = abz_time_begin(ctx);
  if (ctx->user_module) {          /* kernels compiled for this model at run time (abz_jit.hip) */
    if (int rc = abz_jit_launch_mc(ctx, &a, ntiles)) return rc;
  } else {
    ok = abz_dispatch(ctx->h_model.sim_id, ctx->L, ctx->C, [&](auto S, auto LL, auto CC) {
      if (ctx->prior_plain) {
        auto kern = mc_swarm_kernel<S(), LL(), CC(), true>;
        hipLaunchKernelGGL(kern, dim3(abz_persistent_grid(ctx, kern, ntiles, ABZ_BLOCK)), dim3(ABZ_BLOCK), 0, ctx->stream, a);
      } else {
        auto kern = mc_swarm_kernel<S(), LL(), CC(), false>;
        hipLaunchKernelGGL(kern, dim3(abz_persistent_grid(ctx, kern, ntiles, ABZ_BLOCK)), dim3(ABZ_BLOCK), 0, ctx->stream, a);
      }
    });
  }
  abz_time_end(ctx, tk, n_local);
  if (!ok) { abz_set_error("mc_swarm: no kernel for this (simulator, ld, lanes) combination"); return -3; }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

/* ---- S8 gathers: thetas .= thetas[inds] etc. (src/abcdez_smc.jl:96-103) on the packed store: source = current row of
 * inds[s], destination = the OTHER slot of s (never a current row, so no source is overwritten); bits_flip_kernel then
 * flips every bit, in both bit arrays.                                                                          */
template <int L, int C>
__global__ __launch_bounds__(ABZ_BLOCK) void resample_gather_packed_kernel(
    const uint32_t* __restrict__ inds, uint32_t N, const uint32_t* __restrict__ bits, double* __restrict__ slot0,
    double* __restrict__ slot1, const double* __restrict__ logpi, const double* __restrict__ delta,
    double* __restrict__ nlogpi, double* __restrict__ ndelta, double* __restrict__ wns, uint8_t* __restrict__ alive,
    const uint64_t* __restrict__ stamp, uint64_t* __restrict__ nstamp) {
  constexpr int LD = L * C;
  const uint32_t gid = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  const uint32_t s = gid / L;
  const int j = (int)(gid % L);
  if (s >= N) return;
  const uint32_t src = inds[s];
  uint32_t bs = (bits[src >> 5] >> (src & 31u)) & 1u, bd = ((bits[s >> 5] >> (s & 31u)) & 1u) ^ 1u;
  /* double-buffered rows: the sources (alive, so inside the prefix) all live in the prefix's slot P = bit of position 0;
   * EVERY destination is the other slot, whatever a dead position's stale bit says -- bits_set_kernel then gives all N
   * positions that one parity, and the sweeps' donors can keep assuming it */
  if constexpr (ABZ_ROWS_DOUBLE_BUFFERED(LD)) { bs = bits[0] & 1u; bd = bs ^ 1u; }
  double t[C];
  load_row<L, C>((bs ? slot1 : slot0) + (size_t)src * LD, j, t);
  store_row<L, C>((bd ? slot1 : slot0) + (size_t)s * LD, j, t);
  if (j == 0) {
    nlogpi[s] = logpi[src];
    ndelta[s] = delta[src];
    if (nstamp) nstamp[s] = stamp[src];                          /* blobs .= blobs[inds], smc:99 */
    wns[s] = 1.0 / (double)N;
    alive[s] = 1;
  }
}
__global__ __launch_bounds__(ABZ_BLOCK) void bits_flip_kernel(uint32_t* __restrict__ bits, uint32_t* __restrict__ other,
                                                              uint32_t nwords) {
  const uint32_t w = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  if (w < nwords) { const uint32_t v = ~bits[w]; bits[w] = v; other[w] = v; }
}
/* double-buffered rows: after a resampling every position's current slot is the one the gather wrote, 1 - P (P was read from
 * word 0 BEFORE this launch: it is passed by value) */
__global__ __launch_bounds__(ABZ_BLOCK) void bits_set_kernel(uint32_t* __restrict__ bits, uint32_t* __restrict__ other,
                                                             uint32_t nwords, const uint32_t* __restrict__ parity_src) {
  const uint32_t w = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  const uint32_t v = (parity_src[0] & 1u) ? 0xFFFFFFFFu : 0u;
  if (w < nwords) { bits[w] = v; other[w] = v; }
}
__global__ void bits_parity_kernel(const uint32_t* __restrict__ bits, uint32_t* __restrict__ out) { out[0] = (bits[0] & 1u) ^ 1u; }
template <int L, int C>
__global__ __launch_bounds__(ABZ_BLOCK) void packed_gather_kernel(const uint32_t* __restrict__ bits, uint32_t N,
                                                                  const double* __restrict__ slot0,
                                                                  const double* __restrict__ slot1,
                                                                  double* __restrict__ out) {
  constexpr int LD = L * C;
  const uint32_t gid = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  const uint32_t s = gid / L;
  const int j = (int)(gid % L);
  if (s >= N) return;
  const uint32_t b = (bits[s >> 5] >> (s & 31u)) & 1u;
  double t[C];
  load_row<L, C>((b ? slot1 : slot0) + (size_t)s * LD, j, t);
  store_row<L, C>(out + (size_t)s * LD, j, t);
}

int abz_launch_resample_gather_packed(abcdez_ctx* ctx, const uint32_t* inds, uint32_t N, uint32_t* bits, uint32_t* bits_other,
                                      double* slot0, double* slot1, const double* logpi, const double* delta, double* nlogpi,
                                      double* ndelta, double* wns, uint8_t* alive) {
  bool ok = abz_dispatch_lc(ctx->L, ctx->C, [&](auto LL, auto CC) {
    hipLaunchKernelGGL((resample_gather_packed_kernel<LL(), CC()>), dim3(abz_grid((uint64_t)N * LL())), dim3(ABZ_BLOCK), 0,
                       ctx->stream, inds, N, (const uint32_t*)bits, slot0, slot1, logpi, delta, nlogpi, ndelta, wns, alive,
                       (const uint64_t*)ctx->stamp_cur, ctx->stamp_cur ? ctx->stamp_nxt : nullptr);
  });
  if (!ok) { abz_set_error("resample_gather_packed: unsupported layout"); return -3; }
  const uint32_t nwords = (N + 31u) / 32u;
  if (ABZ_ROWS_DOUBLE_BUFFERED(ctx->h_model.ld)) {
    uint32_t* par = (uint32_t*)(ctx->d_scal + ABZ_S_PARITY);       /* 1 - P */
    hipLaunchKernelGGL(bits_parity_kernel, dim3(1), dim3(1), 0, ctx->stream, (const uint32_t*)bits, par);
    hipLaunchKernelGGL(bits_set_kernel, dim3(abz_grid(nwords)), dim3(ABZ_BLOCK), 0, ctx->stream, bits, bits_other, nwords,
                       (const uint32_t*)par);
  } else {
    hipLaunchKernelGGL(bits_flip_kernel, dim3(abz_grid(nwords)), dim3(ABZ_BLOCK), 0, ctx->stream, bits, bits_other, nwords);
  }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}
int abz_launch_packed_gather(abcdez_ctx* ctx, const uint32_t* bits, uint32_t N, const double* slot0, const double* slot1,
                             double* out) {
  bool ok = abz_dispatch_lc(ctx->L, ctx->C, [&](auto LL, auto CC) {
    hipLaunchKernelGGL((packed_gather_kernel<LL(), CC()>), dim3(abz_grid((uint64_t)N * LL())), dim3(ABZ_BLOCK), 0, ctx->stream,
                       bits, N, slot0, slot1, out);
  });
  if (!ok) { abz_set_error("packed_gather: unsupported layout"); return -3; }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

/* ---- T2 push_p over the population (result P, smc:382, mc:166) ---- */
__global__ __launch_bounds__(ABZ_BLOCK) void push_p_kernel(const abz_model* __restrict__ M,
                                                           const double* __restrict__ theta, uint64_t total,
                                                           double* __restrict__ out) {
  const uint64_t e = (uint64_t)blockIdx.x * ABZ_BLOCK + threadIdx.x;
  if (e >= total) return;
  const int k = (int)(e % (uint64_t)M->ld);
  out[e] = abz_push_p(&M->prior[k], theta[e]);
}

int abz_launch_push_p(abcdez_ctx* ctx, const double* theta, int64_t N, double* out) {
  const uint64_t total = (uint64_t)N * (uint64_t)ctx->h_model.ld;
  if (total == 0) return 0;
  hipLaunchKernelGGL(push_p_kernel, dim3(abz_grid(total)), dim3(ABZ_BLOCK), 0, ctx->stream, ctx->d_model, theta,
                     total, out);
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}
