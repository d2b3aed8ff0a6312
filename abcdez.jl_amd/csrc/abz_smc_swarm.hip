/*
 * abz_smc_swarm.hip -- one DE-Metropolis sweep over the alive particles of the packed population, and its replay on
 * the replicas of a sharded run.
 *
 * Replaces abcdesmc_swarm! (src/abcdez_smc.jl:106-153); the identity. copies of src/abcdez_smc.jl:337-340 have no
 * counterpart: a rejected particle writes nothing.  One fused kernel: donor draw -> proposal -> push_p -> prior
 * support / log-density -> simulator -> distance -> Metropolis accept -> accepted row to the position's other slot,
 * with the acceptance / simulation counters added per block.  The alive particles are the positions [0, n_alive)
 * (abcdez_smc_partition), so every wave is dense and the donors are addressed directly.
 *
 * HBM roofline: per update 3 rows of 8 ld bytes read (own row, two donor rows) + 16 B state, and for an
 * accepted proposal one row + 16 B written.  Everything that does not depend on loaded data (prior descriptors,
 * data vector, model scalars, sampler tables) is fetched before the first dependent load, so a wave waits on two
 * memory round trips: slot bits (an L2-resident bitmap) -> rows.
 */
#include "abz_dispatch.h"
#include "abz_kernels.h"

/* ================================================================ packed population (abz_kernels.h) */
#ifndef ABZ_SWEEP_WAVES
#define ABZ_SWEEP_WAVES_ATTR
#else
#define ABZ_SWEEP_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(ABZ_SWEEP_WAVES, ABZ_SWEEP_WAVES)))
#endif
template <int SIM, int L, int C, bool PLAIN>
__global__ __launch_bounds__((abz_sweep_block<SIM, L, C>())) ABZ_SWEEP_WAVES_ATTR void smc_swarm_packed_kernel(const SmcPackedArgs a) {
  smc_swarm_packed_body<SIM, L, C, PLAIN>(a);
}
/* Occupancy of the two-phase 4 x 8 kernel (d = 32): 81 registers, 27,000 B of LDS -- five waves per SIMD as the compiler leaves it.
 * Forcing six (-DABZ_SWEEP_WAVES=6: 80 registers, three dwords spilled in phase 2; the LDS budget allows it) was measured on two boxes
 * of the pool: -1.6 % on one, +5.6 % on the other (profiles/r05_two_phase_ab2.jsonl, r05_two_phase_ab3.jsonl) -- more rows in flight
 * is not uniformly better for random 256-byte reads, so the default stays. */
/* Lotka-Volterra: the sweep as two launches with the hand-over list between them (abz_kernels.h, smc_lv_phase1_body) */
template <bool PLAIN>
__global__ __launch_bounds__(ABZ_BLOCK) void smc_lv_phase1_kernel(const SmcPackedArgs a, const LvHandList h) {
  smc_lv_phase1_body<PLAIN, ABZ_BLOCK>(a, h);
}
template <bool PLAIN>
__global__ __launch_bounds__(ABZ_LV_BLOCK2) void smc_lv_phase2_kernel(const SmcPackedArgs a, const LvHandList h) {
  smc_lv_phase2_body<PLAIN, ABZ_LV_BLOCK2>(a, h);
}
template <int L, int C, bool PLAIN>
__global__ __launch_bounds__(ABZ_BLOCK) void smc_replay_packed_kernel(const SmcReplayPackedArgs a) {
  smc_replay_packed_body<L, C, PLAIN>(a);
}

int abz_launch_smc_swarm_packed(abcdez_ctx* ctx, const uint32_t* bits, uint32_t* bits_out, uint32_t n_alive, uint32_t r_lo,
                                uint32_t r_hi, double* slot0, double* slot1, double* logpi, double* delta, uint8_t* flags,
                                double eps, double gamma0, double gsig, uint32_t sweep, int want_counts,
                                const unsigned long long* stop) {
  SmcPackedArgs a;
  a.stop = stop;
  a.hm = ctx->hot; a.bits = bits; a.bits_out = bits_out; a.slot0 = slot0; a.slot1 = slot1; a.logpi = logpi; a.delta = delta;
  a.cslots = ctx->d_scal + ABZ_S_CSLOT0;
  a.c_cls = want_counts ? ABZ_C_NACC : ABZ_C_DISCARD;
  a.flags = flags; a.stamp = ctx->stamp_cur;
  a.eps = eps; a.gamma0 = gamma0; a.gsig = gsig;
  a.n_alive = n_alive; a.r_lo = r_lo; a.n_work = r_hi - r_lo; a.sweep = sweep;
  if (a.n_work == 0) return 0;
  a.rev = ctx->serpentine ? (uint32_t)(ctx->sweep_launch_seq++ & 1) : 0u;
  const int L = ctx->L, C = ctx->C;
  if (L > 8) { abz_set_error("smc_swarm_packed: at most 8 lanes per particle (a block must cover whole bitmap words)"); return -3; }
  unsigned nblocks = abz_grid((uint64_t)a.n_work * (uint64_t)L);
  /* two launches: phase 1 over the positions, phase 2 over the proposals it hands over -- every wavefront of the simulator full */
  const bool jit = ctx->user_module != nullptr;      /* this model's kernels were compiled at run time (abz_jit.hip) */
  const bool split = abz_sweep_in_two_launches(ctx) && (!jit || abz_jit_has_smc_split(ctx));
  LvHandList h{};
  if (split) {
    if (int rc = abz_lv_hand_reserve(ctx, (size_t)a.n_work)) return rc;      /* (a no-op after abcdez_ctx_reserve) */
    char* base = (char*)ctx->lv_hand;
    const size_t cap = ctx->lv_hand_cap;
    const unsigned par = (unsigned)(ctx->lv_seq++ & 1ull);       /* two counters in turn: this sweep's is zero (abz_lv_hand_reserve, then the sweeps) */
    h.count = (unsigned int*)base + par;
    h.count_next = (unsigned int*)base + (1u - par);
    h.tp = (double*)(base + 256);
    h.wl = h.tp + cap * (size_t)C; h.kdi = h.wl + cap; h.logu = h.kdi + cap;
    h.pos = (uint32_t*)(h.logu + cap);
  }
  const int tk = abz_time_begin(ctx);
  bool ok = true;
  if (jit) {
    if (split) {
      if (int rc = abz_jit_launch_smc_split(ctx, &a, &h, abz_grid((uint64_t)a.n_work))) return rc;
    } else {
      const unsigned blk = abz_jit_smc_block(ctx);
      nblocks = (unsigned)(((uint64_t)a.n_work * (uint64_t)L + blk - 1) / blk);
      if (int rc = abz_jit_launch_smc_packed(ctx, &a, nblocks)) return rc;
    }
  } else if (split) {
    {
    const unsigned nb1 = (unsigned)(((uint64_t)a.n_work + ABZ_BLOCK - 1) / ABZ_BLOCK);
    const unsigned nb2 = (unsigned)(((uint64_t)a.n_work + ABZ_LV_BLOCK2 - 1) / ABZ_LV_BLOCK2);
    if (ctx->prior_plain) {
      hipLaunchKernelGGL((smc_lv_phase1_kernel<true>), dim3(nb1), dim3(ABZ_BLOCK), 0, ctx->stream, a, h);
      hipLaunchKernelGGL((smc_lv_phase2_kernel<true>), dim3(nb2), dim3(ABZ_LV_BLOCK2), 0, ctx->stream, a, h);
    } else {
      hipLaunchKernelGGL((smc_lv_phase1_kernel<false>), dim3(nb1), dim3(ABZ_BLOCK), 0, ctx->stream, a, h);
      hipLaunchKernelGGL((smc_lv_phase2_kernel<false>), dim3(nb2), dim3(ABZ_LV_BLOCK2), 0, ctx->stream, a, h);
    }
    }
  } else {
    ok = abz_dispatch(ctx->h_model.sim_id, L, C, [&](auto S, auto LL, auto CC) {
      if constexpr (LL() <= 8) {
        constexpr unsigned BLK = (unsigned)abz_sweep_block<S(), LL(), CC()>();       /* 256 threads, 512 for Lotka-Volterra */
        const unsigned nb = (unsigned)(((uint64_t)a.n_work * (uint64_t)LL() + BLK - 1) / BLK);
        if (ctx->prior_plain)
          hipLaunchKernelGGL((smc_swarm_packed_kernel<S(), LL(), CC(), true>), dim3(nb), dim3(BLK), 0, ctx->stream, a);
        else
          hipLaunchKernelGGL((smc_swarm_packed_kernel<S(), LL(), CC(), false>), dim3(nb), dim3(BLK), 0, ctx->stream, a);
      }
    });
  }
  abz_time_end(ctx, tk, a.n_work);
  if (!ok) { abz_set_error("smc_swarm_packed: no kernel for this (simulator, ld, lanes) combination"); return -3; }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

/* After sweep k of a group: totals of the (nacc, nsim) slots -> snapshot k; the test of smc:352 on the group's
 * acceptances so far -> stop flag for the sweeps enqueued behind it.  One block. */
__global__ __launch_bounds__(ABZ_CSLOTS) void group_check_kernel(unsigned long long* __restrict__ scal, int k,
                                                                 unsigned long long base_acc, uint32_t n_alive, double kmin, int cls) {
  __shared__ unsigned long long s_a[ABZ_CSLOTS / 64], s_s[ABZ_CSLOTS / 64];
  const unsigned long long* cs = scal + ABZ_S_CSLOT0 + (size_t)threadIdx.x * ABZ_CSTRIDE;
  unsigned long long va = cs[cls], vs = cs[cls + 1];     /* (nacc, nsim) of the sweeps, or of the replays of a sharded run */
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { va += __shfl_xor(va, o); vs += __shfl_xor(vs, o); }
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = va; s_s[threadIdx.x >> 6] = vs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (k == 0 || scal[ABZ_S_GRP_STOP] == 0ull) {
      unsigned long long ta = 0, ts = 0;
      for (int w = 0; w < ABZ_CSLOTS / 64; ++w) { ta += s_a[w]; ts += s_s[w]; }
      scal[ABZ_S_GRP_SNAP + 2 * k] = ta;
      scal[ABZ_S_GRP_SNAP + 2 * k + 1] = ts;
      scal[ABZ_S_GRP_DONE] = (unsigned long long)(k + 1);
      scal[ABZ_S_GRP_STOP] = ((double)(ta - base_acc) / (double)n_alive >= kmin) ? 1ull : 0ull;   /* smc:352 */
    }
  }
}
int abz_launch_group_check(abcdez_ctx* ctx, int k, unsigned long long base_acc, uint32_t n_alive, double kmin, int cls) {
  hipLaunchKernelGGL(group_check_kernel, dim3(1), dim3(ABZ_CSLOTS), 0, ctx->stream, ctx->d_scal, k, base_acc, n_alive, kmin, cls);
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

int abz_launch_smc_replay_packed(abcdez_ctx* ctx, const uint32_t* bits, uint32_t* bits_out, uint32_t n_alive,
                                 uint32_t skip_lo, uint32_t skip_hi, double* slot0, double* slot1, double* logpi,
                                 const uint8_t* flags, double gamma0, double gsig, uint32_t sweep, const unsigned long long* stop) {
  SmcReplayPackedArgs a;
  a.stop = stop;
  a.hm = ctx->hot; a.bits = bits; a.bits_out = bits_out; a.flags = flags; a.slot0 = slot0; a.slot1 = slot1; a.logpi = logpi;
  a.stamp = ctx->stamp_cur;
  a.cslots = ctx->d_scal + ABZ_S_CSLOT0;
  a.gamma0 = gamma0; a.gsig = gsig; a.n_alive = n_alive; a.skip_lo = skip_lo; a.skip_hi = skip_hi; a.sweep = sweep;
  const unsigned nblocks = (unsigned)(((uint64_t)n_alive + ABZ_REPLAY_CHUNK - 1) / ABZ_REPLAY_CHUNK);
  if (ctx->user_module && abz_jit_has_replay(ctx)) return abz_jit_launch_replay(ctx, &a, nblocks);
  bool ok = abz_dispatch_lc(ctx->L, ctx->C, [&](auto LL, auto CC) {
    if (ctx->prior_plain)
      hipLaunchKernelGGL((smc_replay_packed_kernel<LL(), CC(), true>), dim3(nblocks), dim3(ABZ_BLOCK), 0, ctx->stream, a);
    else
      hipLaunchKernelGGL((smc_replay_packed_kernel<LL(), CC(), false>), dim3(nblocks), dim3(ABZ_BLOCK), 0, ctx->stream, a);
  });
  if (!ok) { abz_set_error("smc_replay_packed: unsupported layout"); return -3; }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}
