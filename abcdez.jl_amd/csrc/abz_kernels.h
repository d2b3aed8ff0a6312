/*
 * abz_kernels.h -- bodies of the simulator-dependent kernels (initial population, SMC sweep + replay, MC sweep)
 * as __device__ function templates.  The library instantiates them for the built-in simulators
 * (abz_init.hip, abz_smc_swarm.hip, abz_mc_swarm.hip); abz_jit.hip compiles the same text with
 * hiprtc around a user-supplied abz_user_dist.
 */
#ifndef ABZ_KERNELS_H
#define ABZ_KERNELS_H

#include "abz_device.h"

#define ABZ_LV_BLOCK 512                                  /* threads per workgroup of the Lotka-Volterra sweep (abz_sweep_block) */
#define ABZ_REPLAY_PER 8                                  /* alive ranks per thread in the scan phase of the replay kernels */
#define ABZ_REPLAY_CHUNK (ABZ_BLOCK * ABZ_REPLAY_PER)     /* alive ranks per block */

/* ================================================================ S1: abcde_init! (src/abcdez_init.jl:2-22) */
#define ABZ_MAX_RETRY 100000u

template <int SIM, int L, int C>
__device__ inline void init_kernel_body(const HotModel& M, double* __restrict__ theta,
                                                         double* __restrict__ logpi, double* __restrict__ delta,
                                                         uint32_t i0, uint32_t n, unsigned long long* __restrict__ bad,
                                                         uint64_t* __restrict__ stamp = nullptr) {
  constexpr int LD = L * C;
  constexpr uint32_t PB = ABZ_BLOCK / L;
  __shared__ ModelLds<LD> s_model;
  {
    ModelStage<SIM, LD> stage;
    stage.load(M);
    stage.store(s_model);
  }
  __syncthreads();
  const abz_prior_dim* pd = s_model.prior;
  const int j = (int)(threadIdx.x % L);
  const uint64_t seed = M.seed;
  const uint32_t ntiles = (n + PB - 1) / PB;
  ABZ_TILE_LOOP(tile, ntiles) {
    const uint32_t grp = tile * PB + threadIdx.x / L;
    if (grp >= n) continue;                     /* whole groups leave together; no barrier inside the loop */
    const uint32_t i = i0 + grp;
    double th[C], pp[C];
    double lp, dl;
    uint32_t retry = 0;
    for (;;) {
      if constexpr (C == 1) {
        const abz_u64x2 w = abz_rng(seed, i, retry, 0, ABZ_RNG_INIT_PRIOR);
        double z0, z1;
        abz_normal_pair(w, &s_model.tab, &z0, &z1);
        th[0] = abz_prior_draw1(&pd[0], w.w0, z0);
        if (pd[0].family >= ABZ_PRIOR_BETA) th[0] = abz_prior_draw_extx(&pd[0], seed, i, retry, 0u, &s_model.tab, M.ext);
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
            th[2 * m] = abz_prior_draw_extx(&pd[k], seed, i, retry, (uint32_t)k, &s_model.tab, M.ext);
          if (pd[k + 1].family >= ABZ_PRIOR_BETA)
            th[2 * m + 1] = abz_prior_draw_extx(&pd[k + 1], seed, i, retry, (uint32_t)(k + 1), &s_model.tab, M.ext);
        }
      }
      if (M.mv) {                     /* correlated Normal prior: the row drawn so far is z ~ N(0, I); theta = mu + L z */
        double zz[C];
#pragma unroll
        for (int q = 0; q < C; ++q) zz[q] = th[q];
        group_lower_matvec<L, C>(M.mv + LD + (size_t)LD * LD, j, zz, th);
#pragma unroll
        for (int q = 0; q < C; ++q) {
          const int k = Lay<L, C>::comp(j, q / 2, q & 1);
          th[q] = k < M.d ? M.mv[k] + th[q] : 0.0;
        }
      }
      lp = group_logprior<L, C>(pd, j, th, pp, M.mv, M.ext);
      dl = ABZ_NAN;
      if (abz_isfinite(lp)) dl = sim_dist<SIM, L, C>(M, &s_model.tab, j, pp, s_model.y, i, retry, ABZ_RNG_INIT_SIM);   /* init.jl:9-13,17 */
      if (abz_isfinite(dl) && abz_isfinite(lp)) break;                                          /* init.jl:14 */
      if (++retry >= ABZ_MAX_RETRY) {
        if (j == 0) atomicAdd(bad, 1ull);
        break;
      }
    }
    store_row<L, C>(theta + (size_t)i * LD, j, th);
    if (j == 0) {
      logpi[i] = lp; delta[i] = dl;
      if (stamp) stamp[i] = abz_stamp(i, retry, 1);              /* which simulator call made this distance (blobs) */
    }
  }
}

/* ================================================================ S2+S3: abcdesmc_swarm! (src/abcdez_smc.jl:106-153) on the PACKED population
 * The alive particles are the positions [0, n_alive) (abcdez_smc_partition keeps them a prefix), so "alive rank r" is
 * "position r": the own row streams in, the two donors are addressed directly, and the only indirection left is ONE
 * BIT per position -- which of its two row slots is current -- in a bitmap of N / 8 bytes (512 KB at N = 2^22: resident
 * in every XCD's L2, so the donor look-ups cost no HBM / Infinity-Cache traffic).  An accepted proposal is written to
 * the position's other slot and its bit flips in bits_out (the sweep is synchronous: donors read bits / rows of the
 * generation before, smc:337-350); a rejected one writes nothing; log-prior and distance are updated in place.      */
struct SmcPackedArgs {
  HotModel hm;
  const uint32_t* bits;         /* current slot of every position, 32 positions per word */
  uint32_t* bits_out;
  double* slot0;
  double* slot1;
  double* logpi;                /* in place */
  double* delta;
  unsigned long long* cslots;   /* cumulative counter slots; (nacc, nsim) go to classes c_cls, c_cls + 1 */
  uint8_t* flags;               /* per position: bit 0 accepted, bit 1 simulated (sharded runs; may be NULL) */
  uint64_t* stamp;              /* blob stamps, in place; NULL when blobs are off */
  const unsigned long long* stop;   /* group of sweeps: non-zero = the early exit of smc:352 held before this sweep; NULL = always run */
  double eps, gamma0, gsig;
  uint32_t n_alive, r_lo, n_work, sweep, c_cls;
  uint32_t rev;                 /* non-zero: workgroup b owns tile gridDim.x - 1 - b (serpentine order, abz_smc_swarm.hip) */
};

__device__ inline uint32_t packed_bit(const uint32_t* __restrict__ bits, uint32_t p) { return (bits[p >> 5] >> (p & 31u)) & 1u; }

template <int SIM, int L, int C, bool PLAIN = false>
__device__ inline void smc_swarm_packed_body_1p(const SmcPackedArgs& a) {
  constexpr int LD = L * C;
  constexpr int PB = ABZ_BLOCK / L;                 /* positions per block: whole words of the bitmap */
  static_assert(PB % 32 == 0, "packed sweeps need at least 32 particles per block (lanes <= 8)");
  const HotModel& M = a.hm;
  if (a.stop && *a.stop) return;                    /* grid-uniform: written by the kernel before this one */
  /* Which tile a workgroup owns changes no result (everything is keyed by position).  Workgroups are dispatched in index
   * order, so rev walks the prefix from its end: the rows the sweep before read LAST are the ones this sweep reads FIRST, while
   * they still sit in the 256 MiB Infinity Cache (launcher: abz_launch_smc_swarm_packed). */
  const uint32_t tile = a.rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
  const uint32_t gid = tile * ABZ_BLOCK + threadIdx.x;
  const uint32_t grp = gid / L;
  const int j = (int)(gid % L);
  const bool active = grp < a.n_work;
  const uint32_t ri = a.r_lo + (active ? grp : 0u);

  __shared__ ModelLds<LD> s_model;
  __shared__ uint32_t s_acc[PB / 32];

  /* Order of issue = order of need.  Nothing below waits for the model tables before the rows are on their way:
   *   slot bit of the own position | Philox words -> donor positions (smc:119-126) -> their slot bits   (one L2 round trip)
   *   the three rows, log-prior, distance                                                                (one HBM round trip)
   *   meanwhile: tables staged in LDS, gamma = gamma0 (1 + randn gamma_sigma) (smc:128), log(rand) (smc:145)              */
  /* Rows of at most two doubles are DOUBLE-BUFFERED (ABZ_ROWS_DOUBLE_BUFFERED): a sweep writes every swept position's row
   * to its other slot (the proposal, or a copy of the row) and flips every swept bit, so the alive prefix always shares ONE
   * slot parity and the donors need no bit look-up -- at 8 or 16 bytes per row the two random 4-byte look-ups cost as
   * much as the donor rows themselves (the d = 1 sweep is bound by the gather rate of the CU's address unit), and copying a
   * rejected row is a coalesced 8 bytes. */
  constexpr bool DBUF = ABZ_ROWS_DOUBLE_BUFFERED(LD);
  ModelStage<SIM, LD> stage;
  stage.load(M);
  const uint32_t wi = a.bits[ri >> 5];
  ParticleDraws<L> draws;
  uint32_t ra, rb;
  draws.words(M.seed, ri, a.sweep, j, a.n_alive, ri, &ra, &rb);
  uint32_t wa = 0u, wb = 0u;
  if constexpr (!DBUF) { wa = a.bits[ra >> 5]; wb = a.bits[rb >> 5]; }
  const double lpi = a.logpi[ri];
  const double dli = a.delta[ri];
  if (threadIdx.x < PB / 32) s_acc[threadIdx.x] = 0u;
  const uint32_t bi = (wi >> (ri & 31u)) & 1u;
  const uint32_t ba = DBUF ? bi : (wa >> (ra & 31u)) & 1u, bb = DBUF ? bi : (wb >> (rb & 31u)) & 1u;
  double ti[C], ta[C], tb[C];
  load_row<L, C>((bi ? a.slot1 : a.slot0) + (size_t)ri * LD, j, ti);
  load_row<L, C>((ba ? a.slot1 : a.slot0) + (size_t)ra * LD, j, ta);
  load_row<L, C>((bb ? a.slot1 : a.slot0) + (size_t)rb * LD, j, tb);
  stage.store(s_model);
  __syncthreads();                                                /* sampler + model tables staged */
  double g, log_u;
  draws.finish(&s_model.tab, a.gamma0, a.gsig, &g, &log_u);

  double tp[C], pp[C];
#pragma unroll
  for (int q = 0; q < C; ++q) tp[q] = ti[q] + (ta[q] - tb[q]) * g;

  const double lp = group_logprior<L, C, PLAIN>(s_model.prior, j, tp, pp, M.mv, M.ext);   /* smc:134 */
  const bool insupport = !(lp == ABZ_NINF);                       /* smc:135 */
  bool acc = false;
  double dp = dli;
  /* Narrow rows with a Normal prior (in support for every finite proposal): the simulator is not put behind a branch --
   * its random numbers do not depend on the proposal and can be produced while the rows are in flight, which shortens
   * the dependent chain these latency-bound kernels run on; the result is used only when the proposal is in support,
   * as in the branch.  Wide rows keep the branch: hoisting costs the d = 32 kernel its fifth wave (88 -> 99 VGPRs). */
  constexpr bool UNBRANCH = PLAIN && LD <= 4;
  if (UNBRANCH || insupport) {
    const double ds = sim_dist<SIM, L, C, false, PLAIN>(M, &s_model.tab, j, pp, s_model.y, ri, a.sweep, ABZ_RNG_SIM);   /* smc:137 */
    const double w = ((lp - lpi) + kernel_logpdf_dev(M.abck, a.eps, ds)) - kernel_logpdf_dev(M.abck, a.eps, dli);  /* smc:140-141, left to right as the reference */
    if (insupport) {
      dp = ds;
      acc = (0.0 <= w) || (log_u < w);                            /* smc:145 */
    }
  }
  acc = acc && active;
  if constexpr (DBUF) {
    if (active) {
      double to[C];
#pragma unroll
      for (int q = 0; q < C; ++q) to[q] = acc ? tp[q] : ti[q];
      store_row<L, C>((bi ? a.slot0 : a.slot1) + (size_t)ri * LD, j, to);
      if (j == 0) atomicOr(&s_acc[(threadIdx.x / L) >> 5], 1u << ((threadIdx.x / L) & 31u));
    }
  } else if (acc) {                                               /* smc:146-150 */
    store_row<L, C>((bi ? a.slot0 : a.slot1) + (size_t)ri * LD, j, tp);
    if (j == 0) atomicOr(&s_acc[(threadIdx.x / L) >> 5], 1u << ((threadIdx.x / L) & 31u));
  }
  if (acc && j == 0) {
    a.logpi[ri] = lp; a.delta[ri] = dp;
    if (a.stamp) a.stamp[ri] = abz_stamp(ri, a.sweep, 0);
  }
  if (active && j == 0 && a.flags) a.flags[ri] = (uint8_t)((acc ? 1 : 0) | (insupport ? 2 : 0));
  block_count2((j == 0 && acc) ? 1u : 0u, (active && j == 0 && insupport) ? 1u : 0u, a.cslots, a.c_cls);      /* (its barrier publishes s_acc) */
  if (threadIdx.x < PB / 32) {
    const uint32_t w = (a.r_lo + tile * PB) / 32u + threadIdx.x;
    if (w * 32u < a.r_lo + a.n_work) a.bits_out[w] = a.bits[w] ^ s_acc[threadIdx.x];
  }
}

/* ---- the same sweep in TWO PHASES, for rows spread over several lanes (the d = 32 kernel of BASELINE configs[2]).
 *
 * smc:137-145 simulate every in-support proposal and then accept iff  0 <= w  or  log(rand) < w,
 *     w = ((lp - lpi) + K(dp)) - K(di)          (smc:140-141, evaluated left to right)
 * K = logpdf of the ABC kernel, never positive (types.jl:26-73).  Rounding is monotone, so w <= w_max = ((lp - lpi) + 0) - K(di)
 * for EVERY distance the simulator could return -- and when neither `0 <= w_max` nor `log(rand) < w_max` holds, the proposal is
 * rejected whatever dist! returns: its call cannot change any output (the accept decision; nsims counts in-support proposals,
 * smc:138; a blob is kept on acceptance only, smc:148; random numbers are addressed by counter, so nothing shifts).  With a
 * Normal prior in 32 dimensions that is three proposals out of four (profiles/HISTORY.md, round 5), and the simulator -- four
 * Philox blocks, four Box-Muller pairs, the distance -- is two thirds of the sweep's vector instructions.
 *
 * A skipped simulation only saves issue cycles when WHOLE WAVES skip it, hence two phases with a compaction in between:
 *   phase 1, every lane group: slot bits -> rows -> proposal (smc:128) -> log-prior (smc:134) -> support (smc:135) -> w_max;
 *            the proposals that may still be accepted are handed over through LDS (row + five scalars), packed densely;
 *   phase 2, lane group k takes hand-over slot k: simulator + distance (smc:137) -> w -> accept (smc:145) -> row to the other slot.
 *            Wavefronts whose groups all lie beyond the number of survivors wait at the workgroup's barrier and issue nothing.
 * Same results bit for bit as the one-phase body (and the oracle, which simulates every proposal as the reference does). */
template <int L, int C, int PB>
struct SweepHand {
  double tp[PB][L * C];                 /* proposal rows, 16-byte units swizzled by slot (hand_unit) */
  double wl[PB], kdi[PB], logu[PB];             /* lp - lpi, K(di), log(rand) of smc:140-145 (lp itself is re-evaluated in phase 2) */
  uint16_t pos[PB];                             /* position inside the tile | own slot bit << 15 */
};
/* LDS of the two-phase d = 32 kernel: 7 KB sampler tables + 1.8 KB model + 16 KB rows + 1.6 KB scalars = 27,000 B: room for six
 * workgroups per CU (163,840 / 6 = 27,306) -- the hand-over carries no more than it must */
template <int L, int C>
__device__ inline int hand_unit(int slot, int m, int j) {   /* 16-byte unit of (load m, lane j) inside hand-over row `slot` */
  constexpr int MM = C / 2;
  return ((m ^ (slot & (MM - 1))) * L) + j;                 /* neighbouring slots start in different LDS bank quarters */
}
template <int L, int C>
__device__ inline void group_push_p(const abz_prior_dim* pd, int j, const double (&p)[C], double (&pp)[C]) {
#pragma unroll
  for (int q = 0; q < C; ++q) pp[q] = abz_push_p(&pd[Lay<L, C>::comp(j, q / 2, q & 1)], p[q]);
}

template <int SIM, int L, int C, bool PLAIN = false, int BLOCK = ABZ_BLOCK>
__device__ inline void smc_swarm_packed_body_2p(const SmcPackedArgs& a) {
  constexpr int LD = L * C;
  constexpr int PB = BLOCK / L;
  constexpr int GW = 64 / L;                        /* lane groups per wavefront */
  static_assert(PB % 32 == 0 && L >= 1 && L <= 8 && C >= 2 && (C & 1) == 0 && !ABZ_ROWS_DOUBLE_BUFFERED(LD),
                "two-phase sweep: 1 <= lanes <= 8, an even number of components per lane, rows of more than two doubles");
  const HotModel& M = a.hm;
  if (a.stop && *a.stop) return;                    /* grid-uniform: written by the kernel before this one */
  const uint32_t tile = a.rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x;      /* serpentine order: smc_swarm_packed_body_1p */
  const uint32_t gid = tile * BLOCK + threadIdx.x;
  const uint32_t grp = gid / L;
  const int j = (int)(gid % L);
  const bool active = grp < a.n_work;
  const uint32_t tile_base = a.r_lo + tile * (uint32_t)PB;
  const uint32_t ri = a.r_lo + (active ? grp : 0u);

  __shared__ ModelLds<LD> s_model;
  __shared__ SweepHand<L, C, PB> s_hand;
  __shared__ uint32_t s_acc[PB / 32], s_ins[PB / 32];    /* per position of the tile: accepted / in support (bit masks) */
  __shared__ uint32_t s_n;

  /* ---------------- phase 1: order of issue = order of need (smc_swarm_packed_body_1p) */
  ModelStage<SIM, LD, BLOCK> stage;
  stage.load(M);
  const uint32_t wi = a.bits[ri >> 5];
  ParticleDraws<L> draws;
  uint32_t ra, rb;
  draws.words(M.seed, ri, a.sweep, j, a.n_alive, ri, &ra, &rb);
  const uint32_t wa = a.bits[ra >> 5], wb = a.bits[rb >> 5];
  const double lpi = a.logpi[ri];
  const double dli = a.delta[ri];
  if (threadIdx.x < PB / 32) { s_acc[threadIdx.x] = 0u; s_ins[threadIdx.x] = 0u; }
  if (threadIdx.x == 0) s_n = 0u;
  const uint32_t bi = (wi >> (ri & 31u)) & 1u, ba = (wa >> (ra & 31u)) & 1u, bb = (wb >> (rb & 31u)) & 1u;
  double tp[C];
  {
    double ti[C], ta[C], tb[C];
    load_row<L, C>((bi ? a.slot1 : a.slot0) + (size_t)ri * LD, j, ti);
    load_row<L, C>((ba ? a.slot1 : a.slot0) + (size_t)ra * LD, j, ta);
    load_row<L, C>((bb ? a.slot1 : a.slot0) + (size_t)rb * LD, j, tb);
    stage.store(s_model);
    __syncthreads();                                              /* sampler + model tables staged; s_acc, s_n zeroed */
    double g, log_u;
    draws.finish(&s_model.tab, a.gamma0, a.gsig, &g, &log_u);
#pragma unroll
    for (int q = 0; q < C; ++q) tp[q] = ti[q] + (ta[q] - tb[q]) * g;              /* smc:128 */
    double pp[C];
    const double lp = group_logprior<L, C, PLAIN>(s_model.prior, j, tp, pp, M.mv, M.ext);   /* smc:134 */
    const bool insupport = !(lp == ABZ_NINF);                     /* smc:135 */
    const double kdi = kernel_logpdf_dev(M.abck, a.eps, dli);
    const double wl = lp - lpi;
    const double w_max = (wl + 0.0) - kdi;                        /* smc:140-141 with K(dp) at its maximum */
    const bool may = active && insupport && ((0.0 <= w_max) || (log_u < w_max));
    /* compaction: the wave's surviving groups take consecutive hand-over slots from a workgroup counter */
    const unsigned lane = threadIdx.x & 63u;
    const unsigned long long mk = __ballot(may && j == 0);
    unsigned int base = 0u;
    if (lane == 0u && mk) base = atomicAdd(&s_n, (unsigned)__popcll(mk));
    base = __shfl(base, 0, 64);
    const int slot1 = (int)(base + (unsigned)__popcll(mk & ((1ull << (lane - (unsigned)j)) - 1ull)));
    if (may) {
      const int slot = slot1;
      double2* row = reinterpret_cast<double2*>(s_hand.tp[slot]);
#pragma unroll
      for (int m = 0; m < C / 2; ++m) { double2 t; t.x = tp[2 * m]; t.y = tp[2 * m + 1]; row[hand_unit<L, C>(slot, m, j)] = t; }
      if (j == 0) {
        s_hand.wl[slot] = wl; s_hand.kdi[slot] = kdi; s_hand.logu[slot] = log_u;
        s_hand.pos[slot] = (uint16_t)((ri - tile_base) | (bi << 15));
      }
    }
    if (active && j == 0 && insupport && a.flags) atomicOr(&s_ins[(threadIdx.x / L) >> 5], 1u << ((threadIdx.x / L) & 31u));
    /* nsims counts the in-support proposals (smc:138), simulated here or not */
    const unsigned nsim1 = (active && j == 0 && insupport) ? 1u : 0u;
    __syncthreads();                                              /* hand-over complete */

    /* ---------------- phase 2: lane group k takes slot k */
    const unsigned n = s_n;
    const unsigned sg = threadIdx.x / L;
    bool acc = false;
    if constexpr (SIM == ABZ_SIM_LV && L == 1) {
      /* Lotka-Volterra: the distance is a running sum of squared errors over the observations (abz_device.h, lv_observe), so it
       * only grows -- once it has passed eps^2 the proposal is rejected whatever the rest of the trajectory does (every ABC kernel
       * is zero beyond eps, types.jl:26-73; K(dp) = -Inf makes w -Inf or NaN, smc:140-145), and the 100 RK4 steps to the next
       * observation need not be made.  As with the skipped simulator calls this only pays when whole wavefronts stop, so phase 2 runs
       * ROUND BY ROUND: one observation + one interval per round for the proposals still alive, which are re-packed into the
       * leading lanes after every round.  State lives in LDS by hand-over slot; a proposal that survives every round has had
       * exactly sim_dist's operations in sim_dist's order. */
      __shared__ double s_lx[PB], s_ly[PB], s_lacc[PB];
      __shared__ uint16_t s_list[2][PB];
      __shared__ unsigned int s_live[3];
      const LvConst k = lv_const(M);
      /* certain rejection: acc >= bound > eps^2 (1 + 2^-41) => sqrt(acc) > eps for the strict and the non-strict kernels alike;
       * eps = Inf or 0: never (bound NaN): dp = Inf is in the support of Indicator0toeps(Inf), dp = 0 in that of Indicator0toeps(0) */
      const double bound = (a.eps > 0.0 && a.eps < 1.0e300) ? (a.eps * a.eps) * (1.0 + 0x1p-40) : ABZ_NAN;
      if (threadIdx.x < n) { s_list[0][threadIdx.x] = (uint16_t)threadIdx.x; s_lx[threadIdx.x] = M.sim_p[0]; s_ly[threadIdx.x] = M.sim_p[1]; s_lacc[threadIdx.x] = 0.0; }
      if (threadIdx.x < 3) s_live[threadIdx.x] = 0u;
      __syncthreads();
      unsigned n_live = n;
      int cur = 0;
      const unsigned lane = threadIdx.x & 63u, wave0 = threadIdx.x & ~63u;
      /* ABZ_LV_ROUND observations (and the intervals behind them) per round: a round ends with a workgroup barrier, at which the
       * working wavefronts wait for the slowest of them -- measured: 1 per round costs 4 % of the sweep where few proposals leave
       * early (profiles/r05_lv_early_exit_ab.jsonl) */
#define ABZ_LV_ROUND 2
      int round = 0;
      for (int jo0 = 0; jo0 < k.nobs; jo0 += ABZ_LV_ROUND, ++round) {
        if (wave0 < n_live) {                                      /* wave-uniform: this wavefront still has proposals */
          const bool on = threadIdx.x < n_live;
          const unsigned sl = s_list[cur][on ? threadIdx.x : 0u];  /* idle lanes of a working wave shadow the first proposal */
          const double2* row = reinterpret_cast<const double2*>(s_hand.tp[sl]);
          double tq[C], pq[C];
#pragma unroll
          for (int m = 0; m < C / 2; ++m) { const double2 t = row[hand_unit<L, C>((int)sl, m, 0)]; tq[2 * m] = t.x; tq[2 * m + 1] = t.y; }
          group_push_p<L, C>(s_model.prior, 0, tq, pq);
          double x = s_lx[sl], y = s_ly[sl], dsum = s_lacc[sl];
          const uint32_t rs = tile_base + (s_hand.pos[sl] & 0x7FFFu);
          bool dead = false;
          for (int jo = jo0; jo < jo0 + ABZ_LV_ROUND && jo < k.nobs; ++jo) {
            lv_observe<false>(M, &s_model.tab, k, rs, a.sweep, ABZ_RNG_SIM, jo, x, y, dsum, nullptr);
            dead = dead || (dsum >= bound);                        /* (false for a NaN sum: it stays, and is rejected at the end) */
            if (jo + 1 < k.nobs) lv_advance(k, pq[0], pq[1], pq[2], pq[3], x, y);
          }
          const bool keep = on && !dead;
          if (keep) { s_lx[sl] = x; s_ly[sl] = y; s_lacc[sl] = dsum; }
          const unsigned long long mk = __ballot(keep);
          unsigned int base = 0u;
          if (lane == 0u && mk) base = atomicAdd(&s_live[round % 3], (unsigned)__popcll(mk));
          base = __shfl(base, 0, 64);
          if (keep) s_list[1 - cur][base + (unsigned)__popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)sl;
        }
        __syncthreads();
        n_live = s_live[round % 3];
        if (threadIdx.x == 0) s_live[(round + 2) % 3] = 0u;        /* the counter of the round after next (last read a round ago) */
        cur = 1 - cur;
      }
      if (wave0 < n_live) {                                        /* the proposals whose distance stayed below the bound to the end */
        const bool on = threadIdx.x < n_live;
        const unsigned sl = s_list[cur][on ? threadIdx.x : 0u];
        const double2* row = reinterpret_cast<const double2*>(s_hand.tp[sl]);
        double tq[C], pq[C];
#pragma unroll
        for (int m = 0; m < C / 2; ++m) { const double2 t = row[hand_unit<L, C>((int)sl, m, 0)]; tq[2 * m] = t.x; tq[2 * m + 1] = t.y; }
        const double lps = group_logprior<L, C, PLAIN>(s_model.prior, 0, tq, pq, M.mv, M.ext);
        const uint32_t pw = s_hand.pos[sl];
        const uint32_t rs = tile_base + (pw & 0x7FFFu), bs = pw >> 15;
        const double ds = abz_sqrt(s_lacc[sl]);                                                                    /* smc:137 */
        const double w = (s_hand.wl[sl] + kernel_logpdf_dev(M.abck, a.eps, ds)) - s_hand.kdi[sl];                  /* smc:140-141 */
        acc = on && ((0.0 <= w) || (s_hand.logu[sl] < w));         /* smc:145 */
        if (acc) {                                                 /* smc:146-150 */
          store_row<L, C>((bs ? a.slot0 : a.slot1) + (size_t)rs * LD, 0, tq);
          const uint32_t t = rs - tile_base;
          atomicOr(&s_acc[t >> 5], 1u << (t & 31u));
          a.logpi[rs] = lps; a.delta[rs] = ds;
          if (a.stamp) a.stamp[rs] = abz_stamp(rs, a.sweep, 0);
        }
      }
    } else
    if ((threadIdx.x >> 6) * (unsigned)GW < n) {                  /* wave-uniform: this wavefront has at least one slot */
      const bool on = sg < n;
      const int slot = on ? (int)sg : 0;                          /* idle groups of a working wave shadow slot 0: shuffles stay converged */
      const double2* row = reinterpret_cast<const double2*>(s_hand.tp[slot]);
      double tq[C], pq[C];
#pragma unroll
      for (int m = 0; m < C / 2; ++m) { const double2 t = row[hand_unit<L, C>(slot, m, j)]; tq[2 * m] = t.x; tq[2 * m + 1] = t.y; }
      /* push_p (types.jl:20-23) and the log-prior again: the same function of the same row in the same lanes as in phase 1 -- the
       * same bits -- for 45 instructions of the few wavefronts that get here, instead of 512 bytes of LDS in every workgroup */
      const double lps = group_logprior<L, C, PLAIN>(s_model.prior, j, tq, pq, M.mv, M.ext);
      const uint32_t pw = s_hand.pos[slot];
      const uint32_t rs = tile_base + (pw & 0x7FFFu), bs = pw >> 15;
      const double ds = sim_dist<SIM, L, C, false, PLAIN>(M, &s_model.tab, j, pq, s_model.y, rs, a.sweep, ABZ_RNG_SIM);   /* smc:137 */
      const double w = (s_hand.wl[slot] + kernel_logpdf_dev(M.abck, a.eps, ds)) - s_hand.kdi[slot];              /* smc:140-141 */
      acc = on && ((0.0 <= w) || (s_hand.logu[slot] < w));        /* smc:145 */
      if (acc) {                                                  /* smc:146-150 */
        /* the accepted row is not read again before the next sweep: non-temporal stores (-0.8 % / -1.8 % on the sweep on two boxes
         * against plain stores, profiles/r05_two_phase_ab2.jsonl, _ab3.jsonl) */
        {
          typedef double d2v __attribute__((ext_vector_type(2)));
          double* dst = (bs ? a.slot0 : a.slot1) + (size_t)rs * LD;
#pragma unroll
          for (int m = 0; m < C / 2; ++m) { d2v t; t.x = tq[2 * m]; t.y = tq[2 * m + 1]; __builtin_nontemporal_store(t, (d2v*)(dst + m * 2 * L + 2 * j)); }
        }
        if (j == 0) {
          const uint32_t t = rs - tile_base;
          atomicOr(&s_acc[t >> 5], 1u << (t & 31u));
          a.logpi[rs] = lps; a.delta[rs] = ds;
          if (a.stamp) a.stamp[rs] = abz_stamp(rs, a.sweep, 0);
        }
      }
    }
    block_count2<BLOCK>((j == 0 && acc) ? 1u : 0u, nsim1, a.cslots, a.c_cls);      /* (its barrier publishes s_acc) */
  }
  if (threadIdx.x < PB / 32) {
    const uint32_t w = tile_base / 32u + threadIdx.x;
    if (w * 32u < a.r_lo + a.n_work) a.bits_out[w] = a.bits[w] ^ s_acc[threadIdx.x];
  }
  if (a.flags && threadIdx.x < PB && tile_base + threadIdx.x < a.r_lo + a.n_work)      /* sharded runs: bit 0 accepted, bit 1 simulated */
    a.flags[tile_base + threadIdx.x] = (uint8_t)(((s_acc[threadIdx.x >> 5] >> (threadIdx.x & 31u)) & 1u) |
                                                 (((s_ins[threadIdx.x >> 5] >> (threadIdx.x & 31u)) & 1u) << 1));
}

/* ---- Lotka-Volterra (BASELINE configs[3]): the two phases as TWO KERNELS, the hand-over through global memory.
 *
 * Inside one kernel (above) a workgroup's phase 2 occupies only the wavefronts its own survivors fill -- a third of the tile's
 * proposals at the start of a run, fewer later -- while the other wavefronts of the workgroup wait at its barriers: 1500 dependent RK4
 * steps per proposal run at a third of the occupancy the registers allow.  Here phase 1 appends every proposal that may still be
 * accepted to ONE list (64 bytes per record: the proposal row, lp - lpi, K(di), log(rand), the position), and phase 2 is a launch of
 * its own over that list: every workgroup starts with all its lanes on a proposal.  Random numbers are addressed by position
 * and every output is keyed by position, so the order of the list (an atomic counter's) changes nothing: the same bits as the one-kernel
 * body, the one-phase body and the oracle.  Phase 1 leaves bits_out = bits and the flag bytes without the accepted bit; phase 2
 * flips / sets them for the proposals it accepts. */
struct LvHandList {
  double* tp;               /* [cap][C] proposal rows (C = 4: Lotka-Volterra; 4, 8 or 16: user-supplied simulators) */
  double* wl;               /* lp - lpi */
  double* kdi;              /* K(di) */
  double* logu;             /* log(rand) of smc:145 */
  uint32_t* pos;            /* position | own slot bit << 31 */
  unsigned int* count;      /* records in the list: zero when phase 1 starts */
  unsigned int* count_next; /* the counter of the NEXT sweep (the two alternate): phase 2 zeroes it, whether or not the sweep runs */
};
#define ABZ_LV_BLOCK2 256   /* threads per workgroup of the second launch */

/* phase 1 for any simulator that runs one lane per particle with rows of C = 4, 8 or 16 doubles (Lotka-Volterra; user-supplied simulators) */
template <int SIM, int C, bool PLAIN, int BLOCK = ABZ_BLOCK>
__device__ inline void smc_split_phase1_body(const SmcPackedArgs& a, const LvHandList& h) {
  constexpr int L = 1, LD = C, PB = BLOCK;
  static_assert(C >= 2 && (C & 1) == 0 && !ABZ_ROWS_DOUBLE_BUFFERED(LD), "two-launch sweep: rows of 4, 8 or 16 doubles");
  const HotModel& M = a.hm;
  if (a.stop && *a.stop) return;                    /* grid-uniform: written by the kernel before this one */
  const uint32_t tile = a.rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
  const uint32_t grp = tile * BLOCK + threadIdx.x;
  const bool active = grp < a.n_work;
  const uint32_t tile_base = a.r_lo + tile * (uint32_t)PB;
  const uint32_t ri = a.r_lo + (active ? grp : 0u);

  __shared__ ModelLds<LD> s_model;
  __shared__ uint32_t s_ins[PB / 32];
  __shared__ uint32_t s_n, s_base;

  ModelStage<SIM, LD, BLOCK> stage;
  stage.load(M);
  const uint32_t wi = a.bits[ri >> 5];
  ParticleDraws<L> draws;
  uint32_t ra, rb;
  draws.words(M.seed, ri, a.sweep, 0, a.n_alive, ri, &ra, &rb);
  const uint32_t wa = a.bits[ra >> 5], wb = a.bits[rb >> 5];
  const double lpi = a.logpi[ri];
  const double dli = a.delta[ri];
  if (threadIdx.x < PB / 32) s_ins[threadIdx.x] = 0u;
  if (threadIdx.x == 0) s_n = 0u;
  const uint32_t bi = (wi >> (ri & 31u)) & 1u, ba = (wa >> (ra & 31u)) & 1u, bb = (wb >> (rb & 31u)) & 1u;
  double tp[C], ti[C], ta[C], tb[C];
  load_row<L, C>((bi ? a.slot1 : a.slot0) + (size_t)ri * LD, 0, ti);
  load_row<L, C>((ba ? a.slot1 : a.slot0) + (size_t)ra * LD, 0, ta);
  load_row<L, C>((bb ? a.slot1 : a.slot0) + (size_t)rb * LD, 0, tb);
  stage.store(s_model);
  __syncthreads();                                                /* sampler + model tables staged; s_ins, s_n zeroed */
  double g, log_u;
  draws.finish(&s_model.tab, a.gamma0, a.gsig, &g, &log_u);
#pragma unroll
  for (int q = 0; q < C; ++q) tp[q] = ti[q] + (ta[q] - tb[q]) * g;                /* smc:128 */
  double pp[C];
  const double lp = group_logprior<L, C, PLAIN>(s_model.prior, 0, tp, pp, M.mv, M.ext);     /* smc:134 */
  const bool insupport = !(lp == ABZ_NINF);                       /* smc:135 */
  const double kdi = kernel_logpdf_dev(M.abck, a.eps, dli);
  const double wl = lp - lpi;
  const double w_max = (wl + 0.0) - kdi;                          /* smc:140-141 with K(dp) at its maximum */
  const bool may = active && insupport && ((0.0 <= w_max) || (log_u < w_max));
  const unsigned lane = threadIdx.x & 63u;
  const unsigned long long mk = __ballot(may);
  unsigned int base = 0u;
  if (lane == 0u && mk) base = atomicAdd(&s_n, (unsigned)__popcll(mk));
  base = __shfl(base, 0, 64);
  const unsigned int local = base + (unsigned)__popcll(mk & ((1ull << lane) - 1ull));
  if (active && insupport && a.flags) atomicOr(&s_ins[threadIdx.x >> 5], 1u << (threadIdx.x & 31u));
  const unsigned nsim1 = (active && insupport) ? 1u : 0u;         /* nsims counts the in-support proposals (smc:138) */
  __syncthreads();
  if (threadIdx.x == 0) s_base = s_n ? atomicAdd(h.count, s_n) : 0u;          /* one reservation per workgroup */
  __syncthreads();
  if (may) {
    const size_t r = (size_t)s_base + local;
    double2* row = reinterpret_cast<double2*>(h.tp + r * LD);
#pragma unroll
    for (int m = 0; m < C / 2; ++m) { double2 t; t.x = tp[2 * m]; t.y = tp[2 * m + 1]; row[m] = t; }
    h.wl[r] = wl; h.kdi[r] = kdi; h.logu[r] = log_u;
    h.pos[r] = ri | (bi << 31);
  }
  if (threadIdx.x < PB / 32) {                                    /* nothing accepted yet: the second launch flips the bits it accepts */
    const uint32_t w = tile_base / 32u + threadIdx.x;
    if (w * 32u < a.r_lo + a.n_work) a.bits_out[w] = a.bits[w];
  }
  if (a.flags && tile_base + threadIdx.x < a.r_lo + a.n_work)
    a.flags[tile_base + threadIdx.x] = (uint8_t)(((s_ins[threadIdx.x >> 5] >> (threadIdx.x & 31u)) & 1u) << 1);
  block_count2<BLOCK>(0u, nsim1, a.cslots, a.c_cls);
}

template <bool PLAIN, int BLOCK = ABZ_BLOCK>
__device__ inline void smc_lv_phase1_body(const SmcPackedArgs& a, const LvHandList& h) {
  smc_split_phase1_body<ABZ_SIM_LV, 4, PLAIN, BLOCK>(a, h);
}

/* phase 2 for a simulator whose distance is opaque (user-supplied, hiprtc): one lane per record of the list, the whole call, no rounds --
 * what it gains over the one-kernel body is that every wavefront that simulates is full */
template <int SIM, int C, bool PLAIN, int BLOCK = ABZ_BLOCK>
__device__ inline void smc_split_phase2_body(const SmcPackedArgs& a, const LvHandList& h) {
  constexpr int L = 1, LD = C;
  const HotModel& M = a.hm;
  if (blockIdx.x == 0 && threadIdx.x == 0) *h.count_next = 0u;
  if (a.stop && *a.stop) return;
  const unsigned n_list = *h.count;
  const unsigned c0 = blockIdx.x * (unsigned)BLOCK;
  if (c0 >= n_list) return;                                       /* workgroup-uniform */
  __shared__ ModelLds<LD> s_model;
  ModelStage<SIM, LD, BLOCK> stage;
  stage.load(M);
  const bool on = c0 + threadIdx.x < n_list;
  const size_t r = on ? (size_t)c0 + threadIdx.x : (size_t)c0;    /* lanes past the end shadow the chunk's first record */
  double tq[C], pq[C];
  {
    const double2* row = reinterpret_cast<const double2*>(h.tp + r * LD);
#pragma unroll
    for (int m = 0; m < C / 2; ++m) { const double2 t = row[m]; tq[2 * m] = t.x; tq[2 * m + 1] = t.y; }
  }
  const double wl = h.wl[r], kdi = h.kdi[r], logu = h.logu[r];
  const uint32_t pw = h.pos[r];
  stage.store(s_model);
  __syncthreads();
  const double lps = group_logprior<L, C, PLAIN>(s_model.prior, 0, tq, pq, M.mv, M.ext);
  const uint32_t rs = pw & 0x7FFFFFFFu, bs = pw >> 31;
  const double ds = sim_dist<SIM, L, C, false, PLAIN>(M, &s_model.tab, 0, pq, s_model.y, rs, a.sweep, ABZ_RNG_SIM);     /* smc:137 */
  const double w = (wl + kernel_logpdf_dev(M.abck, a.eps, ds)) - kdi;                                              /* smc:140-141 */
  const bool acc = on && ((0.0 <= w) || (logu < w));              /* smc:145 */
  if (acc) {                                                      /* smc:146-150 */
    store_row<L, C>((bs ? a.slot0 : a.slot1) + (size_t)rs * LD, 0, tq);
    atomicXor(&a.bits_out[rs >> 5], 1u << (rs & 31u));            /* phase 1 left bits_out = bits */
    if (a.flags) a.flags[rs] = (uint8_t)3u;
    a.logpi[rs] = lps; a.delta[rs] = ds;
    if (a.stamp) a.stamp[rs] = abz_stamp(rs, a.sweep, 0);
  }
  block_count2<BLOCK>(acc ? 1u : 0u, 0u, a.cslots, a.c_cls);
}

template <bool PLAIN, int BLOCK = ABZ_LV_BLOCK2>
__device__ inline void smc_lv_phase2_body(const SmcPackedArgs& a, const LvHandList& h) {
  constexpr int L = 1, C = 4, LD = 4, PB = BLOCK;
  const HotModel& M = a.hm;
  if (blockIdx.x == 0 && threadIdx.x == 0) *h.count_next = 0u;     /* last read by the sweep before this one; no memset launch */
  if (a.stop && *a.stop) return;
  const unsigned n_list = *h.count;
  const unsigned c0 = blockIdx.x * (unsigned)PB;
  if (c0 >= n_list) return;                                       /* workgroup-uniform: the grid covers the longest possible list */
  const unsigned n = (n_list - c0 < (unsigned)PB) ? n_list - c0 : (unsigned)PB;

  __shared__ ModelLds<LD> s_model;
  __shared__ double s_tp[PB][C];
  __shared__ double s_wl[PB], s_kdi[PB], s_logu[PB];
  __shared__ uint32_t s_pos[PB];
  __shared__ double s_lx[PB], s_ly[PB], s_lacc[PB];
  __shared__ uint16_t s_list[2][PB];
  __shared__ unsigned int s_live[3];

  ModelStage<ABZ_SIM_LV, LD, BLOCK> stage;
  stage.load(M);
  if (threadIdx.x < n) {
    const size_t r = (size_t)c0 + threadIdx.x;
    const double2* row = reinterpret_cast<const double2*>(h.tp + r * LD);
    const double2 t0 = row[0], t1 = row[1];
    s_tp[threadIdx.x][0] = t0.x; s_tp[threadIdx.x][1] = t0.y; s_tp[threadIdx.x][2] = t1.x; s_tp[threadIdx.x][3] = t1.y;
    s_wl[threadIdx.x] = h.wl[r]; s_kdi[threadIdx.x] = h.kdi[r]; s_logu[threadIdx.x] = h.logu[r];
    s_pos[threadIdx.x] = h.pos[r];
    s_list[0][threadIdx.x] = (uint16_t)threadIdx.x;
    s_lx[threadIdx.x] = M.sim_p[0]; s_ly[threadIdx.x] = M.sim_p[1]; s_lacc[threadIdx.x] = 0.0;
  }
  if (threadIdx.x < 3) s_live[threadIdx.x] = 0u;
  stage.store(s_model);
  __syncthreads();

  const LvConst k = lv_const(M);
  /* certain rejection: see smc_swarm_packed_body_2p */
  const double bound = (a.eps > 0.0 && a.eps < 1.0e300) ? (a.eps * a.eps) * (1.0 + 0x1p-40) : ABZ_NAN;
  unsigned n_live = n;
  int cur = 0;
  const unsigned lane = threadIdx.x & 63u, wave0 = threadIdx.x & ~63u;
  int round = 0;
  for (int jo0 = 0; jo0 < k.nobs; jo0 += ABZ_LV_ROUND, ++round) {
    if (wave0 < n_live) {                                         /* wave-uniform: this wavefront still has proposals */
      const bool on = threadIdx.x < n_live;
      const unsigned sl = s_list[cur][on ? threadIdx.x : 0u];     /* idle lanes of a working wave shadow the first proposal */
      double tq[C], pq[C];
#pragma unroll
      for (int q = 0; q < C; ++q) tq[q] = s_tp[sl][q];
      group_push_p<L, C>(s_model.prior, 0, tq, pq);
      double x = s_lx[sl], y = s_ly[sl], dsum = s_lacc[sl];
      const uint32_t rs = s_pos[sl] & 0x7FFFFFFFu;
      bool dead = false;
      for (int jo = jo0; jo < jo0 + ABZ_LV_ROUND && jo < k.nobs; ++jo) {
        lv_observe<false>(M, &s_model.tab, k, rs, a.sweep, ABZ_RNG_SIM, jo, x, y, dsum, nullptr);
        dead = dead || (dsum >= bound);                           /* (false for a NaN sum: it stays, and is rejected at the end) */
        if (jo + 1 < k.nobs) lv_advance(k, pq[0], pq[1], pq[2], pq[3], x, y);
      }
      const bool keep = on && !dead;
      if (keep) { s_lx[sl] = x; s_ly[sl] = y; s_lacc[sl] = dsum; }
      const unsigned long long mk = __ballot(keep);
      unsigned int base = 0u;
      if (lane == 0u && mk) base = atomicAdd(&s_live[round % 3], (unsigned)__popcll(mk));
      base = __shfl(base, 0, 64);
      if (keep) s_list[1 - cur][base + (unsigned)__popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)sl;
    }
    __syncthreads();
    n_live = s_live[round % 3];
    if (threadIdx.x == 0) s_live[(round + 2) % 3] = 0u;           /* the counter of the round after next (last read a round ago) */
    cur = 1 - cur;
  }
  bool acc = false;
  if (wave0 < n_live) {                                           /* the proposals whose distance stayed below the bound to the end */
    const bool on = threadIdx.x < n_live;
    const unsigned sl = s_list[cur][on ? threadIdx.x : 0u];
    double tq[C], pq[C];
#pragma unroll
    for (int q = 0; q < C; ++q) tq[q] = s_tp[sl][q];
    const double lps = group_logprior<L, C, PLAIN>(s_model.prior, 0, tq, pq, M.mv, M.ext);
    const uint32_t pw = s_pos[sl];
    const uint32_t rs = pw & 0x7FFFFFFFu, bs = pw >> 31;
    const double ds = abz_sqrt(s_lacc[sl]);                                                        /* smc:137 */
    const double w = (s_wl[sl] + kernel_logpdf_dev(M.abck, a.eps, ds)) - s_kdi[sl];                /* smc:140-141 */
    acc = on && ((0.0 <= w) || (s_logu[sl] < w));                 /* smc:145 */
    if (acc) {                                                    /* smc:146-150 */
      store_row<L, C>((bs ? a.slot0 : a.slot1) + (size_t)rs * LD, 0, tq);
      atomicXor(&a.bits_out[rs >> 5], 1u << (rs & 31u));          /* phase 1 left bits_out = bits */
      if (a.flags) a.flags[rs] = (uint8_t)3u;                     /* accepted | simulated (phase 1 wrote the second bit) */
      a.logpi[rs] = lps; a.delta[rs] = ds;
      if (a.stamp) a.stamp[rs] = abz_stamp(rs, a.sweep, 0);
    }
  }
  block_count2<BLOCK>(acc ? 1u : 0u, 0u, a.cslots, a.c_cls);
}

/* Two phases wherever a skipped simulation is worth a hand-over through LDS: rows spread over 2, 4 or 8 lanes (the d-dimensional
 * Normal simulator: four Philox blocks and Box-Muller pairs per lane), and the Lotka-Volterra simulator (1500 RK4 steps per call; with
 * its bounded prior half of the proposals and more leave the support, smc:135, and in the one-phase body their lanes idle through
 * their wave-mates' simulations), and user-supplied simulators of 3 to 8 parameters (cost unknown, usually the bulk of the sweep; in two LAUNCHES, the default for them, up to 16 parameters).
 * One phase for the rest: the cheap one-lane simulators, rows of one or two doubles (double-buffered), the widest lane groups.  ABZ_SWEEP_ONE_PHASE forces the one-phase body everywhere (A/B measurements). */
/* threads per workgroup of the sweep: 512 for the Lotka-Volterra simulator -- about a third of a tile's proposals reach phase 2, and
 * 170 of 512 fill three wavefronts to 89 % where 85 of 256 fill two to 67 % (the simulator is all of that kernel's time) */
template <int SIM, int L, int C>
constexpr int abz_sweep_block() {
  return (SIM == ABZ_SIM_LV && L == 1 && C == 4) ? ABZ_LV_BLOCK : ABZ_BLOCK;
}
template <int SIM, int L, int C, bool PLAIN = false>
__device__ inline void smc_swarm_packed_body(const SmcPackedArgs& a) {
  if constexpr ((L >= 2 && L <= 8 && C >= 2 && C <= 8) || ((SIM == ABZ_SIM_LV || SIM == ABZ_SIM_USER) && L == 1 && (C == 4 || C == 8)))
    smc_swarm_packed_body_2p<SIM, L, C, PLAIN, abz_sweep_block<SIM, L, C>()>(a);
  else
    smc_swarm_packed_body_1p<SIM, L, C, PLAIN>(a);
}

/* replay of a packed sweep on a replica (multi-GPU): every rank keeps the whole population, rank r sweeps a range of
 * positions and publishes one flag byte per position; the accepted proposal theta_i + gamma (theta_a - theta_b)
 * (smc:128) is a function of replicated rows and of the position's counter-based random numbers, so the other ranks
 * REBUILD it -- and its log-prior -- from their replica instead of receiving the row.  The own range [skip_lo, skip_hi)
 * starts and ends at multiples of 64 (or at n_alive), so a wave's 64 positions are all own (counted only) or all foreign. */
struct SmcReplayPackedArgs {
  HotModel hm;
  const uint32_t* bits;
  uint32_t* bits_out;
  const uint8_t* flags;         /* by position: bit 0 accepted, bit 1 simulated */
  double* slot0;
  double* slot1;
  double* logpi;
  uint64_t* stamp;              /* blob stamps (in place), NULL when blobs are off */
  unsigned long long* cslots;   /* (nacc, nsim) over ALL positions of the prefix -> ABZ_C_RACC, ABZ_C_RSIM */
  const unsigned long long* stop;   /* group of sweeps: non-zero = the early exit of smc:352 held before this sweep; NULL = always run */
  double gamma0, gsig;
  uint32_t n_alive, skip_lo, skip_hi, sweep;
};

template <int L, int C, bool PLAIN = false>
__device__ inline void smc_replay_packed_body(const SmcReplayPackedArgs& a) {
  constexpr int LD = L * C;
  if (a.stop && *a.stop) return;                          /* grid-uniform: the sweep this replay belongs to did not run */
  __shared__ ModelLds<LD> s_model;
  __shared__ uint32_t s_list[ABZ_REPLAY_CHUNK];           /* accepted foreign positions of this block */
  __shared__ unsigned int s_n;
  __shared__ unsigned int s_cnt[2][ABZ_BLOCK / 64];

  ModelStage<-1, LD> stage;                               /* sampler tables + prior descriptors (no simulator data) */
  stage.load(a.hm);
  if (threadIdx.x == 0) s_n = 0u;
  __syncthreads();

  const uint32_t base = blockIdx.x * (uint32_t)ABZ_REPLAY_CHUNK;
  const unsigned lane = threadIdx.x & 63u;
  unsigned int wacc = 0u, wsim = 0u;                      /* wave-uniform counters */
  unsigned fv[ABZ_REPLAY_PER];
#pragma unroll
  for (int k = 0; k < ABZ_REPLAY_PER; ++k) {
    const uint32_t r = base + (uint32_t)k * ABZ_BLOCK + threadIdx.x;
    fv[k] = r < a.n_alive ? (unsigned)a.flags[r] : 0u;
  }
#pragma unroll
  for (int k = 0; k < ABZ_REPLAY_PER; ++k) {
    const uint32_t r = base + (uint32_t)k * ABZ_BLOCK + threadIdx.x;
    const unsigned f = fv[k];
    wacc += (unsigned)__popcll(__ballot((f & 1u) != 0u));
    wsim += (unsigned)__popcll(__ballot((f & 2u) != 0u));
    /* the wave's 64 positions are all own or all foreign: the own range starts and ends at multiples of 64 */
    const uint32_t r0 = r - lane;
    const bool foreign_blk = !(r0 >= a.skip_lo && r0 < a.skip_hi);
    const bool acc = foreign_blk && (f & 1u) != 0u;
    const unsigned long long m = __ballot(acc);
    unsigned long long flip = m;
    if constexpr (ABZ_ROWS_DOUBLE_BUFFERED(LD)) {         /* every swept position moves to its other slot (smc_swarm_packed_body) */
      const bool in = foreign_blk && r < a.n_alive;
      flip = __ballot(in);
      if (in && !acc) {
        const uint32_t b = packed_bit(a.bits, r);
        const double* src = (b ? a.slot1 : a.slot0) + (size_t)r * LD;
        double* dst = (b ? a.slot0 : a.slot1) + (size_t)r * LD;
#pragma unroll
        for (int q = 0; q < LD; ++q) dst[q] = src[q];
      }
    }
    if (foreign_blk && lane == 0u) {                      /* the wave's 64 positions = two words of the bitmap */
      const uint32_t w = (r - lane) >> 5;
      if (w * 32u < a.n_alive) a.bits_out[w] = a.bits[w] ^ (uint32_t)flip;
      if ((w + 1u) * 32u < a.n_alive) a.bits_out[w + 1u] = a.bits[w + 1u] ^ (uint32_t)(flip >> 32);
    }
    const unsigned cnt = (unsigned)__popcll(m);
    unsigned int at = 0u;
    if (lane == 0u && cnt) at = atomicAdd(&s_n, cnt);
    at = __shfl(at, 0, 64);
    if (acc) s_list[at + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = r;
  }
  if (lane == 0u) { s_cnt[0][threadIdx.x >> 6] = wacc; s_cnt[1][threadIdx.x >> 6] = wsim; }
  stage.store(s_model);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long x = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
    const unsigned long long y = s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3];
    unsigned long long* s = a.cslots + (size_t)(blockIdx.x & (ABZ_CSLOTS - 1)) * ABZ_CSTRIDE;
    if (x) (void)__hip_atomic_fetch_add(s + ABZ_C_RACC, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (y) (void)__hip_atomic_fetch_add(s + ABZ_C_RSIM, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  const unsigned n = s_n;
  const int j = (int)(threadIdx.x % L);
  for (unsigned t = threadIdx.x / L; t < ((n + ABZ_BLOCK / L - 1) / (ABZ_BLOCK / L)) * (ABZ_BLOCK / L); t += ABZ_BLOCK / L) {
    const bool on = t < n;                                /* whole groups idle together; shuffles stay converged */
    const uint32_t ri = s_list[on ? t : 0u];
    uint32_t ra, rb;
    double g, log_u;
    particle_draws<L>(&s_model.tab, a.hm.seed, ri, a.sweep, j, a.n_alive, ri, a.gamma0, a.gsig, &ra, &rb, &g, &log_u);
    const uint32_t bi = packed_bit(a.bits, ri), ba = packed_bit(a.bits, ra), bb = packed_bit(a.bits, rb);
    double ti[C], ta[C], tb[C], tp[C], pp[C];
    load_row<L, C>((bi ? a.slot1 : a.slot0) + (size_t)ri * LD, j, ti);
    load_row<L, C>((ba ? a.slot1 : a.slot0) + (size_t)ra * LD, j, ta);
    load_row<L, C>((bb ? a.slot1 : a.slot0) + (size_t)rb * LD, j, tb);
#pragma unroll
    for (int q = 0; q < C; ++q) tp[q] = ti[q] + (ta[q] - tb[q]) * g;                       /* smc:128 */
    const double lp = group_logprior<L, C, PLAIN>(s_model.prior, j, tp, pp, a.hm.mv, a.hm.ext);      /* what the owner stored, smc:147 */
    if (on) {
      store_row<L, C>((bi ? a.slot0 : a.slot1) + (size_t)ri * LD, j, tp);
      if (j == 0) {
        a.logpi[ri] = lp;
        if (a.stamp) a.stamp[ri] = abz_stamp(ri, a.sweep, 0);
      }
    }
  }
}

/* ================================================================ S4: abcdemc_swarm! (src/abcdez_mc.jl:5-61) */
struct McSwarmArgs {
  HotModel hm;
  const uint32_t* order;
  const uint32_t* cnt;          /* by particle: #{j : Ds[j] <= Ds[i]} (valid where Ds[i] > eps_pop) */
  const double* theta;
  const double* logpi;
  const double* delta;
  double* ntheta;
  double* nlogpi;
  double* ndelta;
  unsigned long long* cslots;   /* cumulative counter slots: #(new Ds > eps_target) -> ABZ_C_MCGT, nsim -> ABZ_C_MCSIM */
  unsigned long long* mm_cur;   /* [ABZ_MMSLOTS][2] (min key, max key) of the new distances: this sweep's bank ... */
  unsigned long long* mm_nxt;   /* ... and the bank it resets for its successor */
  double eps_pop, eps_target, gamma0, gsig;
  const unsigned long long* eps_pop_dev;   /* non-NULL: eps_pop (f64 bits) was made on the device (mc_window_kernel) */
  uint32_t N, i0, n_local, sweep;
  const uint64_t* stamp;        /* blob stamps, both NULL when blobs are off */
  uint64_t* nstamp;
  const unsigned long long* seq_dev;   /* non-NULL: the RNG epoch is sweep + *seq_dev (generations replayed as a graph, abz_ctx.h) */
  /* the better particle of mc:23 by rank (order / cnt) or by rejection (include/abcdez_spec.h, abz_mc_draws_by_rejection):
   * non-NULL = #(Ds > eps_target) of the distances this sweep reads, kept on the device -- the sweep applies the rule itself
   * (abcdez_mc_generation_async); NULL = the caller applied it: by rejection iff order == NULL */
  const unsigned long long* nabove_dev;
  unsigned long long* reject_fail;     /* set when a particle found no better particle in 1024 trials (include/abcdez_spec.h) */
  /* a rank pass for THIS generation's distances has been enqueued before the sweep (order / cnt are its outputs).  The host decides
   * whether to launch one, the device decides rank-or-rejection from nabove_dev: should the two ever disagree, order / cnt are not
   * this generation's (or were never written) and must not be used as indices */
  uint32_t rank_built;
};

__device__ inline uint32_t upper_bound_f64(const double* __restrict__ v, uint32_t n, double x) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (v[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

template <int SIM, int L, int C, bool PLAIN = false>
__device__ inline void mc_swarm_kernel_body(const McSwarmArgs& a) {
  constexpr int LD = L * C;
  constexpr uint32_t PB = ABZ_BLOCK / L;
  const HotModel& M = a.hm;
  const uint64_t seed = M.seed;
  const uint32_t sweep = a.sweep + (a.seq_dev ? (uint32_t)*a.seq_dev : 0u);      /* wave-uniform: a scalar load */
  __shared__ ModelLds<LD> s_model;
  {
    ModelStage<SIM, LD> stage;
    stage.load(M);
    stage.store(s_model);
  }
  __syncthreads();                                                          /* once per workgroup */
  const int j = (int)(threadIdx.x % L);
  const double eps_pop = a.eps_pop_dev ? abz_u2d(*a.eps_pop_dev) : a.eps_pop;
  /* wave-uniform (scalar loads): how this generation draws its better particles */
  const bool reject = a.nabove_dev ? abz_mc_draws_by_rejection(*a.nabove_dev, a.N) != 0 : a.order == nullptr;
  const uint32_t ntiles = (a.n_local + PB - 1) / PB;
  /* driver reductions of the generation this sweep leaves behind (mc:146,156,163), carried over the workgroup's tiles:
   * #(Ds > eps_target), nsims, extrema(Ds) as order keys */
  unsigned int n_gt = 0u, n_sim = 0u;
  unsigned long long lo = ~0ull, hi = 0ull;
  ABZ_TILE_LOOP(tile, ntiles) {
    const uint32_t grp = tile * PB + threadIdx.x / L;
    const bool active = grp < a.n_local;
    const uint32_t i = a.i0 + (active ? grp : 0u);
    /* Order of issue = order of need: three dependent round trips (state + candidate count | order[] | rows); the Philox
     * words need no memory */
    const double lpi = a.logpi[i];
    const double di = a.delta[i];
    const uint32_t cnt_i = reject ? 0u : a.cnt[i];                          /* meaningful only where di > eps (abz_sort.hip) */
    const abz_u64x2 w_donor = abz_rng(seed, i, sweep, 0, ABZ_RNG_DONOR);
    const double eps = di <= a.eps_target ? a.eps_target : eps_pop;         /* mc:19 */
    uint32_t s = i;
    if (di > eps) {                                                         /* mc:20-24 */
      if (reject) {
        int exhausted;
        s = abz_mc_better_by_rejection(seed, i, sweep, a.delta, a.N, di, &exhausted);
        if (exhausted) *a.reject_fail = 1ull;     /* the rule's premise did not hold for the distances read: the host hears of it */
      }
      else if (a.rank_built && cnt_i >= 1u && cnt_i <= a.N) {
        const uint32_t t = a.order[abz_randint(abz_rng(seed, i, sweep, 0, ABZ_RNG_BETTER).w0, cnt_i)];
        if (t < a.N) s = t; else *a.reject_fail = 2ull;
      } else *a.reject_fail = 2ull;   /* no enumeration of this generation to draw from: keep s = i, the host hears of it (abcdez_mc_generation_wait) */
    }
    uint32_t ia, ib;                                                        /* mc:25-32 */
    abz_donor_ranks(w_donor, a.N, s, &ia, &ib);

    double ti[C], ts[C], ta[C], tb[C];
    load_row<L, C>(a.theta + (size_t)i * LD, j, ti);
    load_row<L, C>(a.theta + (size_t)s * LD, j, ts);
    load_row<L, C>(a.theta + (size_t)ia * LD, j, ta);
    load_row<L, C>(a.theta + (size_t)ib * LD, j, tb);

    double z0, z1;
    abz_normal_pair(abz_rng(seed, i, sweep, 0, ABZ_RNG_JITTER), &s_model.tab, &z0, &z1);
    const double g = a.gamma0 * (1.0 + z0 * a.gsig);                        /* mc:34 */
    double tp[C], pp[C];
#pragma unroll
    for (int q = 0; q < C; ++q) tp[q] = ts[q] + (ta[q] - tb[q]) * g;

    const double lp = group_logprior<L, C, PLAIN>(s_model.prior, j, tp, pp, M.mv, M.ext);        /* mc:41 */
    const double w_prior = lp - lpi;                                        /* mc:42 */
    const double u = abz_u01_open(abz_rng(seed, i, sweep, 0, ABZ_RNG_ACCEPT).w0);
    double mn = w_prior < 0.0 ? w_prior : 0.0;
    if (abz_isnan(w_prior)) mn = w_prior;
    const bool simulate = !(abz_log_tab(u, &s_model.tab) > mn);                               /* mc:43 */
    bool acc = false;
    double dp = di;
    if (simulate) {
      dp = sim_dist<SIM, L, C>(M, &s_model.tab, j, pp, s_model.y, i, sweep, ABZ_RNG_SIM);       /* mc:45 */
      const double thr = eps > di ? eps : di;
      acc = dp <= thr;                                                      /* mc:54 */
    }
    if (active) {
      double to[C];
#pragma unroll
      for (int q = 0; q < C; ++q) to[q] = acc ? tp[q] : ti[q];
      store_row<L, C>(a.ntheta + (size_t)i * LD, j, to);
      if (j == 0) {
        a.nlogpi[i] = acc ? lp : lpi;
        a.ndelta[i] = acc ? dp : di;
        if (a.nstamp) a.nstamp[i] = acc ? abz_stamp(i, sweep, 0) : a.stamp[i];
      }
    }
    const double dn = acc ? dp : di;
    if (active && j == 0) {
      n_gt += dn > a.eps_target ? 1u : 0u;
      n_sim += simulate ? 1u : 0u;
      const unsigned long long key = f64_order_key(dn);
      lo = key < lo ? key : lo;
      hi = key > hi ? key : hi;
    }
  }
  block_count2(n_gt, n_sim, a.cslots, ABZ_C_MCGT);   /* (MCGT, MCSIM) */
  {
    __shared__ unsigned long long s_mm[2][ABZ_BLOCK / 64];
    for (int off = 32; off; off >>= 1) {
      const unsigned long long x = __shfl_xor(lo, off, 64), y = __shfl_xor(hi, off, 64);
      lo = x < lo ? x : lo;
      hi = y > hi ? y : hi;
    }
    if ((threadIdx.x & 63) == 0) { s_mm[0][threadIdx.x >> 6] = lo; s_mm[1][threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < ABZ_BLOCK / 64; ++w) { lo = s_mm[0][w] < lo ? s_mm[0][w] : lo; hi = s_mm[1][w] > hi ? s_mm[1][w] : hi; }
      unsigned long long* m = a.mm_cur + (size_t)(blockIdx.x & (ABZ_MMSLOTS - 1)) * 2;
      if (lo != ~0ull) {
        (void)__hip_atomic_fetch_min(m, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_fetch_max(m + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (blockIdx.x == 0 && threadIdx.x < 2 * ABZ_MMSLOTS) a.mm_nxt[threadIdx.x] = (threadIdx.x & 1) ? 0ull : ~0ull;
  }
}

/* ================================================================ blobs: the second return value of dist! (smc:137,148)
 * Rebuilt when the result is read: particle s re-runs the ONE simulator call its stamp names (origin particle,
 * epoch, init-or-sweep stream) on its current push_p-cast parameters and writes the simulated data to blob[s][:].
 * delta_out[s] is the distance of that re-run -- it must equal the stored distance bit for bit (checked by the
 * host).  theta: dense current rows [n][LD]; blob rows are nbw doubles wide (LD for the MVN simulator, whose blob
 * is laid out like a theta row; n_blob otherwise).                                                           */
template <int SIM>
struct BlobLocal {
  static constexpr int N = (SIM == ABZ_SIM_WIENER || SIM == ABZ_SIM_LV || SIM == ABZ_SIM_USER) ? ABZ_MAX_BLOB : 2;
};

template <int SIM, int L, int C>
__device__ inline void blob_eval_kernel_body(const HotModel& M, const double* __restrict__ theta,
                                             const uint64_t* __restrict__ stamp, uint32_t n,
                                             double* __restrict__ blob, double* __restrict__ delta_out, uint32_t nbw) {
  constexpr int LD = L * C;
  constexpr uint32_t PB = ABZ_BLOCK / L;
  __shared__ ModelLds<LD> s_model;
  {
    ModelStage<SIM, LD> stage;
    stage.load(M);
    stage.store(s_model);
  }
  __syncthreads();
  const int j = (int)(threadIdx.x % L);
  const uint32_t ntiles = (n + PB - 1) / PB;
  ABZ_TILE_LOOP(tile, ntiles) {
    const uint32_t s = tile * PB + threadIdx.x / L;
    if (s >= n) continue;                       /* whole groups leave together; no barrier inside the loop */
    double th[C], pp[C];
    load_row<L, C>(theta + (size_t)s * LD, j, th);
    (void)group_logprior<L, C>(s_model.prior, j, th, pp, M.mv, M.ext);  /* push_p (types.jl:20-23) */
    const uint64_t st = stamp[s];
    const uint32_t purpose = abz_stamp_is_init(st) ? (uint32_t)ABZ_RNG_INIT_SIM : (uint32_t)ABZ_RNG_SIM;
    if constexpr (SIM == ABZ_SIM_MVN) {
      double b[C];
      const double dl = sim_dist<SIM, L, C, true>(M, &s_model.tab, j, pp, s_model.y, abz_stamp_origin(st),
                                                  abz_stamp_epoch(st), purpose, b);
      store_row<L, C>(blob + (size_t)s * nbw, j, b);             /* nbw == LD */
      if (j == 0) delta_out[s] = dl;
    } else {
      static_assert(SIM == ABZ_SIM_MVN || L == 1, "only the MVN simulator spreads a row over lanes");
      double b[BlobLocal<SIM>::N];
      const double dl = sim_dist<SIM, L, C, true>(M, &s_model.tab, j, pp, s_model.y, abz_stamp_origin(st),
                                                  abz_stamp_epoch(st), purpose, b);
      const int nb = M.n_blob < BlobLocal<SIM>::N ? M.n_blob : BlobLocal<SIM>::N;
      for (int q = 0; q < nb; ++q) blob[(size_t)s * nbw + q] = b[q];
      delta_out[s] = dl;
    }
  }
}

#endif /* ABZ_KERNELS_H */
