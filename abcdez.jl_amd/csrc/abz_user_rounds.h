/* abz_user_rounds.h -- the STAGED form of a user-supplied simulator, included behind the user's source in the run-time translation
 * unit (abz_jit.hip).  The reference calls dist!(theta, ve) as one opaque function (src/abcdez_smc.jl:137); a simulator whose distance is
 * a running quantity that only grows -- a sum of squared errors over the observations of a trajectory, say -- can tell the sweep more:
 *
 *     #define ABZ_USER_ROUNDS 8          // the simulation in this many steps
 *     #define ABZ_USER_STATE 3           // doubles of state carried from one step to the next (<= 8; zero before round 0)
 *     __device__ double abz_user_round(const double* theta, int d, const double* data, int n_data, const double* sim_p,
 *                                      abz_user_rng& rng, int round, double* state);
 *
 * abz_user_round advances the simulation by one step and returns a LOWER BOUND of the final distance that never decreases from one
 * round to the next; what the last round returns IS the distance.  rng continues where the round before left it.
 *
 * What the library does with it (abz_kernels.h, the two-launch sweep of rows of 4, 8 or 16 doubles): the second launch runs the proposals
 * round by round, as the built-in Lotka-Volterra simulator does; a proposal whose bound has passed eps is rejected for certain -- every
 * ABC kernel is zero beyond eps (src/abcdez_types.jl:26-73), so smc:140-145 reject it whatever the remaining rounds would return --,
 * leaves, and the proposals still in flight are re-packed into the leading lanes.  The accepted population is bit for bit what the
 * whole simulation gives: a survivor has had exactly abz_user_round's operations in order.  Everywhere else (initial population,
 * abcdemc, other row widths, blobs) the rounds run back to back through the abz_user_dist defined here. */
#ifndef ABZ_USER_ROUNDS_H
#define ABZ_USER_ROUNDS_H

#if defined(ABZ_USER_ROUNDS)
#ifndef ABZ_USER_STATE
#define ABZ_USER_STATE 1
#endif
static_assert(ABZ_USER_ROUNDS >= 1 && ABZ_USER_ROUNDS <= 4096, "ABZ_USER_ROUNDS: 1 .. 4096 rounds");
static_assert(ABZ_USER_STATE >= 1 && ABZ_USER_STATE <= 8, "ABZ_USER_STATE: 1 .. 8 doubles of carried state");

__device__ double abz_user_round(const double* theta, int d, const double* data, int n_data, const double* sim_p, abz_user_rng& rng,
                                 int round, double* state);

/* the whole simulation: every round, in order */
__device__ double abz_user_dist(const double* theta, int d, const double* data, int n_data, const double* sim_p, abz_user_rng& rng) {
  double st[ABZ_USER_STATE];
#pragma unroll
  for (int q = 0; q < ABZ_USER_STATE; ++q) st[q] = 0.0;
  double v = 0.0;
  for (int r = 0; r < ABZ_USER_ROUNDS; ++r) v = abz_user_round(theta, d, data, n_data, sim_p, rng, r, st);
  return v;
}

/* second launch of the two-launch sweep (abz_kernels.h: smc_split_phase1_body is the first): one lane per record of the hand-over
 * list, the simulation round by round, leavers dropped and survivors re-packed after every round */
template <int C, bool PLAIN, int BLOCK = ABZ_BLOCK>
__device__ inline void smc_user_rounds_phase2_body(const SmcPackedArgs& a, const LvHandList& h) {
  constexpr int L = 1, LD = C, PB = BLOCK, NS = ABZ_USER_STATE;
  const HotModel& M = a.hm;
  if (blockIdx.x == 0 && threadIdx.x == 0) *h.count_next = 0u;     /* last read by the sweep before this one; no memset launch */
  if (a.stop && *a.stop) return;
  const unsigned n_list = *h.count;
  const unsigned c0 = blockIdx.x * (unsigned)PB;
  if (c0 >= n_list) return;                                       /* workgroup-uniform: the grid covers the longest possible list */
  const unsigned n = (n_list - c0 < (unsigned)PB) ? n_list - c0 : (unsigned)PB;

  __shared__ ModelLds<LD> s_model;
  __shared__ double s_tp[C][PB];                                  /* component-major: a wavefront's lanes read consecutive words */
  __shared__ double s_wl[PB], s_kdi[PB], s_logu[PB], s_val[PB];
  __shared__ double s_state[NS][PB];
  __shared__ uint32_t s_pos[PB], s_sub[PB];
  __shared__ uint16_t s_list[2][PB];
  __shared__ unsigned int s_live[3];

  ModelStage<ABZ_SIM_USER, LD, BLOCK> stage;
  stage.load(M);
  if (threadIdx.x < n) {
    const size_t r = (size_t)c0 + threadIdx.x;
    const double2* row = reinterpret_cast<const double2*>(h.tp + r * LD);
#pragma unroll
    for (int m = 0; m < C / 2; ++m) { const double2 t = row[m]; s_tp[2 * m][threadIdx.x] = t.x; s_tp[2 * m + 1][threadIdx.x] = t.y; }
    s_wl[threadIdx.x] = h.wl[r]; s_kdi[threadIdx.x] = h.kdi[r]; s_logu[threadIdx.x] = h.logu[r];
    s_pos[threadIdx.x] = h.pos[r];
    s_list[0][threadIdx.x] = (uint16_t)threadIdx.x;
#pragma unroll
    for (int q = 0; q < NS; ++q) s_state[q][threadIdx.x] = 0.0;
    s_sub[threadIdx.x] = 0u; s_val[threadIdx.x] = 0.0;
  }
  if (threadIdx.x < 3) s_live[threadIdx.x] = 0u;
  stage.store(s_model);
  __syncthreads();

  unsigned n_live = n;
  int cur = 0;
  const unsigned lane = threadIdx.x & 63u, wave0 = threadIdx.x & ~63u;
  for (int round = 0; round < ABZ_USER_ROUNDS; ++round) {
    if (wave0 < n_live) {                                         /* wave-uniform: this wavefront still has proposals */
      const bool on = threadIdx.x < n_live;
      const unsigned sl = s_list[cur][on ? threadIdx.x : 0u];     /* idle lanes of a working wave shadow the first proposal */
      double tq[C], pq[C], st[NS];
#pragma unroll
      for (int q = 0; q < C; ++q) tq[q] = s_tp[q][sl];
      group_push_p<L, C>(s_model.prior, 0, tq, pq);
#pragma unroll
      for (int q = 0; q < NS; ++q) st[q] = s_state[q][sl];
      const uint32_t rs = s_pos[sl] & 0x7FFFFFFFu;
      abz_user_rng rng{M.seed, rs, a.sweep, (uint32_t)ABZ_RNG_SIM, s_sub[sl], &s_model.tab};
      const double lb = abz_user_round(pq, M.d, M.data, M.n_data, M.sim_p, rng, round, st);
      /* certain rejection: the final distance is >= lb > eps, outside the support of every ABC kernel (types.jl:26-73); a NaN bound
       * compares false, stays to the end and is rejected there like any NaN distance */
      const bool keep = on && !(lb > a.eps);
      if (keep) {
#pragma unroll
        for (int q = 0; q < NS; ++q) s_state[q][sl] = st[q];
        s_sub[sl] = rng.sub; s_val[sl] = lb;
      }
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
  if (wave0 < n_live) {                                           /* the proposals whose bound stayed at or below eps to the end */
    const bool on = threadIdx.x < n_live;
    const unsigned sl = s_list[cur][on ? threadIdx.x : 0u];
    double tq[C], pq[C];
#pragma unroll
    for (int q = 0; q < C; ++q) tq[q] = s_tp[q][sl];
    const double lps = group_logprior<L, C, PLAIN>(s_model.prior, 0, tq, pq, M.mv, M.ext);
    const uint32_t pw = s_pos[sl];
    const uint32_t rs = pw & 0x7FFFFFFFu, bs = pw >> 31;
    const double ds = s_val[sl];                                                                    /* smc:137 */
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
#endif /* ABZ_USER_ROUNDS */

#endif /* ABZ_USER_ROUNDS_H */
