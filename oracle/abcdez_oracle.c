/*
 * abcdez_oracle.c -- CPU restatement of ABCdeZ.jl's per-generation population loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package (abcdez.jl_amd/) may
 * import, link or call this file; it is the checker used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg.
 *
 * PARITY STATUS.  PINNED against every known answer the reference's own tests hold for this path
 * (SURVEY.md section 8c), restated as committed fixtures (tests/golden/reference_known_answers.json):
 *   exact:        kernel truth tables test/runtests.jl:48-108, Factored :21-36, push_p :38-46
 *                 -> tests/test_reference_known_answers.py
 *   statistical:  all 17 integration testsets :110-624 with the reference's population sizes and
 *                 tolerances (analytic evidences, Bayes factor, posterior means, Dirac, mixture deciles,
 *                 Wiener, 2-d with Inf distances, mixed discrete prior, Socks)
 *                 -> tests/test_reference_integration.py
 *   structural:   resampling / driver invariants implied by src/abcdez_smc.jl:45-54 and :295-377
 *                 -> tests/test_oracle_equivalence.py, tests/test_reference_integration.py
 * NOT pinnable: seed-for-seed equality with ABCdeZ.jl.  The reference is pure Julia, Julia is not
 * installed in the build container (nor on the GPU box), its RNG (task-local Xoshiro256++, forked per
 * FLoops task) is not observable from here, and test/runtests.jl asserts no seed-specific value.  "Same
 * seed, same result" is therefore defined between this oracle and the HIP kernels (shared Philox stream).
 *
 * Two tiers:
 *   ref_*  literal restatements (sequential fp walks, O(N) donor scans, rejection
 *          loops, left-to-right sums) following the reference line by line, with
 *          the Philox stream standing in for Julia's rng;
 *   orc_*  the "spec" restatement the HIP kernels must match bit for bit: same
 *          algorithm, but with the order-free formulations a GPU needs (rank-skip
 *          donors, integer cumulative weights, fixed summation trees, alive particles
 *          kept as a prefix by orc_smc_partition).  Tests show ref_* and orc_* agree
 *          (exactly where the formulation is exact, in law where only the consumption
 *          of random numbers or the labelling of the particles differs).  The dense
 *          functions (orc_smc_swarm on plain arrays, used by the C driver) and the
 *          *_packed functions (the device's two-slot storage) are two restatements of
 *          the same spec and are tested against each other.
 *
 * All citations are into /root/reference (ABCdeZ.jl v0.6.0).
 * Layout: theta is row-major double[N][ld]; logpi, delta, wns are double[N];
 * alive is uint8[N]; indices are uint32.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/abcdez_spec.h"

#define ORC_API __attribute__((visibility("default")))
#define ORC_MAX_RETRY 100000u
#define ORC_T (&abz_tables_host)

/* ---------------------------------------------------------------- thin exports of the spec math
 * (so the tests can pin them against mpmath / the published Philox vectors)        */
ORC_API void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  abz_u32x4 r = abz_philox4x32(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);       /* ABZ_PHILOX_ROUNDS rounds: the product's stream */
  memcpy(out, r.v, 16);
}
ORC_API void orc_philox_r(int rounds, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  abz_u32x4 r = abz_philox4x32_r(rounds, ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);   /* the same round function, any count */
  memcpy(out, r.v, 16);
}
ORC_API int orc_philox_rounds(void) { return ABZ_PHILOX_ROUNDS; }
ORC_API void orc_math_eval(int fn, const double* x, double* y, double* y2, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    switch (fn) {
      case 0: y[i] = abz_log(x[i]); break;
      case 1: y[i] = abz_exp(x[i]); break;
      case 2: abz_sincos2pi(x[i], &y[i], &y2[i]); break;
      case 3: y[i] = abz_rint(x[i]); break;
      case 4: y[i] = abz_floor(x[i]); break;
      case 5: y[i] = abz_sqrt(x[i]); break;
      case 7: y[i] = abz_log_tab(x[i], ORC_T); break;
      case 8: abz_sincos2pi_tab(x[i], ORC_T, &y[i], &y2[i]); break;
      case 9: y[i] = abz_sqrt_pn(x[i]); break;
      case 10: y[i] = abz_lgamma(x[i]); break;
      default: y[i] = x[i] / (y2 ? y2[i] : 1.0); break;
    }
  }
}
ORC_API void orc_rng_words(uint64_t seed, uint32_t idx, uint32_t epoch, uint32_t sub, uint32_t purpose,
                           uint64_t out[2]) {
  abz_u64x2 w = abz_rng(seed, idx, epoch, sub, purpose);
  out[0] = w.w0; out[1] = w.w1;
}
ORC_API void orc_normal_pairs(uint64_t seed, uint32_t purpose, int64_t n, double* z) {
  for (int64_t i = 0; i < n; ++i) {
    abz_normal_pair(abz_rng(seed, (uint32_t)i, 0, 0, purpose), ORC_T, &z[2 * i], &z[2 * i + 1]);
  }
}
ORC_API void orc_donor_ranks(uint64_t w0, uint64_t w1, uint32_t n_alive, uint32_t ri, uint32_t* ra, uint32_t* rb) {
  abz_u64x2 w; w.w0 = w0; w.w1 = w1;
  abz_donor_ranks(w, n_alive, ri, ra, rb);
}
/* threads of a loop over n particles: a team of 256 costs ~1 ms per parallel region whatever it does -- the small populations of
 * the parity tests (thousands of generations of a few thousand particles) spent minutes there on the GPU box's 256 cores */
static int orc_threads(int64_t n);
/* ... of a loop that runs the simulator once per element: weighted by what one call costs (an ODE integration is thousands of
 * times a 1-D normal draw) */
static int orc_threads_sim(const abz_model* M, int64_t n) {
  const int64_t w = M->sim_id == ABZ_SIM_LV || M->sim_id == ABZ_SIM_WIENER ? 4096 : M->ld > 4 ? M->ld / 4 : 1;
  return orc_threads(n > (INT64_MAX >> 13) ? n : n * w);
}
static int orc_threads(int64_t n) {
#ifdef _OPENMP
  const int t = omp_get_max_threads();
  const int64_t want = n / 512 + 1;
  return want < (int64_t)t ? (int)want : t;
#else
  (void)n;
  return 1;
#endif
}
/* the per-particle scalar draws of one sweep, straightforward evaluation (checker of abcdez_draws_eval):
 * donors smc:119-126, gamma smc:128, log(rand) smc:145; alive rank of particle i = i in a pool of n_pool */
ORC_API void orc_particle_draws(uint64_t seed, int64_t i0, int64_t n, int64_t n_pool, uint32_t sweep, double gamma0,
                                double gsig, uint32_t* ra, uint32_t* rb, double* g, double* log_u) {
#pragma omp parallel for schedule(static) num_threads(orc_threads(n))
  for (int64_t k = 0; k < n; ++k) {
    const uint32_t i = (uint32_t)(i0 + k);
    abz_donor_ranks(abz_rng(seed, i, sweep, 0, ABZ_RNG_DONOR), (uint32_t)n_pool, i, &ra[k], &rb[k]);
    double z0, z1;
    abz_normal_pair(abz_rng(seed, i, sweep, 0, ABZ_RNG_JITTER), ORC_T, &z0, &z1);
    g[k] = gamma0 * (1.0 + z0 * gsig);
    log_u[k] = abz_log_tab(abz_u01_open(abz_rng(seed, i, sweep, 0, ABZ_RNG_ACCEPT).w0), ORC_T);
  }
}
/* abcdemc's better particle by rejection for particle i over `n` consecutive sweeps (test hook of the law: uniform over
 * {j : delta[j] <= delta[i]}); trials[k] = 0 when the trials ran out */
ORC_API void orc_mc_better_by_rejection(uint64_t seed, const double* delta, int64_t N, uint32_t i, uint32_t sweep0, int64_t n,
                                        uint32_t* s_out, uint8_t* found) {
  for (int64_t k = 0; k < n; ++k) {
    int exhausted;
    s_out[k] = abz_mc_better_by_rejection(seed, i, sweep0 + (uint32_t)k, delta, (uint32_t)N, delta[i], &exhausted);
    found[k] = (uint8_t)!exhausted;
  }
}
ORC_API int orc_mc_draws_by_rejection(int64_t n_above, int64_t N) { return abz_mc_draws_by_rejection((uint64_t)n_above, (uint64_t)N); }
ORC_API uint64_t orc_weight_fix(double w, uint32_t n) { return abz_weight_fix(w, n); }
ORC_API double orc_u01(uint64_t w, int kind) { return kind == 1 ? abz_u01_open(w) : (kind == 2 ? abz_u01_52(w) : abz_u01_co(w)); }
ORC_API uint32_t orc_randint(uint64_t w, uint32_t n) { return abz_randint(w, n); }
ORC_API double orc_prior_logpdf1(const abz_prior_dim* pd, double x) { return abz_prior_logpdf1(pd, x); }
/* the same for a factor of a MODEL (wrapper families read their records from the model's ext table) */
ORC_API double orc_model_prior_logpdf(const abz_model* M, int k, double x) { return abz_prior_logpdf1x(&M->prior[k], x, M->ext); }
/* one draw of factor k as abcde_init! makes it for particle i at retry `retry` (families beyond Normal / (Discrete)Uniform) */
ORC_API double orc_model_prior_draw_ext(const abz_model* M, int k, uint32_t i, uint32_t retry) { return abz_prior_draw_extx(&M->prior[k], M->seed, i, retry, (uint32_t)k, ORC_T, M->ext); }
ORC_API double orc_kernel_pdf(int kind, double eps, double x) { return abz_kernel_pdf(kind, eps, x); }
ORC_API double orc_kernel_logpdf(int kind, double eps, double x) { return abz_kernel_logpdf(kind, eps, x); }

/* ---------------------------------------------------------------- priors (priors.jl:40-46, types.jl:20-23) */
static void push_row(const abz_model* M, const double* th, double* out) {
  for (int k = 0; k < M->ld; ++k) out[k] = abz_push_p(&M->prior[k], th[k]);
}
/* a correlated Normal prior (abcdez_spec.h, abz_model.mv): the components the per-dimension log-densities are taken of */
static const double* whitened(const abz_model* M, const double* pushed, double* z) {
  if (!M->mv) return pushed;
  for (int k = 0; k < M->ld; ++k) z[k] = k < M->d ? abz_mv_whiten1(M->mv, M->ld, k, pushed) : 0.0;
  return z;
}
/* spec: pairwise tree over components */
static double logprior_tree(const abz_model* M, const double* pushed) {
  double t[ABZ_MAX_D], zb[ABZ_MAX_D];
  const double* x = whitened(M, pushed, zb);
  for (int k = 0; k < M->ld; ++k) t[k] = abz_prior_logpdf1x(&M->prior[k], x[k], M->ext);
  return abz_tree_sum_small(t, M->ld);
}
/* literal: left-to-right sum, priors.jl:41-45 */
static double logprior_seq(const abz_model* M, const double* pushed) {
  double zb[ABZ_MAX_D];
  const double* x = whitened(M, pushed, zb);
  double s = abz_prior_logpdf1x(&M->prior[0], x[0], M->ext);
  for (int k = 1; k < M->d; ++k) s += abz_prior_logpdf1x(&M->prior[k], x[k], M->ext);
  return s;
}
ORC_API void orc_push_p(const abz_model* M, const double* theta, int64_t n, double* out) {
  for (int64_t i = 0; i < n; ++i) push_row(M, theta + i * M->ld, out + i * M->ld);
}
ORC_API void orc_logprior(const abz_model* M, const double* theta, int64_t n, int literal, double* out) {
  double p[ABZ_MAX_D];
  for (int64_t i = 0; i < n; ++i) {
    push_row(M, theta + i * M->ld, p);
    out[i] = literal ? logprior_seq(M, p) : logprior_tree(M, p);
  }
}

/* ---------------------------------------------------------------- simulators = dist!(theta, ve) (smc:137, mc:45, init:10,17)
 * theta arrives push_p-cast.  (i, epoch, purpose) address the particle's noise.    */
/* blob (may be NULL): the simulated data behind the distance -- the second return value of dist! */
static double sim_dist_b(const abz_model* M, const double* th, uint32_t i, uint32_t epoch, uint32_t purpose, double* blob) {
  const uint64_t seed = M->seed;
  switch (M->sim_id) {
    case ABZ_SIM_NORMAL1D: {
      double z0, z1;
      abz_normal_pair(abz_rng(seed, i, epoch, 0, purpose), ORC_T, &z0, &z1);
      double x = abz_fma(M->sim_p[0], z0, th[0]);
      if (blob) blob[0] = x;
      return fabs(x - M->data[0]);
    }
    case ABZ_SIM_MVN: {
      double sq[ABZ_MAX_D];
      for (int m = 0; 2 * m < M->ld; ++m) {
        double z[2];
        abz_normal_pair(abz_rng(seed, i, epoch, (uint32_t)m, purpose), ORC_T, &z[0], &z[1]);
        for (int c = 0; c < 2 && 2 * m + c < M->ld; ++c) {
          int k = 2 * m + c;
          if (k < M->d) {
            double x = abz_fma(M->sim_p[0], z[c], th[k]);
            if (blob) blob[k] = x;
            double e = x - M->data[k];
            sq[k] = e * e;
          } else {
            sq[k] = 0.0;
          }
        }
      }
      return abz_sqrt(abz_tree_sum_small(sq, M->ld));
    }
    case ABZ_SIM_DIRAC: {
      double x = th[0] * th[0] + 1.0;
      if (blob) blob[0] = x;
      return fabs(x - M->sim_p[0]);
    }
    case ABZ_SIM_QUAD2D: {
      double n1, n2;
      abz_normal_pair(abz_rng(seed, i, epoch, 0, purpose), ORC_T, &n1, &n2);
      double u = abz_u01_co(abz_rng(seed, i, epoch, 1, purpose).w0);
      double a = (th[0] + n1 * 0.01) - th[1] * th[1];
      double b = (th[1] - 1.0) + n2 * 0.01;
      if (blob) { blob[0] = a; blob[1] = b; }
      if (u < M->sim_p[0]) return ABZ_INF;
      return 50.0 * (a * a) + b * b;
    }
    case ABZ_SIM_MIXTURE: {
      double n1, n2;
      abz_normal_pair(abz_rng(seed, i, epoch, 0, purpose), ORC_T, &n1, &n2);
      uint64_t coin = abz_rng(seed, i, epoch, 1, purpose).w0 >> 63;
      double x = th[0] + (coin ? n2 : n1 * 0.1);
      if (blob) blob[0] = x;
      return fabs(x - M->sim_p[0]);
    }
    case ABZ_SIM_NORMDU: {
      double n1, n2;
      abz_normal_pair(abz_rng(seed, i, epoch, 0, purpose), ORC_T, &n1, &n2);
      double x = (th[0] * th[0] + th[1]) * (th[0] + n1 * 0.01);
      if (blob) blob[0] = x;
      return fabs(x - M->sim_p[0]);
    }
    case ABZ_SIM_WIENER: {
      double f = 0.95 + 0.1 * abz_u01_co(abz_rng(seed, i, epoch, 0, purpose).w0);
      double acc = 0.0;
      for (int t = 0; t < M->n_data; ++t) {
        double dt = (double)t;
        double v = abz_sqrt(th[0] * th[0] * dt * dt + th[1] * th[1] * dt) * f;
        if (blob && t < ABZ_MAX_BLOB) blob[t] = v;
        acc += fabs(v - M->data[t]);
      }
      return acc / (double)M->n_data;
    }
    case ABZ_SIM_LV: {
      const double a = th[0], b = th[1], c = th[2], e = th[3];
      double x = M->sim_p[0], y = M->sim_p[1];
      const double h = M->sim_p[2], h2 = 0.5 * h, h6 = h / 6.0;
      const int steps = (int)M->sim_p[3];
      const double sn = M->sim_p[4];
      const int nobs = M->n_data / 2;
      double acc = 0.0;
      for (int j = 0; j < nobs; ++j) {
        double z0, z1;
        abz_normal_pair(abz_rng(seed, i, epoch, (uint32_t)j, purpose), ORC_T, &z0, &z1);
        double ox = abz_fma(sn, z0, x), oy = abz_fma(sn, z1, y);
        if (blob && 2 * j + 1 < ABZ_MAX_BLOB) { blob[2 * j] = ox; blob[2 * j + 1] = oy; }
        double ex = ox - M->data[2 * j];
        double ey = oy - M->data[2 * j + 1];
        acc = abz_fma(ex, ex, acc);
        acc = abz_fma(ey, ey, acc);
        if (j + 1 == nobs) break;
        for (int s = 0; s < steps; ++s) {
          double k1x = x * abz_fma(-b, y, a), k1y = y * abz_fma(e, x, -c);
          double xa = abz_fma(h2, k1x, x), ya = abz_fma(h2, k1y, y);
          double k2x = xa * abz_fma(-b, ya, a), k2y = ya * abz_fma(e, xa, -c);
          double xb = abz_fma(h2, k2x, x), yb = abz_fma(h2, k2y, y);
          double k3x = xb * abz_fma(-b, yb, a), k3y = yb * abz_fma(e, xb, -c);
          double xc = abz_fma(h, k3x, x), yc = abz_fma(h, k3y, y);
          double k4x = xc * abz_fma(-b, yc, a), k4y = yc * abz_fma(e, xc, -c);
          x = abz_fma(h6, abz_fma(2.0, k2x, k1x) + abz_fma(2.0, k3x, k4x), x);   /* 2 k is exact: the same bits as (k1 + 2 k2) + (2 k3 + k4) */
          y = abz_fma(h6, abz_fma(2.0, k2y, k1y) + abz_fma(2.0, k3y, k4y), y);
        }
      }
      return abz_sqrt(acc);
    }
    case ABZ_SIM_SOCKS: {
      double ns = th[0];
      if (!(ns >= 0.0)) return ABZ_NAN;
      if (ns > 2147483647.0) ns = 2147483647.0;
      const uint32_t n_socks = (uint32_t)ns;
      const uint32_t n_pairs = (uint32_t)abz_rint(th[1] * abz_floor((double)n_socks * 0.5));
      const uint32_t n_want = (uint32_t)M->sim_p[2];
      const uint32_t m = n_socks < n_want ? n_socks : n_want;
      uint32_t pos[16];          /* picked positions, kept sorted */
      for (uint32_t t = 0; t < m; ++t) {
        abz_u64x2 w = abz_rng(seed, i, epoch, t >> 1, purpose);
        uint32_t j = abz_randint((t & 1) ? w.w1 : w.w0, n_socks - t);    /* j-th not yet picked position */
        uint32_t at = 0;
        while (at < t && j >= pos[at]) { ++j; ++at; }
        for (uint32_t q = t; q > at; --q) pos[q] = pos[q - 1];
        pos[at] = j;
      }
      uint32_t uniq = 0;          /* sorted positions => equal sock ids are adjacent */
      for (uint32_t t = 0; t < m; ++t) {
        const uint32_t id = pos[t] < 2 * n_pairs ? pos[t] >> 1 : pos[t] - n_pairs;
        const uint32_t idp = t ? (pos[t - 1] < 2 * n_pairs ? pos[t - 1] >> 1 : pos[t - 1] - n_pairs) : 0xFFFFFFFFu;
        uniq += (t == 0) || (id != idp);
      }
      const double pairs = (double)(m - uniq), odds = (double)uniq - (double)(m - uniq);
      if (blob) { blob[0] = pairs; blob[1] = odds; }
      return fabs(pairs - M->sim_p[0]) + fabs(odds - M->sim_p[1]);
    }
    default:
      return ABZ_NAN;
  }
}
static double sim_dist(const abz_model* M, const double* th, uint32_t i, uint32_t epoch, uint32_t purpose) {
  return sim_dist_b(M, th, i, epoch, purpose, NULL);
}
ORC_API double orc_sim_dist(const abz_model* M, const double* pushed, uint32_t i, uint32_t epoch, uint32_t purpose) {
  return sim_dist(M, pushed, i, epoch, purpose);
}

/* ---------------------------------------------------------------- blobs: stamps carried with the distances (abz_stamp,
 * abcdez_spec.h) and the rebuild of the simulated data from them.  The stamp arrays of the current / next
 * generation are bound once per call sequence (the checker's counterpart of abcdez_ctx_set_stamps).            */
static uint64_t* g_stamp_cur = NULL;
static uint64_t* g_stamp_nxt = NULL;
ORC_API void orc_set_stamps(uint64_t* cur, uint64_t* nxt) { g_stamp_cur = cur; g_stamp_nxt = nxt; }

ORC_API void orc_blob_eval(const abz_model* M, const double* theta, const uint64_t* stamp, int64_t N, double* blob,
                           int nbw, double* delta_out) {
#pragma omp parallel for schedule(static) num_threads(orc_threads_sim(M, N))
  for (int64_t s = 0; s < N; ++s) {
    double p[ABZ_MAX_D], b[ABZ_MAX_BLOB];
    for (int q = 0; q < ABZ_MAX_BLOB; ++q) b[q] = 0.0;
    push_row(M, theta + s * M->ld, p);
    const uint64_t st = stamp[s];
    delta_out[s] = sim_dist_b(M, p, abz_stamp_origin(st), abz_stamp_epoch(st),
                              abz_stamp_is_init(st) ? ABZ_RNG_INIT_SIM : ABZ_RNG_SIM, b);
    for (int q = 0; q < nbw; ++q) blob[s * nbw + q] = q < ABZ_MAX_BLOB ? b[q] : 0.0;
  }
}

/* ---------------------------------------------------------------- S1: abcde_init!  (init.jl:2-22 + prior draws smc:242-243, mc:117-118)
 * retry 0 is the initial draw of smc:242; every redraw of init.jl:15 bumps retry.  */
static void draw_prior_row(const abz_model* M, uint32_t i, uint32_t retry, double* th) {
  for (int m = 0; 2 * m < M->ld; ++m) {
    abz_u64x2 w = abz_rng(M->seed, i, retry, (uint32_t)m, ABZ_RNG_INIT_PRIOR);
    double z0, z1;
    abz_normal_pair(w, ORC_T, &z0, &z1);
    th[2 * m] = abz_prior_draw1(&M->prior[2 * m], w.w0, z0);
    if (2 * m + 1 < M->ld) th[2 * m + 1] = abz_prior_draw1(&M->prior[2 * m + 1], w.w1, z1);
    for (int c = 0; c < 2 && 2 * m + c < M->ld; ++c)
      if (M->prior[2 * m + c].family >= ABZ_PRIOR_BETA)
        th[2 * m + c] = abz_prior_draw_extx(&M->prior[2 * m + c], M->seed, i, retry, (uint32_t)(2 * m + c), ORC_T, M->ext);
  }
  if (M->mv) {                     /* correlated Normal prior: the row drawn so far is z ~ N(0, I); theta = mu + L z */
    double z[ABZ_MAX_D];
    for (int k = 0; k < M->ld; ++k) z[k] = th[k];
    for (int k = 0; k < M->ld; ++k) th[k] = k < M->d ? abz_mv_forward1(M->mv, M->ld, k, z) : 0.0;
  }
}
/* fills rows [i0, i0+n) of the FULL arrays theta / logpi / delta */
ORC_API int orc_init(const abz_model* M, double* theta, double* logpi, double* delta, int64_t i0, int64_t n) {
  int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad) num_threads(orc_threads_sim(M, n))
  for (int64_t g = i0; g < i0 + n; ++g) {
    uint32_t i = (uint32_t)g;
    double* th = theta + g * M->ld;
    double p[ABZ_MAX_D];
    uint32_t retry = 0;
    for (;;) {
      draw_prior_row(M, i, retry, th);
      push_row(M, th, p);
      double lp = logprior_tree(M, p);
      double dl = ABZ_NAN;
      if (abz_isfinite(lp)) dl = sim_dist(M, p, i, retry, ABZ_RNG_INIT_SIM);   /* init.jl:9-13,17 */
      logpi[g] = lp; delta[g] = dl;
      if (abz_isfinite(dl) && abz_isfinite(lp)) break;                          /* init.jl:14 */
      if (++retry >= ORC_MAX_RETRY) { bad = 1; break; }
    }
    if (g_stamp_cur) g_stamp_cur[g] = abz_stamp(i, retry, 1);
  }
  return bad ? -1 : 0;
}

/* ---------------------------------------------------------------- alive list (implicit in wsample(rng, 1:N, alive), smc:121,125) */
ORC_API int64_t orc_alive_compact(const uint8_t* alive, int64_t N, uint32_t* alive_idx, uint32_t* arank) {
  uint32_t r = 0;
  for (int64_t i = 0; i < N; ++i) {
    if (alive[i]) { alive_idx[r] = (uint32_t)i; arank[i] = r; ++r; }
    else arank[i] = ABZ_DEAD;
  }
  return (int64_t)r;
}

/* ---------------------------------------------------------------- S2+S3: abcdesmc_swarm!  (smc:106-153, copies smc:337-340)
 * Spec tier.  Processes particles [i0, i0+n_local) of a population of N whose
 * generation-t state is (theta, logpi, delta) -- all FULL arrays of N rows -- and
 * writes generation t+1 into (ntheta, nlogpi, ndelta), also full arrays.           */
ORC_API void orc_smc_swarm(const abz_model* M, const uint32_t* alive_idx, const uint32_t* arank, int64_t n_alive,
                           const double* theta, const double* logpi, const double* delta,
                           double* ntheta, double* nlogpi, double* ndelta,
                           double eps, double gamma0, double gsig,
                           int64_t i0, int64_t n_local, uint32_t sweep,
                           int64_t* nacc_out, int64_t* nsim_out) {
  const int ld = M->ld;
  int64_t nacc = 0, nsim = 0;
#pragma omp parallel for schedule(static) reduction(+ : nacc, nsim) num_threads(orc_threads_sim(M, n_local))
  for (int64_t i = i0; i < i0 + n_local; ++i) {
    const double* ti = theta + i * ld;
    double* to = ntheta + i * ld;
    /* identity. copies, smc:337-340 */
    for (int k = 0; k < ld; ++k) to[k] = ti[k];
    nlogpi[i] = logpi[i];
    ndelta[i] = delta[i];
    if (g_stamp_nxt) g_stamp_nxt[i] = g_stamp_cur[i];               /* nblobs = identity.(blobs), smc:340 */
    uint32_t ri = arank[i];
    if (ri == ABZ_DEAD) continue;                                   /* smc:114 */
    uint32_t ra, rb;
    abz_donor_ranks(abz_rng(M->seed, (uint32_t)i, sweep, 0, ABZ_RNG_DONOR), (uint32_t)n_alive, ri, &ra, &rb);
    const double* ta = theta + (int64_t)alive_idx[ra] * ld;        /* smc:119-126 */
    const double* tb = theta + (int64_t)alive_idx[rb] * ld;
    double z0, z1;
    abz_normal_pair(abz_rng(M->seed, (uint32_t)i, sweep, 0, ABZ_RNG_JITTER), ORC_T, &z0, &z1);
    double g = gamma0 * (1.0 + z0 * gsig);                          /* smc:128 */
    double tp[ABZ_MAX_D], pp[ABZ_MAX_D];
    for (int k = 0; k < ld; ++k) tp[k] = ti[k] + (ta[k] - tb[k]) * g;
    push_row(M, tp, pp);
    double lp = logprior_tree(M, pp);                               /* smc:134 */
    if (lp < 0.0 && !abz_isfinite(lp) && !abz_isnan(lp)) continue;  /* smc:135 */
    double dp = sim_dist(M, pp, (uint32_t)i, sweep, ABZ_RNG_SIM);   /* smc:137 */
    nsim += 1;                                                      /* smc:138 */
    double w = ((lp - logpi[i]) + abz_kernel_logpdf(M->abck, eps, dp)) - abz_kernel_logpdf(M->abck, eps, delta[i]);  /* smc:140-141, left to right */
    int acc = (0.0 <= w);
    if (!acc) {                                                     /* smc:145 */
      double u = abz_u01_open(abz_rng(M->seed, (uint32_t)i, sweep, 0, ABZ_RNG_ACCEPT).w0);
      acc = abz_log_tab(u, ORC_T) < w;
    }
    if (acc) {                                                      /* smc:146-150 */
      ndelta[i] = dp;
      for (int k = 0; k < ld; ++k) to[k] = tp[k];
      nlogpi[i] = lp;
      if (g_stamp_nxt) g_stamp_nxt[i] = abz_stamp((uint32_t)i, sweep, 0);   /* nblobs[i] = blob, smc:148 */
      nacc += 1;
    }
  }
  *nacc_out = nacc; *nsim_out = nsim;
}

/* Literal tier of the same sweep: donors by rejection around an O(N) weighted scan
 * exactly as wsample(rng, 1:N, alive) does it (StatsBase: t = rand*sum(w); walk the
 * cumulative weight until cw >= t), left-to-right logpdf.  Different consumption of
 * random numbers => equal in law only.  Used for equivalence tests and as the
 * "reference-faithful" CPU baseline.                                               */
static uint32_t ref_wsample_alive(const uint8_t* alive, int64_t N, int64_t n_alive, double u) {
  double t = u * (double)n_alive;          /* rand(rng) * sum(wv) */
  int64_t i = 0;
  double cw = (double)alive[0];
  while (cw < t && i < N - 1) { ++i; cw += (double)alive[i]; }
  return (uint32_t)i;
}
ORC_API void ref_smc_swarm(const abz_model* M, const uint8_t* alive, int64_t N,
                           const double* theta, const double* logpi, const double* delta,
                           double* ntheta, double* nlogpi, double* ndelta,
                           double eps, double gamma0, double gsig, uint32_t sweep,
                           int64_t* nacc_out, int64_t* nsim_out) {
  const int ld = M->ld;
  int64_t n_alive = 0;
  for (int64_t i = 0; i < N; ++i) n_alive += alive[i];
  int64_t nacc = 0, nsim = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : nacc, nsim) num_threads(orc_threads_sim(M, N))
  for (int64_t i = 0; i < N; ++i) {
    const double* ti = theta + i * ld;
    double* to = ntheta + i * ld;
    for (int k = 0; k < ld; ++k) to[k] = ti[k];
    nlogpi[i] = logpi[i];
    ndelta[i] = delta[i];
    if (!alive[i]) continue;
    uint32_t att = 0;
    int64_t a = i;
    while (a == i) {
      double u = abz_u01_co(abz_rng(M->seed, (uint32_t)i, sweep, att++, ABZ_RNG_DONOR).w0);
      a = ref_wsample_alive(alive, N, n_alive, u);
    }
    int64_t b = a;
    while (b == a || b == i) {
      double u = abz_u01_co(abz_rng(M->seed, (uint32_t)i, sweep, att++, ABZ_RNG_DONOR).w0);
      b = ref_wsample_alive(alive, N, n_alive, u);
    }
    const double* ta = theta + a * ld;
    const double* tb = theta + b * ld;
    double z0, z1;
    abz_normal_pair(abz_rng(M->seed, (uint32_t)i, sweep, 0, ABZ_RNG_JITTER), ORC_T, &z0, &z1);
    double g = gamma0 * (1.0 + z0 * gsig);
    double tp[ABZ_MAX_D], pp[ABZ_MAX_D];
    for (int k = 0; k < ld; ++k) tp[k] = ti[k] + (ta[k] - tb[k]) * g;
    push_row(M, tp, pp);
    double lp = logprior_seq(M, pp);
    if (lp < 0.0 && isinf(lp)) continue;
    double dp = sim_dist(M, pp, (uint32_t)i, sweep, ABZ_RNG_SIM);
    nsim += 1;
    double w = lp - logpi[i] + abz_kernel_logpdf(M->abck, eps, dp) - abz_kernel_logpdf(M->abck, eps, delta[i]);
    int acc = (0.0 <= w);
    if (!acc) {
      double u = abz_u01_open(abz_rng(M->seed, (uint32_t)i, sweep, 0, ABZ_RNG_ACCEPT).w0);
      acc = log(u) < w;
    }
    if (acc) {
      ndelta[i] = dp;
      for (int k = 0; k < ld; ++k) to[k] = tp[k];
      nlogpi[i] = lp;
      nacc += 1;
    }
  }
  *nacc_out = nacc; *nsim_out = nsim;
}

/* ---------------------------------------------------------------- fixed summation tree (spec header, "tile tree") */
static double tile_sum(const double* x, int64_t n) { /* n <= ABZ_TILE, missing = +0.0 */
  double s[256];
  for (int t = 0; t < 256; ++t) {
    double e[8];
    for (int m = 0; m < 4; ++m)
      for (int c = 0; c < 2; ++c) {
        int64_t k = (int64_t)m * 512 + 2 * t + c;
        e[2 * m + c] = k < n ? x[k] : 0.0;
      }
    s[t] = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
  }
  for (int st = 1; st < 256; st <<= 1)
    for (int t = 0; t < 256; t += 2 * st) s[t] = s[t] + s[t + st];
  return s[0];
}
ORC_API double orc_tree_sum(const double* x, int64_t n) {
  if (n <= ABZ_TILE) return tile_sum(x, n);
  int64_t nt = (n + ABZ_TILE - 1) / ABZ_TILE;
  double* part = (double*)malloc((size_t)nt * sizeof(double));
#pragma omp parallel for schedule(static) num_threads(orc_threads(nt * 2048))
  for (int64_t t = 0; t < nt; ++t) {
    int64_t lo = t * ABZ_TILE, len = n - lo < ABZ_TILE ? n - lo : ABZ_TILE;
    part[t] = tile_sum(x + lo, len);
  }
  double r = orc_tree_sum(part, nt);
  free(part);
  return r;
}

/* ---------------------------------------------------------------- S5 + driver lines smc:305-311, S6 (smc:8, :323)
 * ws[i] = exp(logpdf(K_new, D_i) - logpdf(K_old, D_i)) for alive i (smc:77-82);
 * wprod = Wns .* ws; wnorm = sum(wprod); Wns = wprod ./ wnorm; alive = Wns .> 0.
 * A dead particle has Wns = 0 and a stale finite ws, so its product is 0.           */
ORC_API void orc_smc_reweight(int abck, const double* delta, double* wns, uint8_t* alive, int64_t N,
                              double eps_old, double eps_new,
                              double* wnorm_out, double* ess_out, int64_t* n_alive_out) {
  double* wprod = (double*)malloc((size_t)N * sizeof(double));
#pragma omp parallel for schedule(static) num_threads(orc_threads(N))
  for (int64_t i = 0; i < N; ++i) {
    double w = 0.0;
    if (alive[i]) w = abz_exp(abz_kernel_logpdf(abck, eps_new, delta[i]) - abz_kernel_logpdf(abck, eps_old, delta[i]));
    wprod[i] = alive[i] ? wns[i] * w : 0.0;
  }
  double wnorm = orc_tree_sum(wprod, N);
  int64_t na = 0;
#pragma omp parallel for schedule(static) reduction(+ : na) num_threads(orc_threads(N))
  for (int64_t i = 0; i < N; ++i) {
    double W = wprod[i] / wnorm;
    wns[i] = W;
    alive[i] = (uint8_t)(W > 0.0);
    na += alive[i];
    wprod[i] = W * W;
  }
  *wnorm_out = wnorm;
  *ess_out = 1.0 / orc_tree_sum(wprod, N);
  *n_alive_out = na;
  free(wprod);
}
/* The same for an INDICATOR kernel when the weights are UNIFORM over the alive particles -- which they always are in the
 * reference's own runs with an indicator kernel: smc:266-270 sets 1/N, smc:310 divides equal products by their sum, smc:102
 * resets 1/N.  ws[i] is then 1 or 0 (types.jl:26-50), and in exact arithmetic
 *     wnorm = n_new / n_old,   Wns = 1 / n_new on the survivors,   1 / sum(Wns.^2) = n_new.
 * The reference's floating sums of n_new equal terms approximate these values to ~1e-16; the spec tier takes the values
 * themselves (IEEE divisions of exactly represented integers), which needs no summation tree and no pass over the weights:
 * one pass counts the survivors and writes the flags, the weights are a fill.  (The caller knows the weights are uniform;
 * the incoming values of wns[] are not read.)                                                                         */
ORC_API void orc_smc_reweight_uniform(int abck, const double* delta, double* wns, uint8_t* alive, int64_t N, double eps_new,
                                      double* wnorm_out, double* ess_out, int64_t* n_alive_out) {
  int64_t n_old = 0, n_new = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_old, n_new) num_threads(orc_threads(N))
  for (int64_t i = 0; i < N; ++i) {
    const int was = alive[i] != 0;
    const int is = was && abz_kernel_insupport(abck, eps_new, delta[i]);
    alive[i] = (uint8_t)is;
    n_old += was; n_new += is;
  }
  const double sumsq = 1.0 / (double)n_new;             /* = the weight of a survivor = sum(Wns.^2) in exact arithmetic */
#pragma omp parallel for schedule(static) num_threads(orc_threads(N))
  for (int64_t i = 0; i < N; ++i) wns[i] = alive[i] ? sumsq : 0.0;
  *wnorm_out = (double)n_new / (double)n_old;
  *ess_out = 1.0 / sumsq;
  *n_alive_out = n_new;
}
ORC_API double orc_get_ess(const double* wns, int64_t N) {
  double* sq = (double*)malloc((size_t)N * sizeof(double));
  for (int64_t i = 0; i < N; ++i) sq[i] = wns[i] * wns[i];
  double r = 1.0 / orc_tree_sum(sq, N);
  free(sq);
  return r;
}
/* literal: get_ess(Wns) = 1/sum(Wns.^2) and update_ws + normalisation with plain left-to-right sums */
ORC_API double ref_get_ess(const double* wns, int64_t N) {
  double s = 0.0;
  for (int64_t i = 0; i < N; ++i) s += wns[i] * wns[i];
  return 1.0 / s;
}
ORC_API void ref_smc_reweight(int abck, const double* delta, double* ws, double* wns, uint8_t* alive, int64_t N,
                              double eps_old, double eps_new, double* wnorm_out) {
  for (int64_t i = 0; i < N; ++i)
    if (alive[i]) ws[i] = exp(abz_kernel_logpdf(abck, eps_new, delta[i]) - abz_kernel_logpdf(abck, eps_old, delta[i]));
  double wnorm = 0.0;
  for (int64_t i = 0; i < N; ++i) wnorm += wns[i] * ws[i];
  for (int64_t i = 0; i < N; ++i) { wns[i] = (wns[i] * ws[i]) / wnorm; alive[i] = (uint8_t)(wns[i] > 0.0); }
  *wnorm_out = wnorm;
}

/* ---------------------------------------------------------------- S7: wsample_stratified!  (smc:15-56) */
/* literal: sequential fp walk, stratum bounds accumulated by + sval, r = lo + (hi-lo) u */
ORC_API void ref_wsample_stratified(const double* weights, int64_t n, const double* u, int64_t* inds) {
  double sval = 1.0 / (double)n;
  double wsum = 0.0;
  int64_t i = 0;               /* 1-based like the reference */
  double unif0 = 0.0, unif1 = 0.0;
  for (int64_t si = 0; si < n; ++si) {
    unif1 = unif0 + sval;
    double r = unif0 + (unif1 - unif0) * u[si];
    while (r > wsum && i < n) { i += 1; wsum += weights[i - 1]; }   /* clamp: see SURVEY 3.5 */
    unif0 = unif1;
    inds[si] = i - 1;          /* reported 0-based */
  }
}
/* spec: exact integer cumulative weights (spec header) */
ORC_API void orc_wsample_stratified(uint64_t seed, const double* wns, int64_t N, uint32_t draw, uint32_t* inds) {
  uint64_t* cum = (uint64_t*)malloc((size_t)N * sizeof(uint64_t));
  uint64_t c = 0;
  int64_t last_pos = 0;
  for (int64_t i = 0; i < N; ++i) {
    uint64_t f = abz_weight_fix(wns[i], (uint32_t)N);
    if (f) last_pos = i;
    c += f; cum[i] = c;
  }
  int64_t i = 0;
  for (int64_t s = 0; s < N; ++s) {
    uint64_t R = abz_stratum_point(seed, (uint32_t)N, (uint32_t)s, draw);
    while (i < N - 1 && !(cum[i] > R)) ++i;
    int64_t pick = i;
    if (!(cum[pick] > R)) pick = last_pos;   /* total mass rounded below R */
    if (pick > last_pos) pick = last_pos;
    inds[s] = (uint32_t)pick;
  }
  free(cum);
}
/* the uniforms the spec draw corresponds to, for feeding the literal walk in tests */
ORC_API void orc_stratum_uniforms(uint64_t seed, int64_t N, uint32_t draw, double* u) {
  for (int64_t s = 0; s < N; ++s) {
    const int b = abz_stratum_bits((uint32_t)N);
    uint64_t R = abz_stratum_point(seed, (uint32_t)N, (uint32_t)s, draw);
    u[s] = (double)(R & (((uint64_t)1 << b) - 1)) / (double)((uint64_t)1 << b);
  }
}

/* ---------------------------------------------------------------- S8: abcdesmc_resample!  (smc:85-104) */
ORC_API void orc_smc_resample_gather(const abz_model* M, const uint32_t* inds, int64_t N, int64_t i0, int64_t n_local,
                                     const double* theta, const double* logpi, const double* delta,
                                     double* ntheta, double* nlogpi, double* ndelta, double* wns, uint8_t* alive) {
  const int ld = M->ld;
#pragma omp parallel for schedule(static) num_threads(orc_threads(n_local))
  for (int64_t s = i0; s < i0 + n_local; ++s) {
    int64_t j = inds[s];
    memcpy(ntheta + s * ld, theta + j * ld, (size_t)ld * sizeof(double));  /* smc:96 */
    nlogpi[s] = logpi[j];                                                  /* smc:97 */
    ndelta[s] = delta[j];                                                  /* smc:98 */
    if (g_stamp_nxt) g_stamp_nxt[s] = g_stamp_cur[j];                      /* smc:99 */
    wns[s] = 1.0 / (double)N;                                              /* smc:102 */
    alive[s] = 1;                                                          /* smc:103 */
  }
}

/* ---------------------------------------------------------------- packed population: alive particles form a prefix
 * SPEC (deviation from the reference's "a particle keeps its index for life"): after every
 * reweight that kills particles the population is PARTITIONED -- the k-th dead position below n_new (the new number
 * of alive particles) swaps its whole state with the k-th alive position at or above n_new -- so that the alive
 * particles are the positions [0, n_alive).  Everything is keyed by position (random numbers, donor ranks, summation
 * trees, strata).  The reference's algorithm is symmetric under relabelling of the particles (smc:106-153 treats
 * every alive index alike; results are exchangeable), so the law of every output is unchanged; what the device gains
 * is that "alive rank r" IS "position r": no alive list, no index look-ups in front of the donor rows.
 * n_prev = length of the alive prefix before the reweight.  Returns n_new.                                       */
ORC_API int64_t orc_smc_partition(const abz_model* M, int64_t n_prev, uint8_t* alive, double* theta, double* logpi,
                                  double* delta, double* wns) {
  const int ld = M->ld;
  int64_t n_new = 0;
  for (int64_t p = 0; p < n_prev; ++p) n_new += alive[p] != 0;
  int64_t h = 0, f = n_new;
  double tmp[ABZ_MAX_D];
  for (;;) {
    while (h < n_new && alive[h]) ++h;
    while (f < n_prev && !alive[f]) ++f;
    if (h >= n_new || f >= n_prev) break;
    memcpy(tmp, theta + h * ld, (size_t)ld * 8);
    memcpy(theta + h * ld, theta + f * ld, (size_t)ld * 8);
    memcpy(theta + f * ld, tmp, (size_t)ld * 8);
    double t;
    t = logpi[h]; logpi[h] = logpi[f]; logpi[f] = t;
    t = delta[h]; delta[h] = delta[f]; delta[f] = t;
    t = wns[h]; wns[h] = wns[f]; wns[f] = t;
    alive[h] = 1; alive[f] = 0;
    if (g_stamp_cur) { uint64_t u = g_stamp_cur[h]; g_stamp_cur[h] = g_stamp_cur[f]; g_stamp_cur[f] = u; }
    ++h; ++f;
  }
  return n_new;
}

/* The same population in the device's storage: two row slots per position (slot0[N][ld], slot1[N][ld]) and one bit
 * per position naming the current one (bits, 32 positions per word, position p = bit p % 32 of word p / 32).  An
 * accepted proposal goes to the position's OTHER slot and its bit flips in bits_out; log-prior and distance are
 * updated in place.  Checker for abcdez_smc_partition / abcdez_smc_swarm_packed / abcdez_smc_replay_packed /
 * abcdez_smc_resample_gather_packed / abcdez_packed_gather (include/abcdez_hip.h).                               */
#define ORC_PBIT(bits, p) (((bits)[(p) >> 5] >> ((p) & 31)) & 1u)
/* rows of at most two doubles are kept DOUBLE-BUFFERED by the packed sweeps (csrc/abz_kernels.h, smc_swarm_packed_body): every
 * swept position moves to its other slot -- the proposal or a copy -- so the alive prefix shares one slot parity; the
 * kernel then takes the donors' parity from the own position, THIS restatement keeps reading every position's own bit */
#define ORC_DBUF(ld) ((ld) <= 2)
#define ORC_PROW(slot0, slot1, bit, p, ld) ((bit) ? (slot1) : (slot0)) + (int64_t)(p) * (ld)

ORC_API int64_t orc_packed_partition(const abz_model* M, int64_t N, int64_t n_prev, uint8_t* alive, const uint32_t* bits,
                                     uint32_t* bits_other, double* slot0, double* slot1, double* logpi, double* delta,
                                     double* wns) {
  const int ld = M->ld;
  int64_t n_new = 0;
  for (int64_t p = 0; p < n_prev; ++p) n_new += alive[p] != 0;
  int64_t h = 0, f = n_new;
  double tmp[ABZ_MAX_D];
  for (;;) {
    while (h < n_new && alive[h]) ++h;
    while (f < n_prev && !alive[f]) ++f;
    if (h >= n_new || f >= n_prev) break;
    double* rh = (double*)(ORC_PROW(slot0, slot1, ORC_PBIT(bits, h), h, ld));     /* contents swap, bits stay */
    double* rf = (double*)(ORC_PROW(slot0, slot1, ORC_PBIT(bits, f), f, ld));
    memcpy(tmp, rh, (size_t)ld * 8); memcpy(rh, rf, (size_t)ld * 8); memcpy(rf, tmp, (size_t)ld * 8);
    double t;
    t = logpi[h]; logpi[h] = logpi[f]; logpi[f] = t;
    t = delta[h]; delta[h] = delta[f]; delta[f] = t;
    t = wns[h]; wns[h] = wns[f]; wns[f] = t;
    alive[h] = 1; alive[f] = 0;
    if (g_stamp_cur) { uint64_t u = g_stamp_cur[h]; g_stamp_cur[h] = g_stamp_cur[f]; g_stamp_cur[f] = u; }
    ++h; ++f;
  }
  for (int64_t w = 0; w < (N + 31) / 32; ++w) bits_other[w] = bits[w];   /* both bit arrays agree outside the sweeps */
  return n_new;
}

ORC_API void orc_packed_gather(const uint32_t* bits, int64_t N, int ld, const double* slot0, const double* slot1, double* out) {
  for (int64_t p = 0; p < N; ++p) memcpy(out + p * ld, ORC_PROW(slot0, slot1, ORC_PBIT(bits, p), p, ld), (size_t)ld * 8);
}

/* the proposal of position r (alive rank r): smc:119-128 */
static void packed_proposal(const abz_model* M, const uint32_t* bits, int64_t n_alive, const double* slot0,
                            const double* slot1, uint32_t r, double gamma0, double gsig, uint32_t sweep, double* tp) {
  const int ld = M->ld;
  uint32_t ra, rb;
  abz_donor_ranks(abz_rng(M->seed, r, sweep, 0, ABZ_RNG_DONOR), (uint32_t)n_alive, r, &ra, &rb);
  const double* ti = ORC_PROW(slot0, slot1, ORC_PBIT(bits, r), r, ld);
  const double* ta = ORC_PROW(slot0, slot1, ORC_PBIT(bits, ra), ra, ld);
  const double* tb = ORC_PROW(slot0, slot1, ORC_PBIT(bits, rb), rb, ld);
  double z0, z1;
  abz_normal_pair(abz_rng(M->seed, r, sweep, 0, ABZ_RNG_JITTER), ORC_T, &z0, &z1);
  const double g = gamma0 * (1.0 + z0 * gsig);
  for (int k = 0; k < ld; ++k) tp[k] = ti[k] + (ta[k] - tb[k]) * g;
}

/* positions [r_lo, r_hi) of the alive prefix [0, n_alive); flags[p]: bit 0 accepted, bit 1 simulated (may be NULL).
 * bits_out gets the new bit of every position in [r_lo, r_hi); r_lo and r_hi must be multiples of 64 or the ends of
 * the prefix so that no word is shared with another caller.                                                    */
ORC_API void orc_smc_swarm_packed(const abz_model* M, const uint32_t* bits, uint32_t* bits_out, int64_t n_alive,
                                  int64_t r_lo, int64_t r_hi, double* slot0, double* slot1, double* logpi, double* delta,
                                  uint8_t* flags, double eps, double gamma0, double gsig, uint32_t sweep,
                                  int64_t* nacc_out, int64_t* nsim_out) {
  const int ld = M->ld;
  int64_t nacc = 0, nsim = 0;
  uint8_t* acc_tmp = (uint8_t*)calloc((size_t)(r_hi - r_lo) + 1, 1);
#pragma omp parallel for schedule(static) reduction(+ : nacc, nsim) num_threads(orc_threads_sim(M, r_hi - r_lo))
  for (int64_t r = r_lo; r < r_hi; ++r) {
    const uint32_t i = (uint32_t)r;
    double tp[ABZ_MAX_D], pp[ABZ_MAX_D];
    packed_proposal(M, bits, n_alive, slot0, slot1, i, gamma0, gsig, sweep, tp);
    push_row(M, tp, pp);
    const double lp = logprior_tree(M, pp);                                          /* smc:134 */
    int acc = 0, simulated = 0;
    if (!(lp < 0.0 && !abz_isfinite(lp) && !abz_isnan(lp))) {                        /* smc:135 */
      const double dp = sim_dist(M, pp, i, sweep, ABZ_RNG_SIM);                      /* smc:137 */
      nsim += 1;
      simulated = 1;
      const double w = ((lp - logpi[i]) + abz_kernel_logpdf(M->abck, eps, dp)) - abz_kernel_logpdf(M->abck, eps, delta[i]);   /* smc:140-141, left to right */
      acc = (0.0 <= w);
      if (!acc) acc = abz_log_tab(abz_u01_open(abz_rng(M->seed, i, sweep, 0, ABZ_RNG_ACCEPT).w0), ORC_T) < w;  /* smc:145 */
      if (acc) {                                                                     /* smc:146-150 */
        double* to = (double*)(ORC_PROW(slot0, slot1, ORC_PBIT(bits, i) ^ 1u, i, ld));
        for (int k = 0; k < ld; ++k) to[k] = tp[k];
        logpi[i] = lp; delta[i] = dp;
        if (g_stamp_cur) g_stamp_cur[i] = abz_stamp(i, sweep, 0);
        nacc += 1;
      }
    }
    if (!acc && ORC_DBUF(ld)) {        /* double-buffered rows: a rejected particle's row is COPIED to its other slot */
      const double* from = ORC_PROW(slot0, slot1, ORC_PBIT(bits, i), i, ld);
      double* to = (double*)(ORC_PROW(slot0, slot1, ORC_PBIT(bits, i) ^ 1u, i, ld));
      for (int k = 0; k < ld; ++k) to[k] = from[k];
    }
    acc_tmp[r - r_lo] = (uint8_t)acc;
    if (flags) flags[i] = (uint8_t)(acc | (simulated << 1));
  }
  for (int64_t r = r_lo; r < r_hi; ++r) {          /* sequential: several positions share a word */
    const uint32_t m = 1u << (r & 31);
    const uint32_t cur = bits[r >> 5] & m;
    bits_out[r >> 5] = (bits_out[r >> 5] & ~m) | ((acc_tmp[r - r_lo] || ORC_DBUF(ld)) ? (cur ^ m) : cur);
  }
  free(acc_tmp);
  *nacc_out = nacc; *nsim_out = nsim;
}

/* what a replica does for the positions it does not own: rebuild the accepted proposals (and their log-priors)
 * from the flags; counts both flag bits over the whole alive prefix */
ORC_API void orc_smc_replay_packed(const abz_model* M, const uint32_t* bits, uint32_t* bits_out, int64_t n_alive,
                                   int64_t skip_lo, int64_t skip_hi, double* slot0, double* slot1, double* logpi,
                                   const uint8_t* flags, double gamma0, double gsig, uint32_t sweep, int64_t* nacc_out,
                                   int64_t* nsim_out) {
  const int ld = M->ld;
  int64_t nacc = 0, nsim = 0;
#pragma omp parallel for schedule(static) reduction(+ : nacc, nsim) num_threads(orc_threads_sim(M, n_alive))
  for (int64_t r = 0; r < n_alive; ++r) {
    nacc += flags[r] & 1;
    nsim += (flags[r] >> 1) & 1;
    if (r >= skip_lo && r < skip_hi) continue;
    if (flags[r] & 1) {
      double tp[ABZ_MAX_D], pp[ABZ_MAX_D];
      packed_proposal(M, bits, n_alive, slot0, slot1, (uint32_t)r, gamma0, gsig, sweep, tp);
      double* to = (double*)(ORC_PROW(slot0, slot1, ORC_PBIT(bits, r) ^ 1u, r, ld));
      for (int k = 0; k < ld; ++k) to[k] = tp[k];
      push_row(M, tp, pp);
      logpi[r] = logprior_tree(M, pp);              /* the owner stored the same value (smc:147) */
      if (g_stamp_cur) g_stamp_cur[r] = abz_stamp((uint32_t)r, sweep, 0);
    } else if (ORC_DBUF(ld)) {
      const double* from = ORC_PROW(slot0, slot1, ORC_PBIT(bits, r), r, ld);
      double* to = (double*)(ORC_PROW(slot0, slot1, ORC_PBIT(bits, r) ^ 1u, r, ld));
      for (int k = 0; k < ld; ++k) to[k] = from[k];
    }
  }
  for (int64_t r = 0; r < n_alive; ++r) {
    if (r >= skip_lo && r < skip_hi) continue;
    const uint32_t m = 1u << (r & 31);
    const uint32_t cur = bits[r >> 5] & m;
    bits_out[r >> 5] = (bits_out[r >> 5] & ~m) | (((flags[r] & 1) || ORC_DBUF(ld)) ? (cur ^ m) : cur);
  }
  *nacc_out = nacc; *nsim_out = nsim;
}

/* S8 on the packed store: source = current row of inds[s], destination = the OTHER slot of s (never a current row);
 * afterwards every bit flips -- in both bit arrays */
ORC_API void orc_smc_resample_gather_packed(const abz_model* M, const uint32_t* inds, int64_t N, uint32_t* bits,
                                            uint32_t* bits_other, double* slot0, double* slot1, const double* logpi,
                                            const double* delta, double* nlogpi, double* ndelta, double* wns,
                                            uint8_t* alive) {
  const int ld = M->ld;
  /* double-buffered rows: every destination is the slot the alive prefix is NOT in (parity of position 0), whatever a
   * dead position's stale bit says; afterwards all N positions have that parity */
  const uint32_t newp = (bits[0] & 1u) ^ 1u;
#pragma omp parallel for schedule(static) num_threads(orc_threads(N))
  for (int64_t s = 0; s < N; ++s) {
    const int64_t j = inds[s];
    const uint32_t dst = ORC_DBUF(ld) ? newp : (ORC_PBIT(bits, s) ^ 1u);
    memcpy((double*)(ORC_PROW(slot0, slot1, dst, s, ld)), ORC_PROW(slot0, slot1, ORC_PBIT(bits, j), j, ld),
           (size_t)ld * sizeof(double));                                   /* smc:96 */
    nlogpi[s] = logpi[j];                                                  /* smc:97 */
    ndelta[s] = delta[j];                                                  /* smc:98 */
    if (g_stamp_nxt) g_stamp_nxt[s] = g_stamp_cur[j];                      /* smc:99 */
    wns[s] = 1.0 / (double)N;                                              /* smc:102 */
    alive[s] = 1;                                                          /* smc:103 */
  }
  for (int64_t w = 0; w < (N + 31) / 32; ++w) {
    bits[w] = ORC_DBUF(ld) ? (newp ? 0xFFFFFFFFu : 0u) : ~bits[w];
    bits_other[w] = bits[w];
  }
}

/* ---------------------------------------------------------------- S9: quantile(Ds[alive], alpha)  (smc:301)
 * Statistics.quantile default = type 7: h = (n-1) p + 1; j = clamp(floor(h), 1, n-1);
 * g = h - j; q = x_(j) + g (x_(j+1) - x_(j)).                                       */
static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}
ORC_API double orc_quantile7(double xj, double xj1, double g) { return xj + g * (xj1 - xj); }
ORC_API void orc_quantile_pos(int64_t n, double p, int64_t* j_out, double* g_out) { /* j is 1-based */
  double h = (double)(n - 1) * p + 1.0;
  double fl = floor(h);
  int64_t j = (int64_t)fl;
  if (j < 1) j = 1;
  if (j > n - 1) j = n - 1 > 1 ? n - 1 : 1;
  *j_out = j; *g_out = h - (double)j;
}
ORC_API double orc_quantile_alive(const double* delta, const uint8_t* alive, int64_t N, double p,
                                  double* xj_out, double* xj1_out) {
  double* v = (double*)malloc((size_t)N * sizeof(double));
  int64_t n = 0;
  for (int64_t i = 0; i < N; ++i) if (alive[i]) v[n++] = delta[i];
  if (n == 0) { free(v); return ABZ_NAN; }
  qsort(v, (size_t)n, sizeof(double), cmp_double);
  int64_t j; double g;
  orc_quantile_pos(n, p, &j, &g);
  double xj = v[j - 1], xj1 = n > 1 ? v[j] : v[j - 1];
  free(v);
  if (xj_out) *xj_out = xj;
  if (xj1_out) *xj1_out = xj1;
  return orc_quantile7(xj, xj1, g);
}

/* ---------------------------------------------------------------- S10: driver reductions (smc:286,364; mc:133,146,156,163) */
ORC_API void orc_extrema(const double* delta, int64_t N, double* lo, double* hi) {
  double a = delta[0], b = delta[0];
  for (int64_t i = 1; i < N; ++i) { if (delta[i] < a) a = delta[i]; if (delta[i] > b) b = delta[i]; }
  *lo = a; *hi = b;
}
ORC_API int64_t orc_count_gt(const double* delta, int64_t N, double thr) {
  int64_t c = 0;
  for (int64_t i = 0; i < N; ++i) c += delta[i] > thr;
  return c;
}

/* ---------------------------------------------------------------- S4: abcdemc_swarm!  (mc:5-61)
 * order = particle indices sorted by (delta, index); sorted_delta = delta[order].
 * The "better particle" set {j : D_j <= D_i} (mc:23) is the first cnt entries of
 * order, cnt = upper_bound(sorted_delta, D_i); s = order[randint(cnt)].
 * (The reference enumerates the same set in index order; any fixed enumeration
 * gives the same law.)                                                             */
typedef struct { double d; uint32_t i; } orc_key;
static int cmp_key(const void* a, const void* b) {
  const orc_key* x = (const orc_key*)a; const orc_key* y = (const orc_key*)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return (x->i > y->i) - (x->i < y->i);
}
/* order = [particles with Ds <= eps_pop, in index order] ++ [the others sorted by (Ds, index)];
 * sorted_delta[p] = max(Ds[order[p]], eps_pop) (non-decreasing).  A draw (mc:20-24) happens only for Ds[i] > eps_pop,
 * and every particle of the first block then belongs to the candidate set {j : Ds[j] <= Ds[i]}, so the first block
 * needs no sorting: the enumeration differs from "all sorted" only inside that block -- same set, same law.      */
static int64_t upper_bound_d(const double* v, int64_t n, double x);
/* cnt[i] = #{j : Ds[j] <= Ds[i]} = upper_bound(sorted_delta, Ds[i]) for Ds[i] > eps_pop, 0 for the first block */
ORC_API void orc_mc_rank_prepare(const double* delta, int64_t N, double eps_pop, uint32_t* order, double* sorted_delta,
                                 uint32_t* cnt) {
  orc_key* k = (orc_key*)malloc((size_t)N * sizeof(orc_key));
  int64_t nA = 0, nB = 0;
  for (int64_t i = 0; i < N; ++i)
    if (delta[i] <= eps_pop) { order[nA] = (uint32_t)i; sorted_delta[nA] = eps_pop; ++nA; }
    else { k[nB].d = delta[i]; k[nB].i = (uint32_t)i; ++nB; }
  qsort(k, (size_t)nB, sizeof(orc_key), cmp_key);
  for (int64_t i = 0; i < nB; ++i) { order[nA + i] = k[i].i; sorted_delta[nA + i] = k[i].d; }
  free(k);
  for (int64_t i = 0; i < N; ++i) cnt[i] = delta[i] <= eps_pop ? 0u : (uint32_t)upper_bound_d(sorted_delta, N, delta[i]);
}
static int64_t upper_bound_d(const double* v, int64_t n, double x) { /* #elements <= x */
  int64_t lo = 0, hi = n;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (v[mid] <= x) lo = mid + 1; else hi = mid; }
  return lo;
}
/* The three index draws of one proposal (mc:18-32), spec tier: better particle s by rank (order != NULL) or by rejection
 * (order == NULL; include/abcdez_spec.h says when), donors (a, b) by rank-skip.  orc_mc_swarm calls this; the tests call it
 * directly to compare its law with ref_mc_draws. */
ORC_API void orc_mc_draws(const abz_model* M, const uint32_t* order, const uint32_t* cnt_of, int64_t N, const double* delta,
                          double eps_pop, double eps_target, int64_t i, uint32_t sweep, uint32_t* s_out, uint32_t* a_out,
                          uint32_t* b_out, int* exhausted_out) {
  const double di = delta[i];
  const double eps = di <= eps_target ? eps_target : eps_pop;           /* mc:19 */
  uint32_t s = (uint32_t)i;
  int exhausted = 0;
  if (di > eps) {                                                       /* mc:20-24 */
    /* mc:23; order == NULL: by rejection (include/abcdez_spec.h, abz_mc_draws_by_rejection says when) */
    s = order ? order[abz_randint(abz_rng(M->seed, (uint32_t)i, sweep, 0, ABZ_RNG_BETTER).w0, cnt_of[i])]
              : abz_mc_better_by_rejection(M->seed, (uint32_t)i, sweep, delta, (uint32_t)N, di, &exhausted);
  }
  abz_donor_ranks(abz_rng(M->seed, (uint32_t)i, sweep, 0, ABZ_RNG_DONOR), (uint32_t)N, s, a_out, b_out);   /* mc:25-32: uniform over all N */
  *s_out = s;
  if (exhausted_out) *exhausted_out = exhausted;
}

/* ---------------------------------------------------------------- S4, LITERAL tier: abcdemc_swarm! statement by statement (mc:5-61)
 * What the spec tier replaces is kept here as the reference writes it:
 *   mc:23     s = rand(rng, (1:nparticles)[Ds .<= Ds[i]]) -- the mask is materialised in INDEX order (an O(N) scan per
 *             drawing particle) and one uniform picks its k-th member;
 *   mc:25-32  a = s; while a == s: a = rand(rng, 1:N)   /   b = a; while b == a || b == s: b = rand(rng, 1:N) -- two
 *             rejection loops, one fresh uniform integer per trial;
 *   mc:41     logpdf summed left to right (priors.jl:40-46);
 *   mc:43     log(rand(rng)) > min(0, w_prior) && continue -- the uniform is drawn for EVERY particle, libm log;
 *   mc:54     dp <= max(eps, Ds[i]).
 * It consumes the counter-based stream differently from the spec tier (one Philox block per trial, sub-index = trial
 * number), so the two agree IN LAW only: tests/test_oracle_equivalence.py compares the joint law of (s, a, b) and the
 * moved populations, for the spec's draw by rank and by rejection.  Test infrastructure / "reference-faithful" baseline. */
ORC_API void ref_mc_draws(const abz_model* M, int64_t N, const double* delta, double eps_pop, double eps_target, int64_t i,
                          uint32_t sweep, uint32_t* s_out, uint32_t* a_out, uint32_t* b_out, uint32_t* trials_out) {
  uint32_t att = 0;
  uint32_t s = (uint32_t)i;                                             /* mc:18 */
  const double eps = delta[i] <= eps_target ? eps_target : eps_pop;     /* mc:19 */
  if (delta[i] > eps) {                                                 /* mc:20 */
    int64_t cnt = 0;                                                    /* mc:23: (1:N)[Ds .<= Ds[i]] */
    for (int64_t j = 0; j < N; ++j) cnt += delta[j] <= delta[i];
    const uint32_t pick = abz_randint(abz_rng(M->seed, (uint32_t)i, sweep, 0, ABZ_RNG_BETTER).w0, (uint32_t)cnt);
    int64_t seen = 0;
    for (int64_t j = 0; j < N; ++j)
      if (delta[j] <= delta[i]) { if (seen == (int64_t)pick) { s = (uint32_t)j; break; } ++seen; }
  }
  uint32_t a = s;                                                       /* mc:25 */
  while (a == s) a = abz_randint(abz_rng(M->seed, (uint32_t)i, sweep, att++, ABZ_RNG_DONOR).w0, (uint32_t)N);   /* mc:26-28 */
  uint32_t b = a;                                                       /* mc:29 */
  while (b == a || b == s) b = abz_randint(abz_rng(M->seed, (uint32_t)i, sweep, att++, ABZ_RNG_DONOR).w0, (uint32_t)N);   /* mc:30-32 */
  *s_out = s; *a_out = a; *b_out = b;
  if (trials_out) *trials_out = att;
}
ORC_API void ref_mc_swarm(const abz_model* M, int64_t N, const double* theta, const double* logpi, const double* delta,
                          double* ntheta, double* nlogpi, double* ndelta, double eps_pop, double eps_target, double gamma0,
                          double gsig, uint32_t sweep, int64_t* nsim_out) {
  const int ld = M->ld;
  int64_t nsim = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : nsim) num_threads(orc_threads_sim(M, N))
  for (int64_t i = 0; i < N; ++i) {
    const double* ti = theta + i * ld;
    double* to = ntheta + i * ld;
    for (int k = 0; k < ld; ++k) to[k] = ti[k];                         /* nthetas = identity.(thetas) ..., mc:140-143 */
    nlogpi[i] = logpi[i];
    ndelta[i] = delta[i];
    uint32_t s, a, b;
    ref_mc_draws(M, N, delta, eps_pop, eps_target, i, sweep, &s, &a, &b, NULL);     /* mc:18-32 */
    const double eps = delta[i] <= eps_target ? eps_target : eps_pop;  /* mc:19 */
    const double* ts = theta + (int64_t)s * ld;
    const double* ta = theta + (int64_t)a * ld;
    const double* tb = theta + (int64_t)b * ld;
    double z0, z1;
    abz_normal_pair(abz_rng(M->seed, (uint32_t)i, sweep, 0, ABZ_RNG_JITTER), ORC_T, &z0, &z1);
    const double g = gamma0 * (1.0 + z0 * gsig);                        /* mc:34 */
    double tp[ABZ_MAX_D], pp[ABZ_MAX_D];
    for (int k = 0; k < ld; ++k) tp[k] = ts[k] + (ta[k] - tb[k]) * g;   /* op(+, thetas[s], op(*, op(-, thetas[a], thetas[b]), gamma)) */
    push_row(M, tp, pp);
    const double lp = logprior_seq(M, pp);                              /* mc:41 */
    const double w_prior = lp - logpi[i];                               /* mc:42 */
    const double u = abz_u01_open(abz_rng(M->seed, (uint32_t)i, sweep, 0, ABZ_RNG_ACCEPT).w0);
    const double mn = w_prior < 0.0 ? w_prior : 0.0;                    /* Julia's min(0, w_prior); NaN propagates ... */
    if (!abz_isnan(w_prior) && log(u) > mn) continue;                   /* mc:43 (... and `x > NaN` is false: no continue) */
    nsim += 1;                                                          /* mc:44 */
    const double dp = sim_dist(M, pp, (uint32_t)i, sweep, ABZ_RNG_SIM); /* mc:45 */
    if (dp <= (eps > delta[i] ? eps : delta[i])) {                      /* mc:54 */
      ndelta[i] = dp;                                                   /* mc:55-57 */
      for (int k = 0; k < ld; ++k) to[k] = tp[k];
      nlogpi[i] = lp;
    }
  }
  *nsim_out = nsim;
}

ORC_API void orc_mc_swarm(const abz_model* M, const uint32_t* order, const uint32_t* cnt_of, int64_t N,
                          const double* theta, const double* logpi, const double* delta,
                          double* ntheta, double* nlogpi, double* ndelta,
                          double eps_pop, double eps_target, double gamma0, double gsig,
                          int64_t i0, int64_t n_local, uint32_t sweep, int64_t* nsim_out) {
  const int ld = M->ld;
  int64_t nsim = 0;
#pragma omp parallel for schedule(static) reduction(+ : nsim) num_threads(orc_threads_sim(M, n_local))
  for (int64_t i = i0; i < i0 + n_local; ++i) {
    const double* ti = theta + i * ld;
    double* to = ntheta + i * ld;
    for (int k = 0; k < ld; ++k) to[k] = ti[k];                         /* mc:140-143 */
    nlogpi[i] = logpi[i];
    ndelta[i] = delta[i];
    if (g_stamp_nxt) g_stamp_nxt[i] = g_stamp_cur[i];
    double di = delta[i];
    double eps = di <= eps_target ? eps_target : eps_pop;               /* mc:19 */
    uint32_t s, a, b;                                                   /* mc:18-32 */
    orc_mc_draws(M, order, cnt_of, N, delta, eps_pop, eps_target, i, sweep, &s, &a, &b, NULL);
    const double* ts = theta + (int64_t)s * ld;
    const double* ta = theta + (int64_t)a * ld;
    const double* tb = theta + (int64_t)b * ld;
    double z0, z1;
    abz_normal_pair(abz_rng(M->seed, (uint32_t)i, sweep, 0, ABZ_RNG_JITTER), ORC_T, &z0, &z1);
    double g = gamma0 * (1.0 + z0 * gsig);                              /* mc:34 */
    double tp[ABZ_MAX_D], pp[ABZ_MAX_D];
    for (int k = 0; k < ld; ++k) tp[k] = ts[k] + (ta[k] - tb[k]) * g;
    push_row(M, tp, pp);
    double lp = logprior_tree(M, pp);                                   /* mc:41 */
    double w_prior = lp - logpi[i];                                     /* mc:42 */
    double u = abz_u01_open(abz_rng(M->seed, (uint32_t)i, sweep, 0, ABZ_RNG_ACCEPT).w0);
    double mn = w_prior < 0.0 ? w_prior : 0.0;                          /* min(0, w_prior); NaN -> compares false below */
    if (abz_isnan(w_prior)) mn = w_prior;
    if (abz_log_tab(u, ORC_T) > mn) continue;                                      /* mc:43 */
    nsim += 1;                                                          /* mc:44 */
    double dp = sim_dist(M, pp, (uint32_t)i, sweep, ABZ_RNG_SIM);       /* mc:45 */
    double thr = eps > di ? eps : di;                                   /* max(eps, D_i) */
    if (dp <= thr) {                                                    /* mc:54-58 */
      ndelta[i] = dp;
      for (int k = 0; k < ld; ++k) to[k] = tp[k];
      nlogpi[i] = lp;
      if (g_stamp_nxt) g_stamp_nxt[i] = abz_stamp((uint32_t)i, sweep, 0);
    }
  }
  *nsim_out = nsim;
}
