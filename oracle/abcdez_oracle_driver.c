/*
 * abcdez_oracle_driver.c -- CPU restatement of the host drivers abcdesmc!
 * (src/abcdez_smc.jl:215-394) and abcdemc! (src/abcdez_mc.jl:102-172), built on the
 * spec-tier population functions of abcdez_oracle.c.
 *
 * TEST INFRASTRUCTURE ONLY (see abcdez_oracle.c).  Used to check the product's
 * host loop + HIP kernels end to end (same seed => same eps schedule, logZ and
 * final population, bit for bit) and as the timed CPU baseline in bench.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/abcdez_spec.h"

#define ORC_API __attribute__((visibility("default")))

int orc_init(const abz_model*, double*, double*, double*, int64_t, int64_t);
int64_t orc_alive_compact(const uint8_t*, int64_t, uint32_t*, uint32_t*);
int64_t orc_smc_partition(const abz_model*, int64_t, uint8_t*, double*, double*, double*, double*);
void orc_smc_swarm(const abz_model*, const uint32_t*, const uint32_t*, int64_t, const double*, const double*,
                   const double*, double*, double*, double*, double, double, double, int64_t, int64_t, uint32_t,
                   int64_t*, int64_t*);
void orc_smc_reweight(int, const double*, double*, uint8_t*, int64_t, double, double, double*, double*, int64_t*);
void orc_smc_reweight_uniform(int, const double*, double*, uint8_t*, int64_t, double, double*, double*, int64_t*);
double orc_get_ess(const double*, int64_t);
void orc_wsample_stratified(uint64_t, const double*, int64_t, uint32_t, uint32_t*);
void orc_smc_resample_gather(const abz_model*, const uint32_t*, int64_t, int64_t, int64_t, const double*,
                             const double*, const double*, double*, double*, double*, double*, uint8_t*);
double orc_quantile_alive(const double*, const uint8_t*, int64_t, double, double*, double*);
void orc_extrema(const double*, int64_t, double*, double*);
int64_t orc_count_gt(const double*, int64_t, double);
void orc_mc_rank_prepare(const double*, int64_t, double, uint32_t*, double*, uint32_t*);
void orc_mc_swarm(const abz_model*, const uint32_t*, const uint32_t*, int64_t, const double*, const double*,
                  const double*, double*, double*, double*, double, double, double, double, int64_t, int64_t,
                  uint32_t, int64_t*);

/* ping-pong population buffers: the reference's (thetas, nthetas) etc. (smc:337-350) */
typedef struct { double *theta, *logpi, *delta; } pop_t;
static pop_t pop_alloc(int64_t N, int ld) {
  pop_t p;
  p.theta = (double*)malloc((size_t)N * ld * sizeof(double));
  p.logpi = (double*)malloc((size_t)N * sizeof(double));
  p.delta = (double*)malloc((size_t)N * sizeof(double));
  return p;
}
static void pop_free(pop_t p) { free(p.theta); free(p.logpi); free(p.delta); }
static void pop_swap(pop_t* a, pop_t* b) { pop_t t = *a; *a = *b; *b = t; }

typedef struct {
  /* keyword arguments of abcdesmc! (smc:215-220) */
  int64_t nparticles;
  double eps_target, alpha, delta_ess;
  int64_t nsims_max;
  int32_t Kmcmc, max_iters;
  double Kmcmc_min, facc_stop, facc_min, facc_tune;
  /* results (smc:388-393) */
  double eps, logZ;
  int64_t iters, nsims_total, updates_total; /* updates = sum over sweeps of n_alive */
  int32_t n_hist, no_alive;
  /* in: 1 = packed population (abcdez_oracle.c, orc_smc_partition): alive particles are kept as a prefix */
  int32_t packed, reserved;
} orc_smc_run;

/* Outputs: theta[N][ld] (unpushed internal state), logpi, delta (= r.C), wns, alive;
 * history arrays (verboseout, smc:284-292,362-370) need max_iters+1 slots each.     */
ORC_API int orc_abcdesmc(const abz_model* M, orc_smc_run* R,
                         double* theta_out, double* logpi_out, double* delta_out, double* wns, uint8_t* alive,
                         double* h_eps, double* h_lo, double* h_hi, double* h_logZ, double* h_ess,
                         double* h_facc, double* h_gamma0, int32_t* h_K) {
  const int64_t N = R->nparticles;
  const int ld = M->ld;
  pop_t cur = pop_alloc(N, ld), nxt = pop_alloc(N, ld);
  uint32_t* alive_idx = (uint32_t*)malloc((size_t)N * 4);
  uint32_t* arank = (uint32_t*)malloc((size_t)N * 4);
  uint32_t* inds = (uint32_t*)malloc((size_t)N * 4);
  int rc = orc_init(M, cur.theta, cur.logpi, cur.delta, 0, N);      /* smc:242-252 */

  double eps = INFINITY, eps_k = INFINITY;                          /* smc:255-256 */
  const double ess_min = (double)N * R->delta_ess;                  /* smc:259 */
  double logZ = 0.0;                                                /* smc:263 */
  for (int64_t i = 0; i < N; ++i) { wns[i] = 1.0 / (double)N; alive[i] = 1; }   /* smc:266-270 */
  double ess = 0.0, facc = 1.0;
  int Ki = R->Kmcmc;
  double gamma0 = 2.38 / sqrt(2.0 * (double)M->d);                  /* smc:280 */
  const double gsig = 1e-5;                                         /* smc:281 */
  int64_t nsims_total = 0, updates = 0, iters = 0;
  uint32_t sweep = 0, draw = 0;
  int nh = 0, no_alive = 0;
  int64_t n_prev = N;                 /* packed: length of the alive prefix */
  if (rc == 0) {
    double lo, hi; orc_extrema(cur.delta, N, &lo, &hi);             /* smc:284-292 */
    h_eps[nh] = eps; h_lo[nh] = lo; h_hi[nh] = hi; h_logZ[nh] = logZ; h_ess[nh] = orc_get_ess(wns, N);
    h_facc[nh] = facc; h_gamma0[nh] = gamma0; h_K[nh] = Ki; ++nh;
  }
  while (rc == 0) {                                                 /* smc:295 */
    iters += 1;
    double q = orc_quantile_alive(cur.delta, alive, N, R->alpha, 0, 0);
    eps = fmax(fmin(q, eps), R->eps_target);                        /* smc:301 */
    double wnorm; int64_t n_alive;
    /* smc:305-311.  Indicator kernels: the weights are uniform over the alive particles throughout (1/N at smc:266-270 and
     * after every resampling, 1/n_alive after every reweight), so the closed forms apply (abcdez_oracle.c) */
    if (M->abck == ABZ_K_INDICATOR || M->abck == ABZ_K_INDICATOR_STRICT)
      orc_smc_reweight_uniform(M->abck, cur.delta, wns, alive, N, eps, &wnorm, &ess, &n_alive);
    else
      orc_smc_reweight(M->abck, cur.delta, wns, alive, N, eps_k, eps, &wnorm, &ess, &n_alive);
    logZ += log(wnorm);                                             /* smc:315 */
    int64_t naccs = 0;                                              /* smc:318 */
    Ki = R->Kmcmc;
    if (facc < R->facc_min) gamma0 *= R->facc_tune;                 /* smc:320 */
    if (n_alive > 0 && ess < ess_min) {                             /* smc:323-326 */
      orc_wsample_stratified(M->seed, wns, N, draw++, inds);
      orc_smc_resample_gather(M, inds, N, 0, N, cur.theta, cur.logpi, cur.delta,
                              nxt.theta, nxt.logpi, nxt.delta, wns, alive);
      pop_swap(&cur, &nxt);
      ess = orc_get_ess(wns, N);
      n_alive = N;
      n_prev = N;
    }
    if (R->packed && n_alive < n_prev) {
      orc_smc_partition(M, n_prev, alive, cur.theta, cur.logpi, cur.delta, wns);
      n_prev = n_alive;
    }
    if (n_alive >= 3) {                /* the reference's donor loops need 3 alive (smc:119-126) */
      orc_alive_compact(alive, N, alive_idx, arank);   /* packed: the identity over [0, n_alive) */
      for (int k = 1; k <= R->Kmcmc; ++k) {                         /* smc:336-353 */
        int64_t nacc, nsim;
        orc_smc_swarm(M, alive_idx, arank, n_alive, cur.theta, cur.logpi, cur.delta,
                      nxt.theta, nxt.logpi, nxt.delta, eps, gamma0, gsig, 0, N, sweep++, &nacc, &nsim);
        pop_swap(&cur, &nxt);                                       /* smc:347-350 */
        naccs += nacc; nsims_total += nsim; updates += n_alive;
        if ((double)naccs / (double)n_alive >= R->Kmcmc_min) { Ki = k; break; }   /* smc:352 */
      }
    }
    facc = (double)naccs / ((double)n_alive * (double)Ki);          /* smc:357 */
    eps_k = eps;                                                    /* smc:360 */
    if (nh <= R->max_iters) {                                       /* smc:362-370 */
      double lo, hi; orc_extrema(cur.delta, N, &lo, &hi);
      h_eps[nh] = eps; h_lo[nh] = lo; h_hi[nh] = hi; h_logZ[nh] = logZ; h_ess[nh] = ess;
      h_facc[nh] = facc; h_gamma0[nh] = gamma0; h_K[nh] = Ki; ++nh;
    }
    if (n_alive < 3) { no_alive = 1; break; }                       /* smc:375 */
    if (eps <= R->eps_target || nsims_total >= R->nsims_max || facc < R->facc_stop) break;   /* smc:376 */
    if (iters >= R->max_iters) break;
  }
  memcpy(theta_out, cur.theta, (size_t)N * ld * sizeof(double));
  memcpy(logpi_out, cur.logpi, (size_t)N * sizeof(double));
  memcpy(delta_out, cur.delta, (size_t)N * sizeof(double));
  R->eps = eps; R->logZ = logZ; R->iters = iters; R->nsims_total = nsims_total; R->updates_total = updates;
  R->n_hist = nh; R->no_alive = no_alive;
  free(alive_idx); free(arank); free(inds);
  pop_free(cur); pop_free(nxt);
  return rc;
}

typedef struct {
  int64_t nparticles;      /* mc:103 */
  int32_t generations;     /* mc:103 */
  int32_t reserved;
  double eps_target;
  /* results (mc:171) */
  int64_t nsims_total;
  int32_t reached_eps;
  int32_t reserved2;
  double complete;
} orc_mc_run;

ORC_API int orc_abcdemc(const abz_model* M, orc_mc_run* R, double* theta_out, double* logpi_out, double* delta_out) {
  const int64_t N = R->nparticles;
  const int ld = M->ld;
  pop_t cur = pop_alloc(N, ld), nxt = pop_alloc(N, ld);
  uint32_t* order = (uint32_t*)malloc((size_t)N * 4);
  double* sorted = (double*)malloc((size_t)N * sizeof(double));
  uint32_t* cnt = (uint32_t*)calloc((size_t)N, 4);
  int rc = orc_init(M, cur.theta, cur.logpi, cur.delta, 0, N);      /* mc:117-125 */
  const double gamma0 = 2.38 / sqrt(2.0 * (double)M->d), gsig = 1e-5;   /* mc:129-130 */
  int64_t nsims = 0;
  double complete = 1.0 - (double)orc_count_gt(cur.delta, N, R->eps_target) / (double)N;   /* mc:133 */
  for (int it = 0; rc == 0 && it < R->generations; ++it) {          /* mc:134 */
    double lo, hi;
    orc_extrema(cur.delta, N, &lo, &hi);                            /* mc:146 */
    double eps_pop = fmax(R->eps_target, lo + 0.0 * (hi - lo));     /* mc:147, alpha = 0 (mc:107) */
    /* the better particle of mc:23 by rank or by rejection: include/abcdez_spec.h, abz_mc_draws_by_rejection */
    const int reject = abz_mc_draws_by_rejection((uint64_t)orc_count_gt(cur.delta, N, R->eps_target), (uint64_t)N);
    if (hi > R->eps_target && !reject) orc_mc_rank_prepare(cur.delta, N, eps_pop, order, sorted, cnt);   /* only consulted when D_i > eps */
    int64_t nsim;
    orc_mc_swarm(M, reject ? NULL : order, reject ? NULL : cnt, N, cur.theta, cur.logpi, cur.delta, nxt.theta, nxt.logpi, nxt.delta,
                 eps_pop, R->eps_target, gamma0, gsig, 0, N, (uint32_t)it, &nsim);           /* mc:149 */
    pop_swap(&cur, &nxt);                                           /* mc:152-155 */
    nsims += nsim;
    complete = 1.0 - (double)orc_count_gt(cur.delta, N, R->eps_target) / (double)N;          /* mc:156 */
  }
  double lo, hi;
  orc_extrema(cur.delta, N, &lo, &hi);
  R->reached_eps = hi <= R->eps_target;                             /* mc:163 */
  R->nsims_total = nsims; R->complete = complete;
  memcpy(theta_out, cur.theta, (size_t)N * ld * sizeof(double));
  memcpy(logpi_out, cur.logpi, (size_t)N * sizeof(double));
  memcpy(delta_out, cur.delta, (size_t)N * sizeof(double));
  free(order); free(sorted); free(cnt);
  pop_free(cur); pop_free(nxt);
  return rc;
}
