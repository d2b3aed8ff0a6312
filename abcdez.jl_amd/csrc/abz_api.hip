/*
 * abz_api.hip -- the extern "C" surface of libabcdez_hip.so (include/abcdez_hip.h).
 * Argument checking, context life cycle and the host ends of the scalar-returning
 * calls; all device work is launched from the kernel files.
 */
#include <stddef.h>
#include <string.h>

#include "../../include/abcdez_hip.h"
#include <chrono>
#include "abz_ctx.h"

int abz_tree_sum_impl(abcdez_ctx*, const double*, int64_t, int, double*);
int abz_reweight_impl(abcdez_ctx*, const double*, double*, uint8_t*, int64_t, double, double, double*, double*, int64_t*);
int abz_stratified_impl(abcdez_ctx*, const double*, int64_t, uint32_t, uint32_t*);
int abz_select_impl(abcdez_ctx*, const double*, const uint8_t*, int64_t, int64_t, double*, double*, int64_t*);
int abz_extrema_impl(abcdez_ctx*, const double*, int64_t, double*, double*);
int abz_count_gt_impl(abcdez_ctx*, const double*, int64_t, double, int64_t*);
int abz_math_eval_impl(abcdez_ctx*, int, const double*, double*, double*, int64_t);
int abz_rank_prepare_impl(abcdez_ctx*, const double*, int64_t, double, double, uint32_t*, double*, uint32_t*, const unsigned long long*, int64_t, int64_t);
int abz_count_alive_impl(abcdez_ctx*, const uint8_t*, int64_t, int64_t*);
int abz_partition_impl(abcdez_ctx*, uint8_t*, int64_t, int64_t, int64_t, const uint32_t*, uint32_t*, double*, double*, double*, double*, double*, const unsigned long long*, double, bool wfill = false);
int abz_prologue_packed_impl(abcdez_ctx*, const double*, int64_t, int64_t, double*, uint8_t*, double, double, double, double, double, const uint32_t*, uint32_t*, double*, double*, double*, double*, double*, int64_t*, int32_t*);
int abz_launch_smc_swarm_packed(abcdez_ctx*, const uint32_t*, uint32_t*, uint32_t, uint32_t, uint32_t, double*, double*, double*, double*, uint8_t*, double, double, double, uint32_t, int, const unsigned long long*);
int abz_launch_group_check(abcdez_ctx*, int, unsigned long long, uint32_t, double, int);
int abz_launch_smc_replay_packed(abcdez_ctx*, const uint32_t*, uint32_t*, uint32_t, uint32_t, uint32_t, double*, double*, double*, const uint8_t*, double, double, uint32_t, const unsigned long long*);
int abz_launch_resample_gather_packed(abcdez_ctx*, const uint32_t*, uint32_t, uint32_t*, uint32_t*, double*, double*, const double*, const double*, double*, double*, double*, uint8_t*);
int abz_launch_packed_gather(abcdez_ctx*, const uint32_t*, uint32_t, const double*, const double*, double*);
int abz_draws_eval_impl(abcdez_ctx*, int, uint32_t, uint32_t, uint32_t, uint32_t, double, double, uint32_t*, uint32_t*, double*, double*);

static thread_local std::string g_err;
void abz_set_error(const std::string& msg) { g_err = msg; }

#define ABZ_REQUIRE(cond, msg)      \
  do {                              \
    if (!(cond)) {                  \
      abz_set_error(msg);           \
      return -1;                    \
    }                               \
  } while (0)

int abz_ws_reserve(abcdez_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return 0;
  /* growing is rare (first call per population size); it synchronises the stream */
  ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (ctx->ws) ABZ_HIP_CHECK(hipFree(ctx->ws));
  ctx->ws = nullptr; ctx->ws_bytes = 0;
  const size_t want = abz_align(bytes + bytes / 4, 1 << 20);
  ABZ_HIP_CHECK(hipMalloc(&ctx->ws, want));
  ctx->ws_bytes = want;
  return 0;
}

/* which models sweep in two launches (abz_kernels.h, smc_split_phase1_body): one lane per particle, rows of 4, 8 or 16 doubles, a simulator
 * that is the bulk of the sweep -- Lotka-Volterra, and user-supplied simulators (ABZ_USER_ONE_KERNEL=1 in the environment keeps those
 * on the one-kernel two-phase body: the A/B switch, and the choice for a simulator cheaper than a launch and 100 bytes of traffic) */
bool abz_sweep_in_two_launches(const abcdez_ctx* ctx) {
  if (ctx->L != 1) return false;
  if (ctx->h_model.sim_id == ABZ_SIM_LV && ctx->C == 4) return true;
  return ctx->h_model.sim_id == ABZ_SIM_USER && (ctx->C == 4 || ctx->C == 8 || ctx->C == 16) && !ctx->user_one_kernel;
}

/* the hand-over list of a two-launch sweep: [two counters | pad to 256 B | tp 8 ld B | wl 8 | kdi 8 | logu 8 | pos 4] per position */
int abz_lv_hand_reserve(abcdez_ctx* ctx, size_t positions) {
  if (positions <= ctx->lv_hand_cap) return 0;
  ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (ctx->lv_hand) ABZ_HIP_CHECK(hipFree(ctx->lv_hand));
  ctx->lv_hand = nullptr; ctx->lv_hand_cap = 0;
  const size_t cap = abz_align(positions + positions / 8, 1024);
  ABZ_HIP_CHECK(hipMalloc(&ctx->lv_hand, 256 + cap * ((size_t)ctx->h_model.ld * 8 + 32)));
  ABZ_HIP_CHECK(hipMemset(ctx->lv_hand, 0, 256));              /* both counters zero: the invariant the sweeps keep */
  ctx->lv_hand_cap = cap;
  return 0;
}

static bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

/* default lane-group shape: 8 components (64 contiguous bytes) per lane for the
 * component-parallel simulator -- measured best on MI355X at d = 32 (4 lanes
 * x 8 comps 0.557 ms, 8 x 4 0.570, 2 x 16 0.625, 16 x 2 0.721 per 2.7 M-update sweep);
 * the whole row in one thread otherwise                                            */
static void default_shape(const abz_model& m, int* L, int* C) {
  if (m.sim_id == ABZ_SIM_MVN && m.ld > 64) { *L = 8; *C = m.ld / 8; }           /* rows of 128 / 256 doubles: 16 / 32 components per lane */
  else if (m.sim_id == ABZ_SIM_MVN && m.ld > 8) { *C = 8; *L = m.ld / 8; }
  else if (m.sim_id == ABZ_SIM_USER && m.ld > 64) { *L = 8; *C = m.ld / 8; }     /* the cooperative form on rows of 128 / 256 doubles */
  else if (m.sim_id == ABZ_SIM_USER && m.ld > 16) { *C = 8; *L = m.ld / 8; }     /* the cooperative form of a user simulator */
  else { *L = 1; *C = m.ld; }
}

/* an abcdemc chain (abz_ctx.h) ends: its proved tail bound and hint were made for distances that are being rewritten */
static inline void abz_mc_chain_break(abcdez_ctx* ctx) {
  ctx->mc_chain += 1;
  ctx->mc_tail_bound = -1;
  ctx->mc_tail_hint = -1;
  ctx->mc_reject_known = false;
}
/* the population was written by something other than the next asynchronous abcdemc generation */
static inline void abz_population_written(abcdez_ctx* ctx) {
  ctx->ahead = abz_ahead{};
  abz_mc_chain_break(ctx);
}

extern "C" {

/* 400: Philox4x32-10 again, abz_model.mv, group abort; 401: abcdemc's better particle by rejection;
 * 500: abcdez_comm_* and abcdez_smc_sweeps_sharded (RCCL behind the ABI), timing mode 3; 600: abcdez_comm_init_host (a host-supplied
 * transport under the same sharded entry points), librccl opened lazily; 610: abcdez_smc_generation_packed, abz_model.ext.  Hosts refuse a library older than the
 * header they were written against (abcdez.jl_amd/_lib.py, julia/ABCdeZHIP.jl check_abi): a signature that grew an argument links
 * against an old binary without a diagnostic. */
int abcdez_version(void) { return 610; }
int abcdez_rng_rounds(void) { return ABZ_PHILOX_ROUNDS; }

/* sizeof / offsetof of the two structs that cross the boundary, so that a host that mirrors them by hand (the Julia
 * shim, the ctypes binding) can assert its layout instead of trusting it */
int abcdez_abi_layout(int32_t* out, int n) {
  const int32_t lay[] = {
      (int32_t)sizeof(abz_prior_dim), (int32_t)offsetof(abz_prior_dim, family), (int32_t)offsetof(abz_prior_dim, discrete),
      (int32_t)offsetof(abz_prior_dim, p0), (int32_t)offsetof(abz_prior_dim, p1), (int32_t)offsetof(abz_prior_dim, c0),
      (int32_t)offsetof(abz_prior_dim, c1), (int32_t)offsetof(abz_prior_dim, reserved),
      (int32_t)sizeof(abz_model), (int32_t)offsetof(abz_model, d), (int32_t)offsetof(abz_model, ld),
      (int32_t)offsetof(abz_model, sim_id), (int32_t)offsetof(abz_model, abck), (int32_t)offsetof(abz_model, seed),
      (int32_t)offsetof(abz_model, n_data), (int32_t)offsetof(abz_model, n_blob), (int32_t)offsetof(abz_model, sim_p),
      (int32_t)offsetof(abz_model, data), (int32_t)offsetof(abz_model, prior), (int32_t)offsetof(abz_model, mv),
      (int32_t)offsetof(abz_model, ext), (int32_t)offsetof(abz_model, n_ext), (int32_t)offsetof(abz_model, reserved0)};
  const int m = (int)(sizeof(lay) / sizeof(lay[0]));
  for (int k = 0; k < m && k < n; ++k) out[k] = lay[k];
  return m;
}
const char* abcdez_last_error(void) { return g_err.c_str(); }

/* Parameters of one prior descriptor: what the device's log-densities and samplers assume (the hosts check the same when the
 * distribution object is made -- abcdez_amd/priors.py, julia/ABCdeZHIP.jl -- but a descriptor can also arrive through the bare C ABI).
 * Inversion samplers start from exp(-lambda) / q^n, which must not underflow; the rejection samplers need an interval that holds at
 * least 1 % of the parent's mass.  Returns an empty string when everything is in order. */
static std::string prior_dim_problem(const abz_prior_dim& pd, const double* ext, int n_ext, bool nested) {
  const double p0 = pd.p0, p1 = pd.p1;
  auto pos = [](double v) { return v > 0.0 && v < 1.0e300; };
  auto fin = [](double v) { return v == v && v > -1.0e300 && v < 1.0e300; };
  switch (pd.family) {
    case ABZ_PRIOR_PAD: return "";
    case ABZ_PRIOR_NORMAL: return fin(p0) && pos(p1) ? "" : "Normal: need a finite mean and sigma > 0";
    case ABZ_PRIOR_UNIFORM: return fin(p0) && fin(p1) && p0 < p1 ? "" : "Uniform: need a < b";
    case ABZ_PRIOR_DUNIFORM: return fin(p0) && fin(p1) && p0 <= p1 && p0 == abz_rint(p0) && p1 == abz_rint(p1) ? "" : "DiscreteUniform: need integers a <= b";
    case ABZ_PRIOR_BETA: return pos(p0) && pos(p1) ? "" : "Beta: need alpha, beta > 0";
    case ABZ_PRIOR_NEGBIN: return pos(p0) && p1 > 0.0 && p1 <= 1.0 ? "" : "NegativeBinomial: need r > 0 and 0 < p <= 1";
    case ABZ_PRIOR_EXPONENTIAL: return pos(p0) ? "" : "Exponential: need scale > 0";
    case ABZ_PRIOR_GAMMA: case ABZ_PRIOR_INVGAMMA: case ABZ_PRIOR_WEIBULL: case ABZ_PRIOR_PARETO:
      return pos(p0) && pos(p1) ? "" : "Gamma / InverseGamma / Weibull / Pareto: need shape > 0 and scale > 0";
    case ABZ_PRIOR_LOGNORMAL: case ABZ_PRIOR_CAUCHY: case ABZ_PRIOR_LAPLACE: case ABZ_PRIOR_LOGISTIC:
      return fin(p0) && pos(p1) ? "" : "LogNormal / Cauchy / Laplace / Logistic: need a finite location and scale > 0";
    case ABZ_PRIOR_TDIST: return pos(p0) ? "" : "TDist: need nu > 0";
    case ABZ_PRIOR_POISSON: return p0 > 0.0 && p0 <= 700.0 ? "" : "Poisson: need 0 < lambda <= 700 (the sampler inverts from exp(-lambda))";
    case ABZ_PRIOR_BINOMIAL: {
      if (!(p0 >= 1.0 && p0 == abz_rint(p0) && p0 < 4.0e15 && p1 > 0.0 && p1 < 1.0)) return "Binomial: need an integer n >= 1 and 0 < p < 1";
      const double thin = p1 < 0.5 ? p1 : 1.0 - p1;
      return p0 * log1p(-thin) > -700.0 ? "" : "Binomial: n log(1 - min(p, 1 - p)) must stay above -700 (the sampler inverts from q^n)";
    }
    case ABZ_PRIOR_TRUNCNORMAL: {
      if (!(fin(p0) && pos(p1) && pd.c1 < pd.reserved)) return "truncated(Normal): need sigma > 0 and lo < hi";
      const double logmass = -log(p1) - 0.91893853320467274178 - pd.c0;        /* c0 = -log sigma - log(2 pi)/2 - log mass */
      return logmass >= log(0.01) - 1e-9 ? "" : "truncated(Normal): [lo, hi] must hold at least 1 % of the Normal's mass (rejection sampler)";
    }
    case ABZ_PRIOR_TRUNCATED: {
      if (nested) return "a wrapper family inside a wrapper";
      const double off = p0;
      if (!ext || !(off >= 0.0 && off == abz_rint(off) && off + ABZ_EXT_TRUNC <= (double)n_ext)) return "truncated(...): record outside the ext table";
      const double* rec = ext + (size_t)off;
      if (!(rec[0] < rec[1])) return "truncated(...): need lo < hi";
      if (!(rec[2] <= 1e-9 && rec[2] >= log(0.01) - 1e-9)) return "truncated(...): [lo, hi] must hold at least 1 % of the parent's mass (rejection sampler)";
      abz_prior_dim par;
      abz_ext_desc(rec + 3, &par);
      if (par.family <= ABZ_PRIOR_PAD || par.family > ABZ_PRIOR_LAST) return "truncated(...): unknown parent family";
      if ((par.discrete != 0) != (pd.discrete != 0)) return "truncated(...): the discrete flag must be the parent's";
      return prior_dim_problem(par, ext, n_ext, true);
    }
    case ABZ_PRIOR_AFFINE: {
      if (nested) return "a wrapper family inside a wrapper";
      const double off = p0;
      if (!ext || !(off >= 0.0 && off == abz_rint(off) && off + ABZ_EXT_AFFINE <= (double)n_ext)) return "mu + sigma * d: record outside the ext table";
      const double* rec = ext + (size_t)off;
      if (!(fin(rec[0]) && pos(rec[1]) && fabs(rec[2] * rec[1] - 1.0) < 1e-12 && fabs(rec[3] - log(rec[1])) < 1e-12))
        return "mu + sigma * d: need a finite mu, sigma > 0 and the record's 1 / sigma and log sigma to match";
      abz_prior_dim par;
      abz_ext_desc(rec + 4, &par);
      if (par.family <= ABZ_PRIOR_PAD || par.family > ABZ_PRIOR_LAST) return "mu + sigma * d: unknown parent family";
      if (par.discrete || pd.discrete) return "mu + sigma * d: the parent must be continuous";
      return prior_dim_problem(par, ext, n_ext, true);
    }
    case ABZ_PRIOR_MIXTURE: {
      if (nested) return "a wrapper family inside a wrapper";
      const double K = p0, off = p1;
      if (!(K >= 1.0 && K <= (double)ABZ_MAX_MIX && K == abz_rint(K))) return "MixtureModel: 1 .. 16 components";
      if (!ext || !(off >= 0.0 && off == abz_rint(off) && off + K * ABZ_EXT_MIXC <= (double)n_ext)) return "MixtureModel: records outside the ext table";
      const double* rec = ext + (size_t)off;
      double prev = 0.0;
      for (int j = 0; j < (int)K; ++j) {
        const double* r = rec + (size_t)j * ABZ_EXT_MIXC;
        if (!(r[0] <= 1e-12) || !(r[1] >= prev && r[1] <= 1.0 + 1e-9)) return "MixtureModel: weights must be positive and sum to 1";
        prev = r[1];
        abz_prior_dim c;
        abz_ext_desc(r + 2, &c);
        if (c.family <= ABZ_PRIOR_PAD || c.family > ABZ_PRIOR_LAST) return "MixtureModel: unknown component family";
        if ((c.discrete != 0) != (pd.discrete != 0)) return "MixtureModel: the components must be all continuous or all discrete";
        const std::string why = prior_dim_problem(c, ext, n_ext, true);
        if (!why.empty()) return why;
      }
      return fabs(prev - 1.0) < 1e-9 ? "" : "MixtureModel: weights must sum to 1";
    }
    default: return "unknown prior family";
  }
}

static int ctx_create_common(const abz_model* model, const char* user_source, int device, abcdez_ctx** out) {
  ABZ_REQUIRE(model && out, "ctx_create: null argument");
  ABZ_REQUIRE((model->sim_id == ABZ_SIM_USER) == (user_source != nullptr),
              "ctx_create: ABZ_SIM_USER needs abcdez_ctx_create_user with a source, built-in simulators abcdez_ctx_create");
  ABZ_REQUIRE(model->d >= 1 && model->d <= ABZ_MAX_D, "ctx_create: d out of range");
  ABZ_REQUIRE(is_pow2(model->ld) && model->ld >= model->d && model->ld < 2 * model->d + (model->d == 1),
              "ctx_create: ld must be the smallest power of two >= d");
  ABZ_REQUIRE(model->abck >= 0 && model->abck <= 3, "ctx_create: unknown ABC kernel id");
  ABZ_REQUIRE(model->sim_id >= 0 && model->sim_id <= ABZ_SIM_USER, "ctx_create: unknown simulator id");
  ABZ_REQUIRE(model->n_data >= 0 && (model->n_data == 0 || model->data), "ctx_create: data pointer missing");
  for (int k = 0; k < model->ld; ++k) {
    const int fam = model->prior[k].family;
    ABZ_REQUIRE(fam >= ABZ_PRIOR_PAD && fam <= ABZ_PRIOR_LAST, "ctx_create: unknown prior family");
    ABZ_REQUIRE((k < model->d) == (fam != ABZ_PRIOR_PAD), "ctx_create: prior descriptor / d mismatch");
  }
  ABZ_REQUIRE(model->n_ext >= 0 && model->n_ext <= (1 << 16) && (model->n_ext == 0) == (model->ext == nullptr),
              "ctx_create: ext / n_ext mismatch (the records of truncated(...) / MixtureModel priors)");
  for (int k = 0; k < model->d; ++k) {
    const std::string why = prior_dim_problem(model->prior[k], model->ext, model->n_ext, false);
    if (!why.empty()) { abz_set_error("ctx_create: prior factor " + std::to_string(k + 1) + ": " + why); return -1; }
  }
  ABZ_REQUIRE(!(model->mv && model->n_ext), "ctx_create: a correlated Normal prior has no wrapped factors");
  switch (model->sim_id) {
    case ABZ_SIM_NORMAL1D: ABZ_REQUIRE(model->d == 1 && model->n_data >= 1, "normal1d: needs d = 1 and one datum"); break;
    case ABZ_SIM_MVN: ABZ_REQUIRE(model->n_data >= model->d, "mvn: needs d data values"); break;
    case ABZ_SIM_DIRAC: case ABZ_SIM_MIXTURE: ABZ_REQUIRE(model->d == 1, "simulator needs d = 1"); break;
    case ABZ_SIM_QUAD2D: case ABZ_SIM_NORMDU: ABZ_REQUIRE(model->d == 2, "simulator needs d = 2"); break;
    case ABZ_SIM_USER:     /* up to 16 parameters: the whole row in one thread (abz_user_dist); beyond: 8 per lane (abz_user_dist_lanes) */
      ABZ_REQUIRE(model->n_blob == 0 || model->d <= 16, "user simulator: blobs need the whole row in one thread (d <= 16)");
      break;
    case ABZ_SIM_SOCKS:
      ABZ_REQUIRE(model->d == 2 && model->sim_p[2] >= 1.0 && model->sim_p[2] <= 16.0, "socks: needs d = 2 and 1..16 picked socks");
      break;
    case ABZ_SIM_WIENER: ABZ_REQUIRE(model->d == 2 && model->n_data >= 1, "wiener: needs d = 2 and data"); break;
    case ABZ_SIM_LV:
      ABZ_REQUIRE(model->d == 4 && model->n_data >= 2 && model->n_data % 2 == 0, "lv: needs d = 4 and (x,y) data");
      ABZ_REQUIRE(model->sim_p[3] >= 1.0 && model->sim_p[3] <= 1e6, "lv: steps per observation out of range");
      break;
    default: break;
  }
  /* blobs: off (0), or exactly the simulated data of the built-in simulator / the size the user source declares */
  ABZ_REQUIRE(model->n_blob >= 0 && model->n_blob <= ABZ_MAX_BLOB, "model: n_blob must be in 0..64");
  if (model->n_blob > 0 && model->sim_id != ABZ_SIM_USER)
    ABZ_REQUIRE(model->n_blob == abz_sim_blob_size(model->sim_id, model->d, model->n_data),
                "model: n_blob does not match the simulated data of this simulator");
  int ndev = 0;
  ABZ_HIP_CHECK(hipGetDeviceCount(&ndev));
  ABZ_REQUIRE(ndev > 0, "ctx_create: no HIP device visible");
  ABZ_REQUIRE(device >= 0 && device < ndev, "ctx_create: device index out of range");
  ABZ_HIP_CHECK(hipSetDevice(device));
  /* from here on every failure releases what has been allocated so far (abcdez_ctx_destroy copes with a half-built context) */
#define ABZ_CTX_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      abz_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                  \
      abcdez_ctx_destroy(ctx);                                                           \
      return -2;                                                                         \
    }                                                                                    \
  } while (0)
  abcdez_ctx* ctx = new abcdez_ctx();
  ctx->device = device;
  if (const char* g = getenv("ABZ_GRAPHS")) ctx->graphs_on = !(g[0] == '0' && g[1] == 0) && g[0] != 0;
  if (const char* g = getenv("ABZ_SERPENTINE")) ctx->serpentine = !(g[0] == '0' && g[1] == 0);
  if (const char* g = getenv("ABZ_USER_ONE_KERNEL")) ctx->user_one_kernel = !(g[0] == '0' && g[1] == 0) && g[0] != 0;
  ctx->h_model = *model;
  default_shape(*model, &ctx->L, &ctx->C);
  /* every dimension of the row a continuous Normal (d == ld, no padding): the sweeps run the two-instruction log-density
   * instead of the family dispatch and the MVN simulator drops its padding selects (same arithmetic) */
  ctx->prior_plain = model->d == model->ld;
  for (int k = 0; k < ABZ_MAX_D; ++k) {
    abz_prior_dim& pd = ctx->h_model.prior[k];
    if (k < model->d) {
      if (pd.family != ABZ_PRIOR_NORMAL || pd.discrete) ctx->prior_plain = false;
    } else if (k < model->ld) {
      if (pd.family != ABZ_PRIOR_PAD) ctx->prior_plain = false;
      else { pd.discrete = 0; pd.p0 = pd.p1 = pd.c0 = pd.c1 = 0.0; }
    }
  }
  /* The small device objects every kernel reads -- scalars and counter slots, tickets, the model, the sampler tables, the data vector,
   * the maps of a correlated prior -- are ONE allocation: one address translation instead of six.  (Counters, round 5: after the
   * prologue's kernels the first sweep of a generation refetched 5.7 translations per CU, 1,450 UTCL1 misses against 37 in its
   * siblings, and paid 10-15 us for them: profiles/HISTORY.md.) */
  {
    const size_t b_scal = abz_align((size_t)ABZ_S_N * 8, 256), b_sync = abz_align((size_t)ABZ_SYNC_N * 4, 256);
    const size_t b_model = abz_align(sizeof(abz_model), 256), b_tab = abz_align(sizeof(abz_tables), 256);
    const size_t b_data = abz_align((size_t)(model->n_data > 0 ? model->n_data : 0) * 8, 256);
    const size_t b_mv = model->mv ? abz_align(ABZ_MV_DOUBLES(model->ld) * 8, 256) : 0;
    const size_t b_ext = model->n_ext > 0 ? abz_align((size_t)model->n_ext * 8, 256) : 0;
    char* base = nullptr;
    ABZ_CTX_CHECK(hipMalloc((void**)&base, b_scal + b_sync + b_model + b_tab + b_data + b_mv + b_ext + 256));
    ctx->d_block = base;
    ctx->d_scal = (unsigned long long*)base; base += b_scal;
    ctx->d_sync = (unsigned int*)base; base += b_sync;
    ctx->d_model = (abz_model*)base; base += b_model;
    ctx->d_tables = (abz_tables*)base; base += b_tab;
    if (model->n_data > 0) { ctx->d_data = (double*)base; base += b_data; }
    if (model->mv) { ctx->d_mv = (double*)base; base += b_mv; }
    if (model->n_ext > 0) { ctx->d_ext = (double*)base; base += b_ext; }
  }
  if (model->n_ext > 0) ABZ_CTX_CHECK(hipMemcpy(ctx->d_ext, model->ext, (size_t)model->n_ext * 8, hipMemcpyHostToDevice));
  ctx->h_model.ext = ctx->d_ext;
  if (model->n_data > 0) ABZ_CTX_CHECK(hipMemcpy(ctx->d_data, model->data, (size_t)model->n_data * 8, hipMemcpyHostToDevice));
  ctx->h_model.data = ctx->d_data;
  if (model->mv) {                   /* correlated Normal prior: [mu | W | L] travels to the device; no plain-Normal shortcut */
    ABZ_CTX_CHECK(hipMemcpy(ctx->d_mv, model->mv, ABZ_MV_DOUBLES(model->ld) * 8, hipMemcpyHostToDevice));
    ctx->prior_plain = false;
  }
  ctx->h_model.mv = ctx->d_mv;
  ABZ_CTX_CHECK(hipMemcpy(ctx->d_model, &ctx->h_model, sizeof(abz_model), hipMemcpyHostToDevice));
  ABZ_CTX_CHECK(hipMemcpy(ctx->d_tables, &abz_tables_host, sizeof(abz_tables), hipMemcpyHostToDevice));
  {
    int ncu = 0;
    ABZ_CTX_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    ctx->n_cu = ncu > 0 ? ncu : 1;
  }
  ctx->hot.tables = ctx->d_tables;
  ctx->hot.mv = ctx->d_mv;
  ctx->hot.ext = ctx->d_ext;
  ctx->hot.seed = model->seed;
  ctx->hot.prior = (const abz_prior_dim*)((const char*)ctx->d_model + offsetof(abz_model, prior));
  ctx->hot.data = ctx->d_data;
  for (int q = 0; q < 8; ++q) ctx->hot.sim_p[q] = model->sim_p[q];
  ctx->hot.sim_i[0] = model->sim_id == ABZ_SIM_LV ? (int32_t)model->sim_p[3] : 0;
  ctx->hot.sim_i[1] = 0;
  ctx->hot.d = model->d; ctx->hot.abck = model->abck; ctx->hot.n_data = model->n_data; ctx->hot.n_blob = model->n_blob;
  ABZ_CTX_CHECK(hipMemset(ctx->d_scal, 0, ABZ_S_N * 8));
  ABZ_CTX_CHECK(hipMemset(ctx->d_sync, 0, ABZ_SYNC_N * 4));
  {   /* both min / max banks start empty: (min key, max key) = (~0, 0) */
    unsigned long long mm[2 * ABZ_MMSLOTS * 2];
    for (int k = 0; k < 2 * ABZ_MMSLOTS * 2; ++k) mm[k] = (k & 1) ? 0ull : ~0ull;
    ABZ_CTX_CHECK(hipMemcpy(ctx->d_scal + ABZ_S_MM0, mm, sizeof(mm), hipMemcpyHostToDevice));
  }
  ABZ_CTX_CHECK(hipHostMalloc((void**)&ctx->h_scal, (ABZ_S_N + 8) * 8, hipHostMallocMapped | hipHostMallocCoherent));
  memset(ctx->h_scal, 0, (ABZ_S_N + 8) * 8);
  ABZ_CTX_CHECK(hipHostGetDevicePointer((void**)&ctx->h_scal_dev, ctx->h_scal, 0));
  /* kernels compiled for this model at run time: a user-supplied simulator, or prior factors of the wrapper families (truncated(...),
   * MixtureModel), whose log-densities the statically compiled sweeps do not carry (include/abcdez_spec.h, ABZ_PRIOR_WRAP) */
  if (user_source || model->n_ext > 0) {
    const int rc = abz_jit_build(ctx, user_source);
    if (rc) { abcdez_ctx_destroy(ctx); return rc; }
  }
#undef ABZ_CTX_CHECK
  *out = ctx;
  return 0;
}

int abcdez_ctx_create(const abz_model* model, int device, abcdez_ctx** out) {
  return ctx_create_common(model, nullptr, device, out);
}

int abcdez_ctx_create_user(const abz_model* model, const char* user_source, int device, abcdez_ctx** out) {
  ABZ_REQUIRE(user_source, "ctx_create_user: null source");
  return ctx_create_common(model, user_source, device, out);
}

int abcdez_ctx_destroy(abcdez_ctx* ctx) {
  if (!ctx) return 0;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm_kind != ABZ_COMM_NONE) (void)abcdez_comm_destroy(ctx);
  for (abz_mc_graph& g : ctx->mc_graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
  ctx->mc_graphs.clear();
  abz_jit_destroy(ctx);
  for (hipEvent_t e : ctx->ev) if (e) (void)hipEventDestroy(e);
  if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->lv_hand) (void)hipFree(ctx->lv_hand);
  if (ctx->sel_hist) (void)hipFree(ctx->sel_hist);
  if (ctx->h_scal) (void)hipHostFree(ctx->h_scal);
  if (ctx->d_block) (void)hipFree(ctx->d_block);      /* d_scal, d_sync, d_model, d_tables, d_data, d_mv */
  delete ctx;
  return 0;
}

int abcdez_ctx_reserve(abcdez_ctx* ctx, int64_t N) {
  ABZ_REQUIRE(ctx, "ctx_reserve: null context");
  ABZ_REQUIRE(N >= 1 && N <= 0x7FFFFFFFll, "ctx_reserve: N out of range");
  /* the largest user: the resampling (8 N of cumulative weights + tile sums) / the rank pass of abcdemc (12 N + its table) */
  int rc = abz_ws_reserve(ctx, (size_t)N * 16 + ((size_t)8 << 20));
  if (rc) return rc;
  if (abz_sweep_in_two_launches(ctx) && (rc = abz_lv_hand_reserve(ctx, (size_t)N))) return rc;        /* no allocation inside the loop */
  /* The resampling's kernels run for the first time ~14 generations into a run; HIP loads a kernel's code on its first
   * launch (about 0.2 ms each), which would land in the middle of the loop: run them once here on a 64-particle dummy. */
  const int64_t n = 64;
  const size_t ld = (size_t)ctx->h_model.ld;
  char* scratch = nullptr;
  const size_t rows = n * ld * 8;
  ABZ_HIP_CHECK(hipMalloc((void**)&scratch, 2 * rows + 6 * n * 8 + 4 * 64));
  ABZ_HIP_CHECK(hipMemsetAsync(scratch, 0, 2 * rows + 6 * n * 8 + 4 * 64, ctx->stream));
  double* s0 = (double*)scratch; double* s1 = (double*)(scratch + rows);
  double* f = (double*)(scratch + 2 * rows);             /* wns | logpi | delta | nlogpi | ndelta | (inds, alive) */
  uint32_t* inds = (uint32_t*)(f + 5 * n); uint8_t* alive = (uint8_t*)(inds + n);
  uint32_t* bits = (uint32_t*)(scratch + 2 * rows + 6 * n * 8);
  uint64_t* st_cur = ctx->stamp_cur; uint64_t* st_nxt = ctx->stamp_nxt;
  ctx->stamp_cur = ctx->stamp_nxt = nullptr;
  rc = abz_stratified_impl(ctx, f, n, 0u, inds);         /* all-zero weights: every stratum picks index 0 */
  if (!rc) rc = abz_launch_resample_gather_packed(ctx, inds, (uint32_t)n, bits, bits + 8, s0, s1, f + n, f + 2 * n, f + 3 * n, f + 4 * n, f, alive);
  ctx->stamp_cur = st_cur; ctx->stamp_nxt = st_nxt;
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(scratch);
  return rc;
}

int abcdez_ctx_set_stream(abcdez_ctx* ctx, void* hip_stream) {
  ABZ_REQUIRE(ctx, "set_stream: null context");
  if (ctx->stream != (hipStream_t)hip_stream) {
    ctx->ahead = abz_ahead{};      /* a select enqueued ahead sits on the old stream */
    /* asynchronous abcdemc generations take their ring slot, ticket and RNG epoch from a counter on the device: work of the same
     * chain on two streams could run two snapshots at once.  Let the old stream finish what it holds and end the chain. */
    if (ctx->mc_issued > ctx->mc_waited) ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    abz_mc_chain_break(ctx);
  }
  ctx->stream = (hipStream_t)hip_stream;
  return 0;
}

/* Replay abcdemc generations as HIP graphs (default off: measured slower, abz_ctx.h).  Results do not depend on it. */
int abcdez_ctx_set_graphs(abcdez_ctx* ctx, int on) {
  ABZ_REQUIRE(ctx, "set_graphs: null context");
  ctx->graphs_on = on != 0;
  return 0;
}
/* Diagnostics: asynchronous abcdemc generations replayed from a graph / graphs captured / generations enqueued launch by launch. */
int abcdez_graph_stats(abcdez_ctx* ctx, int64_t* replays, int64_t* captures, int64_t* direct) {
  ABZ_REQUIRE(ctx && replays && captures && direct, "graph_stats: null argument");
  *replays = ctx->n_graph_replays; *captures = ctx->n_graph_captures; *direct = ctx->n_graph_direct;
  return 0;
}

/* "The alive particles' weights are uniform, Wns = 1 / n_alive" -- what a host asserts after it wrote them that way (smc:266-270).
 * With an indicator kernel the prologue then uses the closed forms of the reweight (abz_population.hip, ind_reweight_kernel).
 * The library keeps the flag itself afterwards: the indicator fast path keeps it, a resampling sets it (smc:102), a general
 * reweight clears it. */
int abcdez_ctx_set_uniform_weights(abcdez_ctx* ctx, int on) {
  ABZ_REQUIRE(ctx, "set_uniform_weights: null context");
  ctx->w_uniform = on != 0;
  return 0;
}
int abcdez_ctx_get_uniform_weights(abcdez_ctx* ctx, int32_t* on, int64_t* fast_prologues) {
  ABZ_REQUIRE(ctx && on, "get_uniform_weights: null argument");
  *on = ctx->w_uniform ? 1 : 0;
  if (fast_prologues) *fast_prologues = ctx->n_reweight_fast;
  return 0;
}

int abcdez_ctx_set_lanes(abcdez_ctx* ctx, int lanes) {
  ABZ_REQUIRE(ctx, "set_lanes: null context");
  if (lanes <= 0) { default_shape(ctx->h_model, &ctx->L, &ctx->C); return 0; }
  const int ld = ctx->h_model.ld;
  ABZ_REQUIRE(is_pow2(lanes) && lanes <= 16 && ld % lanes == 0, "set_lanes: lanes must be a power of two <= 16 dividing ld");
  ABZ_REQUIRE(!ctx->user_module || lanes == ctx->L,
              "set_lanes: the kernels of this model (a user simulator, or truncated(...) / MixtureModel priors) were compiled for their "
              "lane-group shape when the context was created");
  ABZ_REQUIRE(lanes == 1 || ctx->h_model.sim_id == ABZ_SIM_MVN || ctx->h_model.sim_id == ABZ_SIM_USER,
              "set_lanes: this simulator needs the whole row in one thread");
  const int C = ld / lanes;
  ABZ_REQUIRE(lanes == 1 || C >= 2, "set_lanes: at least two components per lane");
  ctx->L = lanes; ctx->C = C;
  return 0;
}

int abcdez_ctx_get_layout(abcdez_ctx* ctx, int32_t* ld, int32_t* lanes, int32_t* comps) {
  ABZ_REQUIRE(ctx, "get_layout: null context");
  if (ld) *ld = ctx->h_model.ld;
  if (lanes) *lanes = ctx->L;
  if (comps) *comps = ctx->C;
  return 0;
}

int abcdez_sync(abcdez_ctx* ctx) {
  ABZ_REQUIRE(ctx, "sync: null context");
  ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int abcdez_dev_alloc(size_t bytes, void** out) {
  ABZ_REQUIRE(out, "dev_alloc: null argument");
  ABZ_HIP_CHECK(hipMalloc(out, bytes ? bytes : 8));
  return 0;
}
int abcdez_dev_free(void* ptr) {
  if (ptr) ABZ_HIP_CHECK(hipFree(ptr));
  return 0;
}
int abcdez_memcpy_h2d(abcdez_ctx* ctx, void* dst, const void* src, size_t bytes) {
  ABZ_REQUIRE(ctx && dst && src, "memcpy_h2d: null argument");
  /* the host writes device memory of its own choice: what the library remembers about distances or flags -- a select enqueued
   * ahead, the window of the select, an abcdemc chain, a cached count -- is dropped iff the copy overlaps the arrays it describes
   * (uploads of weights, bitmap words, rows leave it alone; abcdez_smc_select_discard says "written" without a copy) */
  {
    const char* lo = (const char*)dst; const char* hi = lo + bytes;
    auto hits = [&](const void* a, size_t n) { return a && n && lo < (const char*)a + n && (const char*)a < hi; };
    const bool mc_hit = hits(ctx->mc_last_out, (size_t)ctx->mc_last_N * 8) || hits(ctx->mc_count_seen.delta, (size_t)ctx->mc_count_seen.N * 8);
    if (mc_hit || hits(ctx->ahead.delta, (size_t)ctx->ahead.N * 8) || hits(ctx->ahead.alive, (size_t)ctx->ahead.N)) abz_population_written(ctx);
    /* the extrema a sweep left on the device and the window made from them describe the overwritten distances too */
    if (mc_hit) { ctx->mc_window_ready = false; ctx->mc_have_bank = false; ctx->mc_count_seen.count = -1; }
  }
  ABZ_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}
int abcdez_host_alloc(size_t bytes, void** out) {
  ABZ_REQUIRE(out, "host_alloc: null argument");
  ABZ_HIP_CHECK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return 0;
}
int abcdez_host_free(void* ptr) {
  if (ptr) ABZ_HIP_CHECK(hipHostFree(ptr));
  return 0;
}
int abcdez_memcpy_d2h_async(abcdez_ctx* ctx, void* dst, const void* src, size_t bytes) {
  ABZ_REQUIRE(ctx && dst && src, "memcpy_d2h_async: null argument");
  ABZ_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return 0;
}
int abcdez_memcpy_d2h(abcdez_ctx* ctx, void* dst, const void* src, size_t bytes) {
  ABZ_REQUIRE(ctx && dst && src, "memcpy_d2h: null argument");
  ABZ_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}

static inline double f64_from_order_key_host(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  double x; memcpy(&x, &u, 8); return x;
}
/* consumes up to `n` completed event pairs of the timing FIFO; a pair counts only if the sweep it brackets did work: its index
 * inside its group of sweeps is below `done` (a group may end early -- the launches enqueued behind a test of smc:352 that
 * held return at once and are not launches of the roofline figure); done < 0 = every pair counts */
static int timing_consume(abcdez_ctx* ctx, long long n, long long done) {
  for (long long k = 0; k < n && ctx->ev_head < ctx->ev_tail; ++k, ++ctx->ev_head) {
    const int slot = (int)(ctx->ev_head % ABZ_GROUP_MAX);
    if (done >= 0 && !ctx->ev_group[slot] && ctx->ev_sweep[slot] >= done) continue;
    float ms = 0.f;
    ABZ_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev[2 * slot], ctx->ev[2 * slot + 1]));
    /* a pair around a whole group of sweeps (mode 3) stands for the `done` launches that did work, each of ev_units updates; the
     * one-block checks between them are inside the pair, so milliseconds / launches is an UPPER bound of the sweep's duration */
    const long long nl = ctx->ev_group[slot] ? (done > 0 ? done : ctx->ev_group[slot]) : 1;
    ctx->swarm_ms += (double)ms;
    ctx->swarm_launches += nl;
    ctx->swarm_units += ctx->ev_units[slot] * nl;
  }
  return 0;
}
/* takes generation `t` of the asynchronous abcdemc ring (its snapshot must be complete) into the counter baseline */
static void ring_fold(abcdez_ctx* ctx, long long t) {
  const int slot = (int)(t % ABZ_MC_RING);
  if (ctx->ring_folded[slot]) return;
  const volatile unsigned long long* snap = ctx->h_ring + (size_t)slot * ABZ_RING_WORDS;
  const unsigned long long tg = snap[0], ts = snap[1];
  if (snap[7] != 0ull) {     /* a sharded generation: its counts over all ranks as they are; this GPU's slot totals keep the baselines */
    ctx->ring_res[slot][0] = (long long)ts;
    ctx->ring_res[slot][1] = (long long)tg;
    ctx->cnt_prev[ABZ_C_MCSIM] = snap[9];
    ctx->cnt_prev[ABZ_C_MCGT] = snap[8];
  } else {
    ctx->ring_res[slot][0] = (long long)(ts - ctx->cnt_prev[ABZ_C_MCSIM]);
    ctx->ring_res[slot][1] = (long long)(tg - ctx->cnt_prev[ABZ_C_MCGT]);
    ctx->cnt_prev[ABZ_C_MCSIM] = ts;
    ctx->cnt_prev[ABZ_C_MCGT] = tg;
  }
  /* at least 1 / 16 of the particles at or below eps_target after this generation: every later generation of the chain draws by
   * rejection (abz_ctx.h, mc_reject_known) */
  if (ctx->ring_chain[slot] == ctx->mc_chain && ctx->ring_res[slot][1] >= 0 &&
      abz_mc_draws_by_rejection((uint64_t)ctx->ring_res[slot][1], (uint64_t)ctx->mc_last_N))
    ctx->mc_reject_known = true;
  if (snap[5] != ~0ull && ctx->ring_chain[slot] == ctx->mc_chain) {
    ctx->mc_tail_hint = (long long)snap[5];        /* the next rank pass sizes its long-tail launches from this */
    double eps_pop;
    const unsigned long long e = snap[4];
    memcpy(&eps_pop, &e, 8);
    if (eps_pop == ctx->ring_eps_target[slot]) ctx->mc_tail_bound = (long long)snap[5];   /* abz_ctx.h: from here on the tail only shrinks */
  }
  ctx->ring_folded[slot] = true;
}
/* ran_limit: of the sweeps timed since the last read-back only those with an index below ran_limit inside their group did
 * work; < 0 = all of them */
/* second half of a counter read-back (the scalars are in h_scal): ring generations first, baseline, timing events */
static int read_counters_finish(abcdez_ctx* ctx, int ran_limit) {
  /* generations still in the ring (complete: everything before the publish kernel has run): their counters come first,
   * their results stay redeemable */
  for (long long t = ctx->mc_waited; t < ctx->mc_issued; ++t) ring_fold(ctx, t);
  abz_fold_counters(ctx);
  const long long n_ev = ctx->ev_tail - ctx->ev_head;
  return timing_consume(ctx, n_ev, ran_limit);
}
static int read_counters(abcdez_ctx* ctx, int ran_limit = -1) {
  if (int rc = abz_publish(ctx, ABZ_S_N)) return rc;
  return read_counters_finish(ctx, ran_limit);
}

int abcdez_ctx_set_timing(abcdez_ctx* ctx, int on) {
  ABZ_REQUIRE(ctx, "set_timing: null context");
  if (on && !ctx->ev[0])      /* timing only: no system-scope fence (cache write-back) when an event is recorded between two sweeps */
    for (hipEvent_t& e : ctx->ev) ABZ_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
  const int mode = on & 0xFF, stride = on >> 8;
  ABZ_REQUIRE(mode <= 3 && stride >= 0, "set_timing: on = mode (0, 1, 2, 3) + 256 * stride");
  ctx->timing = mode != 0;
  ctx->timing_group = mode == 3;       /* 3: one pair around ALL the sweeps of a grouped call (the launches run back to back inside it) */
  ctx->timing_first_only = mode == 2;  /* 2: of a group of sweeps only one is bracketed (an event pair costs ~9 us of queue time) */
  ctx->timing_stride = stride > 1 ? stride : 1; ctx->timing_seq = 0; ctx->timing_rot = 0;
  ctx->swarm_ms = 0.0; ctx->swarm_launches = 0; ctx->swarm_units = 0; ctx->ev_head = ctx->ev_tail;
  return 0;
}

int abcdez_ctx_get_timing(abcdez_ctx* ctx, double* swarm_ms, int64_t* launches, int64_t* units) {
  ABZ_REQUIRE(ctx && swarm_ms && launches && units, "get_timing: null argument");
  *swarm_ms = ctx->swarm_ms; *launches = ctx->swarm_launches; *units = ctx->swarm_units;
  return 0;
}

#define ABZ_PACKED_ALIGN 64      /* sub-ranges of the packed prefix own whole wave-rounds of the replay and whole bitmap words */

/* ---- blobs (second return value of dist!): stamps carried with the distances, data rebuilt on demand ---- */
int abcdez_ctx_set_stamps(abcdez_ctx* ctx, uint64_t* stamp_cur, uint64_t* stamp_nxt) {
  ABZ_REQUIRE(ctx, "set_stamps: null context");
  ABZ_REQUIRE((stamp_cur == nullptr) == (stamp_nxt == nullptr), "set_stamps: pass both arrays or neither");
  ABZ_REQUIRE(stamp_cur == nullptr || stamp_cur != stamp_nxt, "set_stamps: the two arrays must differ");
  ABZ_REQUIRE(stamp_cur == nullptr || ctx->h_model.n_blob > 0, "set_stamps: the model was created with n_blob = 0");
  /* hosts rebind the same arrays before every call: only a real change invalidates a select enqueued ahead (a SWAP of the two
   * arrays follows a resample, which discards the select itself) */
  if (!((ctx->stamp_cur == stamp_cur && ctx->stamp_nxt == stamp_nxt) || (ctx->stamp_cur == stamp_nxt && ctx->stamp_nxt == stamp_cur)))
    ctx->ahead = abz_ahead{};
  ctx->stamp_cur = stamp_cur; ctx->stamp_nxt = stamp_nxt;
  return 0;
}

int abcdez_blob_width(abcdez_ctx* ctx, int32_t* width) {
  ABZ_REQUIRE(ctx && width, "blob_width: null argument");
  *width = ctx->h_model.sim_id == ABZ_SIM_MVN ? ctx->h_model.ld : ctx->h_model.n_blob;
  return 0;
}

int abcdez_blob_eval(abcdez_ctx* ctx, const double* theta, const uint64_t* stamp, int64_t N, double* blob,
                     double* delta_out) {
  ABZ_REQUIRE(ctx && theta && stamp && blob && delta_out, "blob_eval: null argument");
  ABZ_REQUIRE(ctx->h_model.n_blob > 0, "blob_eval: the model was created with n_blob = 0");
  ABZ_REQUIRE(N >= 0 && N <= ABZ_MAX_N, "blob_eval: N out of range");
  ABZ_REQUIRE_LANES(ctx, N, "blob_eval");
  const uint32_t nbw = (uint32_t)(ctx->h_model.sim_id == ABZ_SIM_MVN ? ctx->h_model.ld : ctx->h_model.n_blob);
  abz_population_written(ctx);         /* delta_out may be a distance array the library has state about */
  return abz_launch_blob_eval(ctx, theta, stamp, N, blob, delta_out, nbw);
}

int abcdez_init(abcdez_ctx* ctx, double* theta, double* logpi, double* delta, int64_t i0, int64_t n) {
  ABZ_REQUIRE(ctx && theta && logpi && delta, "init: null argument");
  abz_population_written(ctx);          /* select enqueued ahead, abcdemc chain: they no longer describe the population */
  ABZ_REQUIRE(i0 >= 0 && n >= 0 && i0 + n <= ABZ_MAX_N, "init: range out of bounds");
  ABZ_REQUIRE_LANES(ctx, n, "init");
  ABZ_HIP_CHECK(hipMemsetAsync(ctx->d_scal + ABZ_S_INITBAD, 0, 8, ctx->stream));
  int rc = abz_launch_init(ctx, theta, logpi, delta, i0, n);
  if (rc) return rc;
  rc = read_counters(ctx);
  if (rc) return rc;
  ABZ_REQUIRE(ctx->h_scal[ABZ_S_INITBAD] == 0, "init: a particle found no finite (log-prior, distance) within the retry limit");
  return 0;
}

/* ---- packed population: see the comment at SmcPackedArgs in abz_kernels.h ---- */
int abcdez_smc_partition(abcdez_ctx* ctx, uint8_t* alive, int64_t N, int64_t n_prev, int64_t n_new, const uint32_t* bits,
                         uint32_t* bits_other, double* slot0, double* slot1, double* logpi, double* delta, double* wns) {
  ABZ_REQUIRE(ctx && alive && bits && bits_other && slot0 && slot1 && logpi && delta && wns, "smc_partition: null argument");
  abz_population_written(ctx);          /* select enqueued ahead, abcdemc chain: they no longer describe the population */
  ABZ_REQUIRE(N >= 1 && N <= ABZ_MAX_N && 0 <= n_new && n_new <= n_prev && n_prev <= N, "smc_partition: need 0 <= n_new <= n_prev <= N");
  ABZ_REQUIRE_LANES(ctx, n_prev, "smc_partition");
  ABZ_REQUIRE(bits != bits_other && slot0 != slot1, "smc_partition: the two bit arrays / slots must differ");
  return abz_partition_impl(ctx, alive, N, n_prev, n_new, bits, bits_other, slot0, slot1, logpi, delta, wns, nullptr, 0.0);
}

int abcdez_smc_prologue_packed(abcdez_ctx* ctx, double* delta, double* wns, uint8_t* alive, int64_t N, int64_t n_prev,
                               double alpha, double eps_prev, double eps_target, double eps_k_old, double ess_min,
                               const uint32_t* bits, uint32_t* bits_other, double* slot0, double* slot1, double* logpi,
                               double* eps, double* q, double* wnorm, double* ess, int64_t* n_alive, int32_t* partitioned,
                               double* dmin, double* dmax) {
  ABZ_REQUIRE(ctx && delta && wns && alive && bits && bits_other && slot0 && slot1 && logpi && eps && wnorm && ess && n_alive &&
              partitioned, "smc_prologue_packed: null argument");
  ABZ_REQUIRE(N >= 1 && N <= ABZ_MAX_N && 1 <= n_prev && n_prev <= N, "smc_prologue_packed: need 1 <= n_prev <= N");
  ABZ_REQUIRE_LANES(ctx, n_prev, "smc_prologue_packed");
  ABZ_REQUIRE(alpha >= 0.0 && alpha <= 1.0, "smc_prologue_packed: alpha must be in [0, 1]");
  ABZ_REQUIRE(eps_k_old >= 0.0 && eps_target >= 0.0, "Expected ϵ ≥ 0.0");   /* types.jl:30 */
  ABZ_REQUIRE(bits != bits_other && slot0 != slot1, "smc_prologue_packed: the two bit arrays / slots must differ");
  abz_mc_chain_break(ctx);             /* the partition moves distances: an abcdemc chain no longer describes them */
  double out[6];
  int rc = abz_prologue_packed_impl(ctx, delta, N, n_prev, wns, alive, alpha, eps_prev, eps_target, eps_k_old, ess_min, bits,
                                    bits_other, slot0, slot1, logpi, delta, out, n_alive, partitioned);
  if (rc) return rc;
  *eps = out[0]; *wnorm = out[2]; *ess = out[3];
  if (q) *q = out[1];
  if (dmin) *dmin = out[4];
  if (dmax) *dmax = out[5];
  return 0;
}

int abcdez_smc_swarm_packed(abcdez_ctx* ctx, const uint32_t* bits, uint32_t* bits_out, int64_t n_alive, int64_t r_lo,
                            int64_t r_hi, double* slot0, double* slot1, double* logpi, double* delta, uint8_t* flags,
                            double eps, double gamma0, double gamma_sigma, uint32_t sweep, int64_t* nacc, int64_t* nsim) {
  ABZ_REQUIRE(ctx && bits && bits_out && slot0 && slot1 && logpi && delta, "smc_swarm_packed: null argument");
  abz_population_written(ctx);          /* select enqueued ahead, abcdemc chain: they no longer describe the population */
  ABZ_REQUIRE((nacc == nullptr) == (nsim == nullptr), "smc_swarm_packed: pass both counters or neither");
  /* the reference's donor loops (smc:119-126) never terminate with fewer than 3 alive particles */
  ABZ_REQUIRE(n_alive >= 3 && n_alive <= ABZ_MAX_N, "smc_swarm: needs at least 3 alive particles");
  ABZ_REQUIRE_LANES(ctx, n_alive, "smc_swarm_packed");
  ABZ_REQUIRE(0 <= r_lo && r_lo <= r_hi && r_hi <= n_alive, "smc_swarm_packed: position range out of bounds");
  ABZ_REQUIRE((r_lo % ABZ_PACKED_ALIGN == 0 || r_lo == n_alive) && (r_hi % ABZ_PACKED_ALIGN == 0 || r_hi == n_alive),
              "smc_swarm_packed: a sub-range must start and end at multiples of 64 positions (or at n_alive)");
  ABZ_REQUIRE(slot0 != slot1 && bits != bits_out, "smc_swarm_packed: the two slots / bit arrays must differ");
  const unsigned long long* stop = nullptr;
  if (ctx->grp_k >= 0) {            /* inside abcdez_smc_group_begin / _end: sweep k > 0 runs only while the test of smc:352 has not held */
    ABZ_REQUIRE(nacc == nullptr, "smc_swarm_packed: inside a group of sweeps the counters come from abcdez_smc_group_end");
    if (ctx->grp_k > 0) stop = ctx->d_scal + ABZ_S_GRP_STOP;
  }
  ctx->cur_sweep_k = ctx->grp_k > 0 ? ctx->grp_k : 0;
  int rc = abz_launch_smc_swarm_packed(ctx, bits, bits_out, (uint32_t)n_alive, (uint32_t)r_lo, (uint32_t)r_hi, slot0, slot1,
                                       logpi, delta, flags, eps, gamma0, gamma_sigma, sweep, nacc != nullptr, stop);
  ctx->cur_sweep_k = 0;
  if (rc || !nacc) return rc;       /* no counters wanted: no host synchronisation (smc_replay_packed reports totals) */
  rc = read_counters(ctx);
  if (rc) return rc;
  *nacc = (int64_t)ctx->h_scal[ABZ_S_NACC];
  *nsim = (int64_t)ctx->h_scal[ABZ_S_NSIM];
  if (ctx->h_scal[ABZ_S_PART_ERR] != 0) {       /* reported once: the flag is cleared so that the context stays usable */
    ABZ_HIP_CHECK(hipMemsetAsync(ctx->d_scal + ABZ_S_PART_ERR, 0, 8, ctx->stream));
    abz_set_error("smc_partition: the alive flags did not describe a prefix of length n_prev");
    return -1;
  }
  return 0;
}

/* Arms the NEXT abcdez_smc_sweeps_packed call: behind its last sweep it also enqueues the first third of the next generation's
 * prologue (extrema, rank select, eps of smc:301 with eps_prev = the sweeps' eps and n_prev = their n_alive), which only reads the
 * distances and flags -- so the device works on it while the host reads the sweeps' counters and applies its stop rules.  A
 * following abcdez_smc_prologue_packed with the same arguments starts at the reweight; any other call discards the work. */
int abcdez_smc_select_ahead(abcdez_ctx* ctx, const double* delta, const uint8_t* alive, int64_t N, double alpha, double eps_target) {
  ABZ_REQUIRE(ctx && delta && alive, "smc_select_ahead: null argument");
  ABZ_REQUIRE(N >= 1 && N <= ABZ_MAX_N && alpha >= 0.0 && alpha <= 1.0 && eps_target >= 0.0, "smc_select_ahead: bad argument");
  ctx->ahead = abz_ahead{};
  ctx->ahead.armed = true;
  ctx->ahead.delta = delta; ctx->ahead.alive = alive; ctx->ahead.N = N; ctx->ahead.alpha = alpha; ctx->ahead.eps_target = eps_target;
  return 0;
}

/* Diagnostics: how many prologues found their select already enqueued behind the previous generation's sweeps, and how many had
 * to run it themselves (the first generation, the one after a resample, or a host that invalidated it). */
int abcdez_smc_select_stats(abcdez_ctx* ctx, int64_t* reused, int64_t* inline_runs) {
  ABZ_REQUIRE(ctx && reused && inline_runs, "smc_select_stats: null argument");
  *reused = ctx->n_select_reused; *inline_runs = ctx->n_select_inline;
  return 0;
}

/* Diagnostics: rank passes of abcdemc generations that launched both sorts of the tail / only the one-workgroup LDS sort / only the
 * radix sort (the host launches one alone when it holds a proved bound of the tail's length, abcdez_mc_generation_async). */
int abcdez_mc_rank_stats(abcdez_ctx* ctx, int64_t* both, int64_t* small_only, int64_t* long_only) {
  ABZ_REQUIRE(ctx && both && small_only && long_only, "mc_rank_stats: null argument");
  *both = ctx->n_rank_paths[0]; *small_only = ctx->n_rank_paths[1]; *long_only = ctx->n_rank_paths[2];
  return 0;
}

/* the spec's rule (include/abcdez_spec.h): does a generation that reads n_above distances > eps_target among N draw its
 * better particles by rejection?  1 / 0; -1 for arguments out of range.  Pure function: no context. */
int abcdez_mc_draws_by_rejection(int64_t n_above, int64_t N) {
  if (n_above < 0 || N < 1 || n_above > N) return -1;
  return abz_mc_draws_by_rejection((uint64_t)n_above, (uint64_t)N);
}
int abcdez_mc_draw_stats(abcdez_ctx* ctx, int64_t* by_rejection_no_rank_pass) {
  ABZ_REQUIRE(ctx && by_rejection_no_rank_pass, "mc_draw_stats: null argument");
  *by_rejection_no_rank_pass = ctx->n_mc_reject_gens;
  return 0;
}

/* Forget a select enqueued (or armed) ahead.  The library does so itself in every entry point that changes the distances or
 * the flags; a host that writes them by other means (a copy into the arrays, a resumed checkpoint) or that ends a run says so
 * here -- otherwise a later prologue with equal arguments would reuse eps and extrema made from the old contents. */
int abcdez_smc_select_discard(abcdez_ctx* ctx) {
  ABZ_REQUIRE(ctx, "smc_select_discard: null context");
  abz_population_written(ctx);         /* also ends an abcdemc chain: its tail bound was proved for the old distances */
  return 0;
}

/* The sweeps of one generation (smc:336-353) in ONE enqueue and ONE read-back: sweep k+1 is launched behind a device-side
 * evaluation of the early-exit test `sum(naccs) / n_alive >= Kmcmc_min` (smc:352) on the counters of sweeps 1..k, and
 * returns at once when it holds.  Same arithmetic as the host's test (one IEEE division of exactly represented integers). */
int abcdez_smc_sweeps_packed(abcdez_ctx* ctx, uint32_t* bits_a, uint32_t* bits_b, int64_t n_alive, double* slot0, double* slot1,
                             double* logpi, double* delta, double eps, double gamma0, double gamma_sigma, uint32_t sweep0,
                             int32_t k_max, double kmcmc_min, int64_t* nacc, int64_t* nsim, int32_t* k_done) {
  ABZ_REQUIRE(ctx && bits_a && bits_b && slot0 && slot1 && logpi && delta && nacc && nsim && k_done, "smc_sweeps_packed: null argument");
  ABZ_REQUIRE(n_alive >= 3 && n_alive <= ABZ_MAX_N, "smc_swarm: needs at least 3 alive particles");
  ABZ_REQUIRE_LANES(ctx, n_alive, "smc_sweeps_packed");
  ABZ_REQUIRE(1 <= k_max && k_max <= ABZ_GROUP_MAX, "smc_sweeps_packed: 1 <= k_max <= 16 sweeps per call");
  ABZ_REQUIRE(slot0 != slot1 && bits_a != bits_b, "smc_sweeps_packed: the two slots / bit arrays must differ");
  ABZ_REQUIRE(kmcmc_min >= 0.0, "smc_sweeps_packed: Kmcmc_min must not be negative");
  abz_mc_chain_break(ctx);             /* the sweeps write distances */
  /* the group's acceptances are counted from the (nacc, nsim) slot totals at the last read-back.  Every call that adds to those
   * two classes reads them back before it returns; a sweep launched WITHOUT counters (abcdez_smc_swarm_packed with nacc = NULL,
   * the sharded path) adds to the ABZ_C_DISCARD classes instead, so it cannot leak into the device-side test of smc:352 */
  const unsigned long long base_acc = ctx->cnt_prev[ABZ_C_NACC], base_sim = ctx->cnt_prev[ABZ_C_NSIM];
  const abz_ahead armed = ctx->ahead;
  ctx->ahead = abz_ahead{};
  /* timing mode 2: ONE sweep of the group carries the event pair, and which one rotates from call to call -- the first sweep of a
   * generation runs on a population the partition has just moved and is a few per cent slower than its siblings */
  const int timed_k = ctx->timing_first_only ? (int)(ctx->timing_rot++ % k_max) : -1;
  /* timing mode 3: one event pair around the whole group -- sweeps 2 .. k_max start behind their predecessor like every
   * un-instrumented launch does, instead of on the queue a pair of their own has just drained */
  const int group_tk = ctx->timing_group ? abz_time_begin(ctx) : -1;
  for (int k = 0; k < k_max; ++k) {
    uint32_t* in = (k & 1) ? bits_b : bits_a;
    uint32_t* out = (k & 1) ? bits_a : bits_b;
    const bool timing = ctx->timing;
    if ((ctx->timing_first_only && k != timed_k) || ctx->timing_group) ctx->timing = false;
    ctx->cur_sweep_k = k;
    int rc = abz_launch_smc_swarm_packed(ctx, in, out, (uint32_t)n_alive, 0u, (uint32_t)n_alive, slot0, slot1, logpi, delta,
                                         nullptr, eps, gamma0, gamma_sigma, sweep0 + (uint32_t)k, 1,
                                         k ? ctx->d_scal + ABZ_S_GRP_STOP : nullptr);
    ctx->timing = timing;
    ctx->cur_sweep_k = 0;
    if (rc) return rc;
    if (k + 1 < k_max) {             /* nothing is decided after the last sweep: its counters are the totals the host reads anyway */
      rc = abz_launch_group_check(ctx, k, base_acc, (uint32_t)n_alive, kmcmc_min, ABZ_C_NACC);
      if (rc) return rc;
    }
  }
  if (group_tk >= 0) {
    abz_time_end(ctx, group_tk, (long long)n_alive);
    ctx->ev_group[group_tk] = k_max;
  }
  unsigned long long pub = 0;
  if (int rc = abz_publish_launch(ctx, ABZ_S_N, &pub)) return rc;
  if (armed.armed && armed.delta == delta && n_alive <= armed.N) {
    abz_ahead a = armed;
    a.n_prev = n_alive; a.eps_prev = eps;
    if (int rc = abz_prologue_select_enqueue(ctx, a.delta, a.alive, a.N, a.n_prev, a.alpha, a.eps_prev, a.eps_target, &a.j)) return rc;
    a.armed = false; a.valid = true;
    ctx->ahead = a;
  }
  if (int rc = abz_publish_wait(ctx, ABZ_S_N, pub)) return rc;
  /* totals of the two counter classes now; how many sweeps ran: the test held after sweep `GRP_DONE` (stop flag set by one of
   * the k_max - 1 checks), or never -- then the last sweep ran too */
  unsigned long long tot_acc = 0, tot_sim = 0;
  for (int q = 0; q < ABZ_CSLOTS; ++q) {
    tot_acc += ctx->h_scal[ABZ_S_CSLOT0 + q * ABZ_CSTRIDE + ABZ_C_NACC];
    tot_sim += ctx->h_scal[ABZ_S_CSLOT0 + q * ABZ_CSTRIDE + ABZ_C_NSIM];
  }
  const bool stopped = k_max > 1 && ctx->h_scal[ABZ_S_GRP_STOP] != 0;
  const int done = stopped ? (int)ctx->h_scal[ABZ_S_GRP_DONE] : k_max;
  ABZ_REQUIRE(1 <= done && done <= k_max, "smc_sweeps_packed: inconsistent sweep count read back");
  /* every event pair carries the index of the sweep it brackets: one enqueued behind a test of smc:352 that held returned at
   * once and is not counted (whatever the timing mode and stride) */
  if (int rc = read_counters_finish(ctx, done)) return rc;
  unsigned long long pa = base_acc, ps = base_sim;
  for (int k = 0; k < k_max; ++k) {
    if (k < done) {
      const bool last_unchecked = !stopped && k == k_max - 1;
      const unsigned long long ca = last_unchecked ? tot_acc : ctx->h_scal[ABZ_S_GRP_SNAP + 2 * k];
      const unsigned long long cs = last_unchecked ? tot_sim : ctx->h_scal[ABZ_S_GRP_SNAP + 2 * k + 1];
      nacc[k] = (int64_t)(ca - pa); nsim[k] = (int64_t)(cs - ps);
      pa = ca; ps = cs;
    } else {
      nacc[k] = 0; nsim[k] = 0;
    }
  }
  *k_done = done;
  if (ctx->h_scal[ABZ_S_PART_ERR] != 0) {
    ABZ_HIP_CHECK(hipMemsetAsync(ctx->d_scal + ABZ_S_PART_ERR, 0, 8, ctx->stream));
    abz_set_error("smc_partition: the alive flags did not describe a prefix of length n_prev");
    return -1;
  }
  return 0;
}

/* ONE generation of abcdesmc!'s loop body (smc:301-353) on an unsharded packed population in ONE call: the prologue (eps of smc:301,
 * weights, ESS, extrema, partition), the resampling when ESS < ess_min (smc:323-326), the Kmcmc sweeps behind the device-side test of
 * smc:352 -- the entry points above, composed, with the two decisions between them (resample or not, sweep or not) taken HERE in a few
 * hundred nanoseconds instead of in the host language between three calls.  Nothing new happens on the device; what changes is how long
 * the queue stands empty while an interpreter walks from one call to the next (the partition and the select ahead hide ~50 us each,
 * and a Python host needs most of that).  `bits_cur` names the current slot of every position, (logpi_cur, delta_cur) is the current
 * pair; a resampling gathers into the other pair, which is the current one afterwards (`resampled` tells the host to swap its names).
 * select_ahead: arm the next generation's select behind these sweeps (abcdez_smc_select_ahead with the same alpha, eps_target). */
int abcdez_smc_generation_packed(abcdez_ctx* ctx, int64_t N, int64_t n_prev, uint32_t* bits_cur, uint32_t* bits_oth, double* slot0,
                                 double* slot1, double* logpi_cur, double* delta_cur, double* logpi_oth, double* delta_oth, double* wns,
                                 uint8_t* alive, uint32_t* inds, double alpha, double eps_prev, double eps_target, double eps_k_old,
                                 double ess_min, double gamma0, double gamma_sigma, uint32_t sweep0, uint32_t draw, int32_t k_max,
                                 double kmcmc_min, int32_t select_ahead, double* eps, double* wnorm, double* ess, int64_t* n_alive,
                                 int32_t* partitioned, int32_t* resampled, double* ess_resampled, int64_t* n_swept, int64_t* nacc,
                                 int64_t* nsim, int32_t* k_done, double* dmin, double* dmax) {
  ABZ_REQUIRE(ctx && logpi_oth && delta_oth && inds && eps && wnorm && ess && n_alive && partitioned && resampled && ess_resampled &&
              n_swept && nacc && nsim && k_done, "smc_generation_packed: null argument");
  ABZ_REQUIRE(logpi_cur != logpi_oth && delta_cur != delta_oth, "smc_generation_packed: the two (logpi, delta) pairs must differ");
  ABZ_REQUIRE(1 <= k_max && k_max <= ABZ_GROUP_MAX && kmcmc_min >= 0.0, "smc_generation_packed: 1 <= Kmcmc <= 16 sweeps per call, Kmcmc_min >= 0");
  ABZ_REQUIRE(ctx->comm_kind == ABZ_COMM_NONE || ctx->comm_world == 1, "smc_generation_packed: an unsharded population (sharded: abcdez_smc_sweeps_sharded)");
  *resampled = 0; *ess_resampled = 0.0; *k_done = 0; *n_swept = 0;
  for (int k = 0; k < k_max; ++k) { nacc[k] = 0; nsim[k] = 0; }
  int rc = abcdez_smc_prologue_packed(ctx, delta_cur, wns, alive, N, n_prev, alpha, eps_prev, eps_target, eps_k_old, ess_min, bits_cur,
                                      bits_oth, slot0, slot1, logpi_cur, eps, nullptr, wnorm, ess, n_alive, partitioned, dmin, dmax);
  if (rc) return rc;
  int64_t n = *n_alive;
  double* lp = logpi_cur;
  double* dl = delta_cur;
  if (n > 0 && *ess < ess_min) {                                           /* smc:323-326 */
    rc = abcdez_wsample_stratified(ctx, wns, N, draw, inds);
    if (rc == 0) rc = abcdez_smc_resample_gather_packed(ctx, inds, N, bits_cur, bits_oth, slot0, slot1, logpi_cur, delta_cur, logpi_oth, delta_oth, wns, alive);
    if (rc) return rc;
    if (ctx->stamp_cur) { uint64_t* t = ctx->stamp_cur; ctx->stamp_cur = ctx->stamp_nxt; ctx->stamp_nxt = t; }   /* the stamps were gathered with the distances */
    rc = abcdez_get_ess(ctx, wns, N, ess_resampled);                        /* what the drivers record after a resampling (smc:325) */
    if (rc) return rc;
    *resampled = 1;
    n = N; lp = logpi_oth; dl = delta_oth;
  }
  *n_swept = n;
  if (n < 3) return 0;                                                      /* the donor draws need three alive particles (smc:119-126) */
  if (select_ahead && *eps > eps_target) {                                  /* (at eps_target this is the run's last generation) */
    rc = abcdez_smc_select_ahead(ctx, dl, alive, N, alpha, eps_target);
    if (rc) return rc;
  }
  return abcdez_smc_sweeps_packed(ctx, bits_cur, bits_oth, n, slot0, slot1, lp, dl, *eps, gamma0, gamma_sigma, sweep0, k_max, kmcmc_min, nacc,
                                  nsim, k_done);
}

int abcdez_smc_replay_packed(abcdez_ctx* ctx, const uint32_t* bits, uint32_t* bits_out, int64_t n_alive, int64_t skip_lo,
                             int64_t skip_hi, double* slot0, double* slot1, double* logpi, const uint8_t* flags,
                             double gamma0, double gamma_sigma, uint32_t sweep, int64_t* nacc, int64_t* nsim) {
  ABZ_REQUIRE(ctx && bits && bits_out && slot0 && slot1 && logpi && flags && nacc && nsim, "smc_replay_packed: null argument");
  abz_population_written(ctx);          /* select enqueued ahead, abcdemc chain: they no longer describe the population */
  ABZ_REQUIRE(n_alive >= 3 && n_alive <= ABZ_MAX_N, "smc_replay_packed: needs at least 3 alive particles");
  ABZ_REQUIRE_LANES(ctx, n_alive, "smc_replay_packed");
  ABZ_REQUIRE(0 <= skip_lo && skip_lo <= skip_hi && skip_hi <= n_alive, "smc_replay_packed: position range out of bounds");
  ABZ_REQUIRE((skip_lo % ABZ_PACKED_ALIGN == 0 || skip_lo == n_alive) && (skip_hi % ABZ_PACKED_ALIGN == 0 || skip_hi == n_alive),
              "smc_replay_packed: the own range must start and end at multiples of 64 positions (or at n_alive)");
  ABZ_REQUIRE(slot0 != slot1 && bits != bits_out, "smc_replay_packed: the two slots / bit arrays must differ");
  ABZ_REQUIRE(ctx->grp_k < 0, "smc_replay_packed: inside a group of sweeps use abcdez_smc_group_replay");
  int rc = abz_launch_smc_replay_packed(ctx, bits, bits_out, (uint32_t)n_alive, (uint32_t)skip_lo, (uint32_t)skip_hi, slot0,
                                        slot1, logpi, flags, gamma0, gamma_sigma, sweep, nullptr);
  if (rc) return rc;
  rc = read_counters(ctx);
  if (rc) return rc;
  *nacc = (int64_t)ctx->h_scal[ABZ_S_RACC];
  *nsim = (int64_t)ctx->h_scal[ABZ_S_RSIM];
  return 0;
}

/* ---- the sweeps of one generation on a SHARDED population in one host synchronisation (smc:336-353).
 *   abcdez_smc_group_begin(n_alive, Kmcmc_min)
 *   k = 0 .. Kmcmc-1:  abcdez_smc_swarm_packed(own range, flags, nacc = NULL)   -- sweep k > 0 is gated by the stop flag
 *                      [the host all-gathers the flag bytes: a collective every rank executes, stopped or not]
 *                      abcdez_smc_group_replay(...)                              -- gated too; counts both flag bits over the prefix
 *                                                                                  and evaluates the test of smc:352 on the device
 *   abcdez_smc_group_publish()     -- enqueue the read-back (the host may enqueue more work -- the distance exchange -- behind it)
 *   abcdez_smc_group_end(nacc[], nsim[], &k_done)
 * Every replica counts the same flags, so every rank takes the same decision without talking to the others. */
int abcdez_smc_group_begin(abcdez_ctx* ctx, int64_t n_alive, double kmcmc_min) {
  ABZ_REQUIRE(ctx, "smc_group_begin: null context");
  ABZ_REQUIRE(ctx->grp_k < 0, "smc_group_begin: a group is already open");
  ABZ_REQUIRE(n_alive >= 3 && n_alive <= ABZ_MAX_N && kmcmc_min >= 0.0, "smc_group_begin: bad argument");
  ABZ_REQUIRE_LANES(ctx, n_alive, "smc_group_begin");
  ctx->ahead = abz_ahead{};
  ctx->grp_k = 0; ctx->grp_n_alive = n_alive; ctx->grp_kmin = kmcmc_min; ctx->grp_pub = 0;
  ctx->grp_base_acc = ctx->cnt_prev[ABZ_C_RACC]; ctx->grp_base_sim = ctx->cnt_prev[ABZ_C_RSIM];
  return 0;
}
int abcdez_smc_group_replay(abcdez_ctx* ctx, const uint32_t* bits, uint32_t* bits_out, int64_t skip_lo, int64_t skip_hi, double* slot0,
                            double* slot1, double* logpi, const uint8_t* flags, double gamma0, double gamma_sigma, uint32_t sweep) {
  ABZ_REQUIRE(ctx && bits && bits_out && slot0 && slot1 && logpi && flags, "smc_group_replay: null argument");
  ABZ_REQUIRE(ctx->grp_k >= 0 && ctx->grp_k < ABZ_GROUP_MAX, "smc_group_replay: no group open (or more than 16 sweeps)");
  const int64_t n_alive = ctx->grp_n_alive;
  ABZ_REQUIRE(0 <= skip_lo && skip_lo <= skip_hi && skip_hi <= n_alive, "smc_group_replay: position range out of bounds");
  ABZ_REQUIRE((skip_lo % ABZ_PACKED_ALIGN == 0 || skip_lo == n_alive) && (skip_hi % ABZ_PACKED_ALIGN == 0 || skip_hi == n_alive),
              "smc_group_replay: the own range must start and end at multiples of 64 positions (or at n_alive)");
  ABZ_REQUIRE(slot0 != slot1 && bits != bits_out, "smc_group_replay: the two slots / bit arrays must differ");
  const int k = ctx->grp_k;
  abz_mc_chain_break(ctx);
  int rc = abz_launch_smc_replay_packed(ctx, bits, bits_out, (uint32_t)n_alive, (uint32_t)skip_lo, (uint32_t)skip_hi, slot0, slot1,
                                        logpi, flags, gamma0, gamma_sigma, sweep, k ? ctx->d_scal + ABZ_S_GRP_STOP : nullptr);
  if (rc) return rc;
  rc = abz_launch_group_check(ctx, k, ctx->grp_base_acc, (uint32_t)n_alive, ctx->grp_kmin, ABZ_C_RACC);
  if (rc) return rc;
  ctx->grp_k = k + 1;
  return 0;
}
/* Abandon an open group (a collective or a launch failed between _begin and _end): the context accepts a new group and the
 * counter-returning calls again.  Event pairs of the abandoned sweeps are dropped, the counter baselines re-read. */
int abcdez_smc_group_abort(abcdez_ctx* ctx) {
  ABZ_REQUIRE(ctx, "smc_group_abort: null context");
  if (ctx->grp_k < 0) return 0;
  ctx->grp_k = -1; ctx->grp_pub = 0;
  abz_population_written(ctx);
  ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  ctx->ev_head = ctx->ev_tail;                               /* timing pairs of the abandoned sweeps */
  ABZ_HIP_CHECK(hipMemsetAsync(ctx->d_scal + ABZ_S_GRP_STOP, 0, 16, ctx->stream));
  return read_counters(ctx, 0);                              /* baselines := the slots' totals now */
}
int abcdez_smc_group_publish(abcdez_ctx* ctx) {
  ABZ_REQUIRE(ctx && ctx->grp_k > 0 && ctx->grp_pub == 0, "smc_group_publish: needs an open group with at least one sweep");
  return abz_publish_launch(ctx, ABZ_S_N, &ctx->grp_pub);
}
int abcdez_smc_group_end(abcdez_ctx* ctx, int64_t* nacc, int64_t* nsim, int32_t* k_done) {
  ABZ_REQUIRE(ctx && nacc && nsim && k_done, "smc_group_end: null argument");
  ABZ_REQUIRE(ctx->grp_k > 0, "smc_group_end: no group open, or no sweep in it");
  const int k_max = ctx->grp_k;
  ctx->grp_k = -1;
  if (ctx->grp_pub == 0)
    if (int rc = abz_publish_launch(ctx, ABZ_S_N, &ctx->grp_pub)) return rc;
  if (int rc = abz_publish_wait(ctx, ABZ_S_N, ctx->grp_pub)) return rc;
  /* every replay is followed by a check, so snapshot k holds the totals after sweep k; the stop flag says whether the test
   * held after sweep GRP_DONE (the later launches returned at once) */
  const int done = (int)ctx->h_scal[ABZ_S_GRP_DONE];
  ABZ_REQUIRE(1 <= done && done <= k_max, "smc_group_end: inconsistent sweep count read back");
  if (int rc = read_counters_finish(ctx, done)) return rc;      /* own-range sweeps behind a test that held are not launches that did work */
  unsigned long long pa = ctx->grp_base_acc, ps = ctx->grp_base_sim;
  for (int k = 0; k < k_max; ++k) {
    if (k < done) {
      const unsigned long long ca = ctx->h_scal[ABZ_S_GRP_SNAP + 2 * k], cs = ctx->h_scal[ABZ_S_GRP_SNAP + 2 * k + 1];
      nacc[k] = (int64_t)(ca - pa); nsim[k] = (int64_t)(cs - ps);
      pa = ca; ps = cs;
    } else {
      nacc[k] = 0; nsim[k] = 0;
    }
  }
  *k_done = done;
  return 0;
}

int abcdez_smc_resample_gather_packed(abcdez_ctx* ctx, const uint32_t* inds, int64_t N, uint32_t* bits, uint32_t* bits_other,
                                      double* slot0, double* slot1, const double* logpi, const double* delta,
                                      double* nlogpi, double* ndelta, double* wns, uint8_t* alive) {
  ABZ_REQUIRE(ctx && inds && bits && bits_other && slot0 && slot1 && logpi && delta && nlogpi && ndelta && wns && alive,
              "smc_resample_gather_packed: null argument");
  abz_population_written(ctx);          /* select enqueued ahead, abcdemc chain: they no longer describe the population */
  ABZ_REQUIRE(N >= 1 && N <= ABZ_MAX_N, "smc_resample_gather_packed: N out of range");
  ABZ_REQUIRE_LANES(ctx, N, "smc_resample_gather_packed");
  ABZ_REQUIRE(logpi != nlogpi && delta != ndelta && slot0 != slot1 && bits != bits_other,
              "smc_resample_gather_packed: in/out arrays must differ");
  ctx->w_uniform = true;                /* Wns .= 1/N (smc:102) */
  return abz_launch_resample_gather_packed(ctx, inds, (uint32_t)N, bits, bits_other, slot0, slot1, logpi, delta, nlogpi,
                                           ndelta, wns, alive);
}

int abcdez_packed_gather(abcdez_ctx* ctx, const uint32_t* bits, int64_t N, const double* slot0, const double* slot1,
                         double* out) {
  ABZ_REQUIRE(ctx && bits && slot0 && slot1 && out, "packed_gather: null argument");
  ABZ_REQUIRE(N >= 1 && N <= ABZ_MAX_N, "packed_gather: N out of range");
  ABZ_REQUIRE_LANES(ctx, N, "packed_gather");
  return abz_launch_packed_gather(ctx, bits, (uint32_t)N, slot0, slot1, out);
}

int abcdez_smc_reweight(abcdez_ctx* ctx, const double* delta, double* wns, uint8_t* alive, int64_t N, double eps_old,
                        double eps_new, double* wnorm, double* ess, int64_t* n_alive) {
  ABZ_REQUIRE(ctx && delta && wns && alive && wnorm && ess && n_alive, "smc_reweight: null argument");
  abz_population_written(ctx);          /* select enqueued ahead, abcdemc chain: they no longer describe the population */
  ABZ_REQUIRE(N >= 1 && N <= ABZ_MAX_N, "smc_reweight: N out of range");
  ABZ_REQUIRE(eps_old >= 0.0 && eps_new >= 0.0, "Expected ϵ ≥ 0.0");   /* types.jl:30 */
  ctx->w_uniform = false;               /* the general path: weights as its floating sums leave them */
  return abz_reweight_impl(ctx, delta, wns, alive, N, eps_old, eps_new, wnorm, ess, n_alive);
}

int abcdez_get_ess(abcdez_ctx* ctx, const double* wns, int64_t N, double* ess) {
  ABZ_REQUIRE(ctx && wns && ess, "get_ess: null argument");
  ABZ_REQUIRE(N >= 1 && N <= ABZ_MAX_N, "get_ess: N out of range");
  double s;
  int rc = abz_tree_sum_impl(ctx, wns, N, 1, &s);
  if (rc) return rc;
  *ess = 1.0 / s;
  return 0;
}

int abcdez_tree_sum(abcdez_ctx* ctx, const double* x, int64_t n, double* out) {
  ABZ_REQUIRE(ctx && x && out, "tree_sum: null argument");
  ABZ_REQUIRE(n >= 1 && n <= ABZ_MAX_N, "tree_sum: n out of range");
  return abz_tree_sum_impl(ctx, x, n, 0, out);
}

int abcdez_wsample_stratified(abcdez_ctx* ctx, const double* wns, int64_t N, uint32_t draw, uint32_t* inds) {
  ABZ_REQUIRE(ctx && wns && inds, "wsample_stratified: null argument");
  ABZ_REQUIRE(N >= 1 && N <= ABZ_MAX_N, "wsample_stratified: N out of range");
  return abz_stratified_impl(ctx, wns, N, draw, inds);
}

int abcdez_quantile_alive(abcdez_ctx* ctx, const double* delta, const uint8_t* alive, int64_t N, int64_t n_alive_hint,
                          double p, double* q, double* xj, double* xj1) {
  ABZ_REQUIRE(ctx && delta && alive && q, "quantile_alive: null argument");
  abz_population_written(ctx);          /* select enqueued ahead, abcdemc chain: they no longer describe the population */
  ABZ_REQUIRE(N >= 1 && N <= ABZ_MAX_N, "quantile_alive: N out of range");
  ABZ_REQUIRE(p >= 0.0 && p <= 1.0, "quantile_alive: p must be in [0, 1]");
  int64_t n = n_alive_hint, n_le;
  double a, b;
  int rc = 0;
  if (n < 0) rc = abz_count_alive_impl(ctx, alive, N, &n);   /* the caller usually knows sum(alive) from the last reweight */
  if (rc) return rc;
  ABZ_REQUIRE(n <= N, "quantile_alive: n_alive_hint exceeds N");
  ABZ_REQUIRE(n >= 1, "quantile_alive: no alive particles");
  /* Julia Statistics.quantile, type 7: h = (n-1) p + 1, j = clamp(floor(h), 1, n-1), g = h - j */
  const double h = (double)(n - 1) * p + 1.0;
  int64_t j = (int64_t)__builtin_floor(h);
  if (j < 1) j = 1;
  if (j > n - 1) j = n - 1 > 1 ? n - 1 : 1;
  const double g = h - (double)j;
  rc = abz_select_impl(ctx, delta, alive, N, j - 1, &a, &b, &n_le);
  if (rc) return rc;
  if (n == 1) b = a;
  *q = a + g * (b - a);
  if (xj) *xj = a;
  if (xj1) *xj1 = b;
  return 0;
}

int abcdez_extrema(abcdez_ctx* ctx, const double* delta, int64_t N, double* lo, double* hi) {
  ABZ_REQUIRE(ctx && delta && lo && hi, "extrema: null argument");
  ctx->ahead = abz_ahead{};            /* shares the extrema scalars with a select enqueued ahead */
  ABZ_REQUIRE(N >= 1 && N <= ABZ_MAX_N, "extrema: N out of range");
  return abz_extrema_impl(ctx, delta, N, lo, hi);
}

int abcdez_count_gt(abcdez_ctx* ctx, const double* delta, int64_t N, double thr, int64_t* count) {
  ABZ_REQUIRE(ctx && delta && count, "count_gt: null argument");
  ABZ_REQUIRE(N >= 1 && N <= ABZ_MAX_N, "count_gt: N out of range");
  const int rc = abz_count_gt_impl(ctx, delta, N, thr, count);
  if (rc == 0) { ctx->mc_count_seen.delta = delta; ctx->mc_count_seen.N = N; ctx->mc_count_seen.thr = thr;
                 ctx->mc_count_seen.count = *count; ctx->mc_count_seen.chain = ctx->mc_chain; }     /* valid until something writes distances */
  return rc;
}

int abcdez_mc_rank_prepare(abcdez_ctx* ctx, const double* delta, int64_t N, double eps_pop, double dmax_hint,
                           uint32_t* order, double* sorted_delta, uint32_t* cnt) {
  ABZ_REQUIRE(ctx && delta && order && sorted_delta && cnt, "mc_rank_prepare: null argument");
  ABZ_REQUIRE(N >= 1 && N <= ABZ_MAX_N, "mc_rank_prepare: N out of range");
  ABZ_REQUIRE(eps_pop == eps_pop, "mc_rank_prepare: eps_pop is NaN");
  return abz_rank_prepare_impl(ctx, delta, N, eps_pop, dmax_hint, order, sorted_delta, cnt, nullptr, -1, -1);
}

int abcdez_mc_swarm(abcdez_ctx* ctx, const uint32_t* order, const uint32_t* cnt, int64_t N, const double* theta,
                    const double* logpi, const double* delta, double* ntheta, double* nlogpi, double* ndelta,
                    double eps_pop, double eps_target, double gamma0, double gamma_sigma, int64_t i0, int64_t n_local,
                    uint32_t sweep, int64_t* nsim, int64_t* n_above_target, double* dmin, double* dmax) {
  ABZ_REQUIRE(ctx && theta && logpi && delta && ntheta && nlogpi && ndelta && nsim, "mc_swarm: null argument");
  /* order == cnt == NULL: the better particle of mc:23 by rejection (include/abcdez_spec.h says when a host may ask for it) */
  ABZ_REQUIRE((order == nullptr) == (cnt == nullptr), "mc_swarm: pass both order and cnt (draws by rank) or neither (by rejection)");
  abz_population_written(ctx);          /* select enqueued ahead, abcdemc chain: they no longer describe the population */
  ABZ_REQUIRE(N >= 5 && N <= ABZ_MAX_N, "nparticles must be at least 5");   /* mc:109 */
  ABZ_REQUIRE_LANES(ctx, N, "mc_swarm");
  ABZ_REQUIRE(i0 >= 0 && n_local >= 0 && i0 + n_local <= N, "mc_swarm: particle range out of bounds");
  ABZ_REQUIRE(theta != ntheta && logpi != nlogpi && delta != ndelta, "mc_swarm: in/out arrays must differ (synchronous update)");
  int rc = abz_launch_mc_swarm(ctx, order, cnt, (uint32_t)N, theta, logpi, delta, ntheta, nlogpi, ndelta,
                               eps_pop, eps_target, gamma0, gamma_sigma, (uint32_t)i0, (uint32_t)n_local, sweep, nullptr);
  if (rc) return rc;
  const int bank = ctx->mm_bank;
  if (n_local > 0) ctx->mm_bank = 1 - bank;          /* the kernel reset the other bank for the next sweep */
  ctx->mc_have_bank = (i0 == 0 && n_local == N);     /* the bank holds the extrema of the whole population */
  ctx->mc_window_ready = false;
  rc = read_counters(ctx);
  if (rc) return rc;
  if (ctx->h_scal[ABZ_S_MC_REJFAIL] == 2ull) {
    ABZ_HIP_CHECK(hipMemsetAsync(ctx->d_scal + ABZ_S_MC_REJFAIL, 0, 8, ctx->stream));
    abz_set_error("mc_swarm: order / cnt do not enumerate this population (cnt[i] or order[] out of range): run abcdez_mc_rank_prepare "
                  "on the distances the sweep reads");
    return -3;
  }
  if (ctx->h_scal[ABZ_S_MC_REJFAIL] != 0ull) {
    ABZ_HIP_CHECK(hipMemsetAsync(ctx->d_scal + ABZ_S_MC_REJFAIL, 0, 8, ctx->stream));
    abz_set_error("mc_swarm: a particle drawing its better particle by rejection (order = NULL) ran out of trials: fewer than "
                  "1 / 16 of the particles lie at or below eps_target -- such a generation draws by rank (include/abcdez_spec.h)");
    return -3;
  }
  *nsim = (int64_t)ctx->h_scal[ABZ_S_COUNT];
  if (n_above_target) *n_above_target = (int64_t)ctx->h_scal[ABZ_S_MCGT];
  if (dmin || dmax) {
    double lo = ABZ_INF, hi = ABZ_NINF;
    if (n_local > 0) abz_fold_minmax(ctx, bank, &lo, &hi);
    if (dmin) *dmin = lo;
    if (dmax) *dmax = hi;
  }
  return 0;
}

/* one abcdemc generation in one call: the rank pass (when the population is not converged) and the sweep */
int abcdez_mc_generation(abcdez_ctx* ctx, int64_t N, const double* theta, const double* logpi, const double* delta,
                         double* ntheta, double* nlogpi, double* ndelta, uint32_t* order, double* sorted_delta, uint32_t* cnt,
                         double eps_pop, double eps_target, double dmax, int64_t n_above, double gamma0, double gamma_sigma,
                         uint32_t sweep, int64_t* nsim, int64_t* n_above_target, double* dmin, double* dmax_out) {
  ABZ_REQUIRE(ctx && order && sorted_delta && cnt && delta, "mc_generation: null argument");
  ABZ_REQUIRE(N >= 5 && N <= ABZ_MAX_N, "nparticles must be at least 5");   /* mc:109 */
  ABZ_REQUIRE_LANES(ctx, N, "mc_generation");
  ABZ_REQUIRE(n_above >= -1 && n_above <= N, "mc_generation: n_above = #(Ds > eps_target) of the distances read, or -1");
  if (n_above < 0) {                 /* the caller does not carry mc:156 of the generation before: count */
    if (int rc = abz_count_gt_impl(ctx, delta, N, eps_target, &n_above)) return rc;
  }
  const bool reject = abz_mc_draws_by_rejection((uint64_t)n_above, (uint64_t)N) != 0;
  if (dmax > eps_target && !reject) {           /* mc:20-24 is only reached while some Ds[i] > eps */
    const int rc = abcdez_mc_rank_prepare(ctx, delta, N, eps_pop, dmax, order, sorted_delta, cnt);
    if (rc) return rc;
  }
  return abcdez_mc_swarm(ctx, reject ? nullptr : order, reject ? nullptr : cnt, N, theta, logpi, delta, ntheta, nlogpi, ndelta,
                         eps_pop, eps_target, gamma0, gamma_sigma, 0, N, sweep, nsim, n_above_target, dmin, dmax_out);
}

/* abcdemc!'s loop body (mc:146-149) WITHOUT a host synchronisation.  The population extrema of mc:146 come from the
 * sweep before (still on the device) unless lo_hi is given; eps_pop = max(eps_target, lo + alpha (hi - lo)) (mc:147) is
 * evaluated on the device with the host driver's operations; then the rank pass (do_rank) and the sweep.  A snapshot of
 * the counters and extrema is copied to pinned memory behind an event: abcdez_mc_generation_wait redeems the tickets in
 * the order they were issued (at most ABZ_MC_RING of them in flight). */
} /* extern "C" */
static int mc_generation_async_impl(abcdez_ctx* ctx, int64_t N, const double* theta, const double* logpi, const double* delta,
                                    double* ntheta, double* nlogpi, double* ndelta, uint32_t* order, double* sorted_delta,
                                    uint32_t* cnt, double alpha, double eps_target, const double* lo_hi, int32_t do_rank,
                                    double gamma0, double gamma_sigma, uint32_t sweep, int64_t* ticket, const bool sharded) {
  ABZ_REQUIRE(ctx && order && sorted_delta && cnt && theta && logpi && delta && ntheta && nlogpi && ndelta && ticket,
              "mc_generation_async: null argument");
  /* sharded: this rank sweeps the particles [rank N / world, (rank + 1) N / world); the rank pass, the window and the snapshot run
   * replicated on the whole population, which every rank holds (a better particle or a donor may be anybody, mc:23-32) */
  int64_t i0 = 0, n_local = N;
  if (sharded) {
    ABZ_REQUIRE(ctx->comm_kind != ABZ_COMM_NONE, "mc_generation_sharded_async: no communicator (abcdez_comm_init / abcdez_comm_init_host)");
    ABZ_REQUIRE(!ctx->comm_broken, "mc_generation_sharded_async: the communicator was aborted after a failure on this rank");
    ABZ_REQUIRE(N % ctx->comm_world == 0, "mc_generation_sharded_async: nparticles must be divisible by the number of ranks");
    n_local = N / ctx->comm_world; i0 = (int64_t)ctx->comm_rank * n_local;
    /* the extrema a sweep leaves in its bank are THIS rank's: the window of a sharded chain comes from lo_hi (first generation) or
     * from the snapshot kernel of the generation before, which holds the exchanged ones */
    ABZ_REQUIRE(lo_hi || (ctx->mc_window_ready && ctx->mc_chain_sharded && ctx->mc_alpha == alpha && ctx->mc_eps_target == eps_target),
                "mc_generation_sharded_async: the first generation of a chain (or after a change of alpha / eps_target) needs lo_hi");
  }
  ctx->ahead = abz_ahead{};            /* anything enqueued ahead for the next prologue no longer describes the population */
  ABZ_REQUIRE(N >= 5 && N <= ABZ_MAX_N, "nparticles must be at least 5");   /* mc:109 */
  ABZ_REQUIRE_LANES(ctx, N, "mc_generation_async");
  ABZ_REQUIRE(theta != ntheta && logpi != nlogpi && delta != ndelta, "mc_swarm: in/out arrays must differ (synchronous update)");
  ABZ_REQUIRE(alpha >= 0.0 && alpha <= 1.0 && eps_target >= 0.0, "mc_generation_async: need 0 <= alpha <= 1 and eps_target >= 0");
  ABZ_REQUIRE(ctx->mc_issued - ctx->mc_waited < ABZ_MC_RING, "mc_generation_async: too many generations in flight (redeem a ticket first)");
  ABZ_REQUIRE(lo_hi || ctx->mc_have_bank || (sharded && ctx->mc_window_ready),
              "mc_generation_async: the first generation needs the population's extrema (lo_hi)");
  if (!ctx->h_ring) {
    ABZ_HIP_CHECK(hipHostMalloc((void**)&ctx->h_ring, (size_t)ABZ_MC_RING * ABZ_RING_WORDS * 8, hipHostMallocMapped | hipHostMallocCoherent));
    memset(ctx->h_ring, 0, (size_t)ABZ_MC_RING * ABZ_RING_WORDS * 8);
    ABZ_HIP_CHECK(hipHostGetDevicePointer((void**)&ctx->d_ring, ctx->h_ring, 0));
  }
  int rc = 0;
  if (ctx->mc_seq_dirty) {           /* an earlier generation failed between its launches: re-seat the device's generation counter */
    const unsigned long long v = (unsigned long long)ctx->mc_issued;
    ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ABZ_HIP_CHECK(hipMemcpy(ctx->d_scal + ABZ_S_MCSEQ, &v, 8, hipMemcpyHostToDevice));
    ctx->mc_seq_dirty = false;
  }
  /* a new chain: host-given extrema (the first generation of a run), other parameters, another population than the one the
   * generation before wrote (its outputs are this generation's inputs) */
  if (lo_hi || ctx->mc_alpha != alpha || ctx->mc_eps_target != eps_target || ctx->mc_last_out != (const void*)delta || ctx->mc_last_N != N ||
      ctx->mc_chain_sharded != sharded) {
    /* the driver counted these very distances against this eps_target (mc:133) and nothing has written them since: the
     * chain's first count is known here too, and with it whether its generations need rank passes at all */
    const auto& seen = ctx->mc_count_seen;
    const bool counted = seen.count >= 0 && seen.delta == (const void*)delta && seen.N == N && seen.thr == eps_target && seen.chain == ctx->mc_chain;
    ctx->mc_chain += 1; ctx->mc_tail_bound = -1; ctx->mc_tail_hint = -1;
    ctx->mc_reject_known = counted && abz_mc_draws_by_rejection((uint64_t)seen.count, (uint64_t)N) != 0;
    ctx->mc_count_seen.count = -1;     /* used once: a count describes the distances at the time it was taken */
  }
  /* this generation writes ndelta: a count taken of that buffer earlier (generations ping-pong between two buffers without
   * ending their chain) no longer describes it.  (A stale count could only over-state #(Ds > eps_target), which never grows,
   * mc:54 -- it would cost a rank pass, not correctness; dropping it keeps the cache honest anyway.) */
  if (ctx->mc_count_seen.delta == (const void*)ndelta) ctx->mc_count_seen.count = -1;
  /* how the chain's generations draw their better particles is decided on the device from #(Ds > eps_target): counted once
   * here, carried from sweep to sweep by the snapshot kernel afterwards */
  if (ctx->mc_nabove_chain != ctx->mc_chain) {
    if (int r = abz_launch_mc_chain_start(ctx, delta, N, eps_target)) return r;
    ctx->mc_nabove_chain = ctx->mc_chain;
  }
  /* the rank pass: not once the host has SEEN the chain switch to rejection (until then it is launched and, if the device
   * has switched already, left unread) */
  const bool launch_rank = do_rank && !ctx->mc_reject_known;
  if (do_rank && !launch_rank) ctx->n_mc_reject_gens += 1;
  const bool need_window = lo_hi || !ctx->mc_window_ready || ctx->mc_alpha != alpha || ctx->mc_eps_target != eps_target;
  const unsigned long long* win = ctx->d_scal + ABZ_S_MCW_EPS;
  const unsigned long long* seq_dev = ctx->d_scal + ABZ_S_MCSEQ;
  const uint32_t sweep_base = sweep - (uint32_t)ctx->mc_issued;     /* the kernel adds the device's generation count back */
  const int slot = (int)(ctx->mc_issued % ABZ_MC_RING);
  const abz_rank_plan plan = abz_rank_plan_for(N, ctx->mc_tail_hint, ctx->mc_tail_bound);
  /* would the sweep of this generation carry an event pair (bench.py's kernel timing)?  Events cannot live in a graph. */
  const bool timed_now = ctx->timing && ctx->ev_tail - ctx->ev_head < ABZ_GROUP_MAX &&
                         !(ctx->timing_stride > 1 && (ctx->timing_seq % ctx->timing_stride) != 0);
  /* the body of one generation as stream launches (also what gets captured) */
  auto enqueue = [&]() -> int {
    if (launch_rank) {               /* mc:20-24 is only reached while some Ds[i] > eps */
      if (int r = abz_rank_prepare_impl(ctx, delta, N, 0.0, 0.0, order, sorted_delta, cnt, win, ctx->mc_tail_hint, ctx->mc_tail_bound)) return r;
    }
    if (int r = abz_launch_mc_swarm(ctx, order, cnt, (uint32_t)N, theta, logpi, delta, ntheta, nlogpi, ndelta, 0.0, eps_target, gamma0,
                                    gamma_sigma, (uint32_t)i0, (uint32_t)n_local, sweep_base, win, seq_dev, ctx->d_scal + ABZ_S_MC_NABOVE,
                                    launch_rank ? 1 : 0)) return r;
    if (sharded) {
      /* this rank's counts / extrema / fail word -> the exchange words; the new rows, log-priors, distances (and blob stamps) of
       * every rank's particles and those words travel as one group of collectives; the snapshot then folds GLOBAL values */
      if (int r = abz_launch_mc_partial(ctx, ctx->mm_bank)) return r;
      if (int r = abz_comm_mc_exchange(ctx, ntheta, nlogpi, ndelta, ctx->stamp_cur ? ctx->stamp_nxt : nullptr, n_local, ctx->h_model.ld)) return r;
    }
    return abz_launch_mc_snapshot(ctx, ctx->mm_bank, ctx->d_ring, alpha, eps_target, launch_rank ? ctx->mc_rank_state : nullptr,
                                  (uint32_t)N, sharded ? 1 : 0);
  };
  /* a sharded generation is a collective call: what can fail on this rank alone -- the workspace of the rank pass -- is obtained
   * before anything is enqueued (abz_comm.hip reserves its staging block before its first callback) */
  if (sharded && launch_rank) { if ((rc = abz_ws_reserve(ctx, plan.ws_bytes))) return rc; }
  const long long ev_before = ctx->ev_tail;
  bool replayed = false;
  if (ctx->graphs_on && ctx->stream != nullptr && !need_window && !timed_now && !sharded) {      /* the legacy default stream cannot be captured */
    abz_mc_graph_key key;
    memset(&key, 0, sizeof(key));
    key.theta = theta; key.logpi = logpi; key.delta = delta; key.ntheta = ntheta; key.nlogpi = nlogpi; key.ndelta = ndelta;
    key.order = order; key.sorted_delta = sorted_delta; key.cnt = cnt; key.stamp_cur = ctx->stamp_cur; key.stamp_nxt = ctx->stamp_nxt;
    key.stream = (const void*)ctx->stream; key.N = N; key.alpha = alpha; key.eps_target = eps_target; key.gamma0 = gamma0;
    key.gsig = gamma_sigma; key.sweep_base = sweep_base; key.do_rank = launch_rank ? 1 : 0;
    key.path = launch_rank ? (plan.small_path && plan.long_path ? 0 : plan.small_path ? 1 : 2) : -1;
    key.ltiles = launch_rank && plan.long_path ? plan.ltiles : 0u;
    key.mm_bank = ctx->mm_bank; key.L = ctx->L; key.C = ctx->C;
    if (launch_rank) { if ((rc = abz_ws_reserve(ctx, plan.ws_bytes))) return rc; }     /* no allocation inside a capture */
    key.ws = ctx->ws;
    abz_mc_graph* g = nullptr;
    for (abz_mc_graph& e : ctx->mc_graphs)
      if (memcmp(&e.key, &key, sizeof(key)) == 0) { g = &e; break; }
    if (!g) {
      if (ctx->mc_graphs.size() >= 64) {           /* a host that keeps changing its arguments: start over */
        ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));      /* a cached graph may still be executing */
        for (abz_mc_graph& e : ctx->mc_graphs) (void)hipGraphExecDestroy(e.exec);
        ctx->mc_graphs.clear();
      }
      const long long paths_before[3] = {ctx->n_rank_paths[0], ctx->n_rank_paths[1], ctx->n_rank_paths[2]};
      const long long tseq = ctx->timing_seq;
      hipGraph_t graph = nullptr;
      hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        const int r = enqueue();
        e = hipStreamEndCapture(ctx->stream, &graph);            /* always: the stream must leave capture mode */
        hipGraphExec_t exec = nullptr;
        if (r == 0 && e == hipSuccess && graph) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (r == 0 && e == hipSuccess && exec) {
          abz_mc_graph ng;
          ng.key = key; ng.exec = exec; ng.rank_state = launch_rank ? ctx->mc_rank_state : nullptr; ng.rank_limit = ctx->mc_rank_limit;
          ctx->mc_graphs.push_back(ng);
          g = &ctx->mc_graphs.back();
          ctx->n_graph_captures += 1;
        }
      }
      /* the capture ran the launchers' host-side bookkeeping once; the replay below does it again */
      for (int q = 0; q < 3; ++q) ctx->n_rank_paths[q] = paths_before[q];
      ctx->timing_seq = tseq;
      if (!g) { (void)hipGetLastError(); ctx->graphs_on = false; }     /* capture is not available here: stream launches from now on */
    }
    if (g) {
      ABZ_HIP_CHECK(hipGraphLaunch(g->exec, ctx->stream));
      if (launch_rank) { ctx->n_rank_paths[key.path] += 1; ctx->mc_rank_state = g->rank_state; ctx->mc_rank_limit = g->rank_limit; }
      if (ctx->timing && ctx->timing_stride > 1) ctx->timing_seq += 1;      /* what abz_time_begin would have counted */
      ctx->n_graph_replays += 1;
      replayed = true;
    }
  }
  if (!replayed) {
    ctx->n_graph_direct += 1;
    if (need_window) {
      rc = abz_launch_mc_window(ctx, lo_hi ? -1 : 1 - ctx->mm_bank, lo_hi ? lo_hi[0] : 0.0, lo_hi ? lo_hi[1] : 0.0, alpha, eps_target);
      if (rc) { ctx->mc_seq_dirty = true; if (sharded) abz_comm_abort_after_failure(ctx); return rc; }
    }   /* else: the snapshot kernel of the generation before has already made this generation's eps_pop and window */
    rc = enqueue();
    /* a sharded generation is a collective call: the peers are in (or on their way to) an exchange this rank will not make */
    if (rc) { ctx->mc_seq_dirty = true; if (sharded) abz_comm_abort_after_failure(ctx); return rc; }
  }
  ctx->ring_timed[slot] = ctx->ev_tail != ev_before;
  ctx->ring_folded[slot] = false;
  ctx->ring_chain[slot] = ctx->mc_chain; ctx->ring_eps_target[slot] = eps_target;
  ctx->mc_last_out = (const void*)ndelta; ctx->mc_last_N = N;
  ctx->mc_window_ready = true; ctx->mc_alpha = alpha; ctx->mc_eps_target = eps_target;
  ctx->mm_bank = 1 - ctx->mm_bank;   /* the kernel reset the other bank for the next sweep */
  ctx->mc_have_bank = !sharded;      /* (a sharded sweep leaves this rank's extrema only) */
  ctx->mc_chain_sharded = sharded;
  *ticket = (int64_t)ctx->mc_issued;
  ctx->mc_issued += 1;
  return 0;
}
extern "C" {
int abcdez_mc_generation_async(abcdez_ctx* ctx, int64_t N, const double* theta, const double* logpi, const double* delta,
                               double* ntheta, double* nlogpi, double* ndelta, uint32_t* order, double* sorted_delta,
                               uint32_t* cnt, double alpha, double eps_target, const double* lo_hi, int32_t do_rank,
                               double gamma0, double gamma_sigma, uint32_t sweep, int64_t* ticket) {
  return mc_generation_async_impl(ctx, N, theta, logpi, delta, ntheta, nlogpi, ndelta, order, sorted_delta, cnt, alpha, eps_target, lo_hi,
                                  do_rank, gamma0, gamma_sigma, sweep, ticket, false);
}
/* the same on a population sharded over the ranks of the context's communicator (abcdez_comm_init): include/abcdez_hip.h */
int abcdez_mc_generation_sharded_async(abcdez_ctx* ctx, int64_t N, const double* theta, const double* logpi, const double* delta,
                                       double* ntheta, double* nlogpi, double* ndelta, uint32_t* order, double* sorted_delta,
                                       uint32_t* cnt, double alpha, double eps_target, const double* lo_hi, int32_t do_rank,
                                       double gamma0, double gamma_sigma, uint32_t sweep, int64_t* ticket) {
  return mc_generation_async_impl(ctx, N, theta, logpi, delta, ntheta, nlogpi, ndelta, order, sorted_delta, cnt, alpha, eps_target, lo_hi,
                                  do_rank, gamma0, gamma_sigma, sweep, ticket, true);
}

/* Results of the generation `ticket` (the oldest one not yet redeemed): nsim, #(new Ds > eps_target) (mc:156), extrema of the
 * new distances (mc:146 of the next generation, mc:163), and the eps_pop it ran with.  Waits for THAT generation only. */
int abcdez_mc_generation_wait(abcdez_ctx* ctx, int64_t ticket, int64_t* nsim, int64_t* n_above_target, double* dmin,
                              double* dmax, double* eps_pop) {
  ABZ_REQUIRE(ctx && nsim, "mc_generation_wait: null argument");
  ABZ_REQUIRE(ticket == (int64_t)ctx->mc_waited && ticket < (int64_t)ctx->mc_issued,
              "mc_generation_wait: tickets are redeemed in the order they were issued");
  const int slot = (int)(ticket % ABZ_MC_RING);
  const volatile unsigned long long* snap = ctx->h_ring + (size_t)slot * ABZ_RING_WORDS;
  if (!ctx->ring_folded[slot]) {
    /* the snapshot kernel stores the ticket word last (system-scope release): poll it; a failed launch cannot hang us, and a
     * failure leaves the ring as it was (the ticket stays unredeemed) */
    const int rc = abz_poll_word(ctx, ctx->h_ring + (size_t)slot * ABZ_RING_WORDS + ABZ_RING_WORDS - 1, (unsigned long long)ticket + 1ull);
    if (rc < 0) return rc;
    if (rc == 1) { abz_set_error("mc_generation_wait: the stream drained without the generation's snapshot"); return -2; }
  }
  ring_fold(ctx, ticket);
  ctx->mc_waited += 1;
  if (snap[6] != 0ull && ctx->ring_timed[slot] && ctx->ev_head < ctx->ev_tail) {
    ctx->ev_head += 1;               /* the failed generation's timing pair leaves the FIFO with it: later tickets keep their own pairs */
    ctx->ring_timed[slot] = false;
  }
  if (snap[6] == 2ull) {
    /* the sweep drew its better particles by rejection and a particle found none in 1024 trials: with at least 1 / 16 of the
     * population in every candidate set that does not happen -- the distances are not the ones the chain's count was made of */
    abz_population_written(ctx);
    abz_set_error("mc_generation_wait: a particle drawing its better particle by rejection ran out of trials: the distances were "
                  "written outside the library without abcdez_smc_select_discard; the generation's results are invalid");
    return -3;
  }
  if (snap[6] != 0ull) {
    /* only the one-workgroup sort was launched for this generation (a proved tail bound <= 4096) and the tail was longer: the
     * distances were written behind the library's back.  The generation's sweep drew its better particles from a stale
     * enumeration; nothing after it can be trusted.  (abcdez_population_written / abcdez_smc_select_discard tell the library.) */
    abz_population_written(ctx);
    abz_set_error("mc_generation_wait: the rank pass of this generation was launched for a tail bound that no longer held "
                  "(the distances were written outside the library without abcdez_smc_select_discard); its results are invalid");
    return -3;
  }
  *nsim = (int64_t)ctx->ring_res[slot][0];
  if (n_above_target) *n_above_target = (int64_t)ctx->ring_res[slot][1];
  if (dmin) *dmin = f64_from_order_key_host(snap[2]);
  if (dmax) *dmax = f64_from_order_key_host(snap[3]);
  if (eps_pop) { const unsigned long long e = snap[4]; memcpy(eps_pop, &e, 8); }
  if (!ctx->timing || !ctx->ring_timed[slot]) return 0;
  ABZ_HIP_CHECK(hipEventSynchronize(ctx->ev[2 * (int)(ctx->ev_head % ABZ_GROUP_MAX) + 1]));   /* the sweep's own end event */
  return timing_consume(ctx, 1, -1);
}

int abcdez_push_p(abcdez_ctx* ctx, const double* theta, int64_t N, double* out) {
  ABZ_REQUIRE(ctx && theta && out, "push_p: null argument");
  ABZ_REQUIRE(N >= 0 && N <= ABZ_MAX_N, "push_p: N out of range");
  return abz_launch_push_p(ctx, theta, N, out);
}

int abcdez_math_eval(abcdez_ctx* ctx, int fn, const double* x, double* y, double* y2, int64_t n) {
  ABZ_REQUIRE(ctx && x && y, "math_eval: null argument");
  ABZ_REQUIRE(fn >= 0 && fn <= 11 && (y2 || (fn != 2 && fn != 6 && fn != 8 && fn != 11)), "math_eval: bad function id / missing second array");
  return abz_math_eval_impl(ctx, fn, x, y, y2, n);
}

int abcdez_draws_eval(abcdez_ctx* ctx, int lanes, int64_t i0, int64_t n, int64_t n_pool, uint32_t sweep, double gamma0,
                      double gamma_sigma, uint32_t* ra, uint32_t* rb, double* gamma, double* log_u) {
  ABZ_REQUIRE(ctx && ra && rb && gamma && log_u, "draws_eval: null argument");
  ABZ_REQUIRE(i0 >= 0 && n >= 0 && n_pool >= 3 && i0 + n <= n_pool && n_pool <= ABZ_MAX_N, "draws_eval: range out of bounds");
  return abz_draws_eval_impl(ctx, lanes, (uint32_t)i0, (uint32_t)n, (uint32_t)n_pool, sweep, gamma0, gamma_sigma, ra, rb,
                             gamma, log_u);
}

} /* extern "C" */
