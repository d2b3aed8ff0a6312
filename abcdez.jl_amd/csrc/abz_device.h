/*
 * abz_device.h -- device-side building blocks shared by the gfx950 kernels.
 *
 * Thread mapping ("lane group"): L consecutive lanes of a wavefront own one
 * particle; theta is row-major double[N][ld] with ld = L*C.  Lane j of the group
 * owns the components k(m,c) = m*2L + 2j + c (m < C/2, c < 2), i.e. load m of the
 * group is one contiguous run of L*16 bytes -- a whole 128-B line for L = 8.  Own
 * rows are therefore read fully coalesced and the two DE donor rows, which are
 * uniformly random rows of the table, are fetched as whole contiguous rows.
 * Per-particle sums follow the canonical pairwise tree of abcdez_spec.h: level 0
 * inside the lane, then an xor butterfly over the L lanes, then a binary tree over m.
 */
#ifndef ABZ_DEVICE_H
#define ABZ_DEVICE_H

#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "abcdez_spec.h"
#include "abz_hotmodel.h"

#define ABZ_BLOCK 256

/* ---- component ownership ------------------------------------------------------- */
template <int L, int C>
struct Lay {
  static constexpr int M = (C + 1) / 2;              /* 16-byte loads per row per lane */
  static constexpr int LD = L * C;
  __device__ static inline int comp(int j, int m, int c) { return C == 1 ? 0 : m * 2 * L + 2 * j + c; }
};

template <int L, int C>
__device__ inline void load_row(const double* __restrict__ row, int j, double (&v)[C]) {
  if constexpr (C == 1) {
    v[0] = row[0];
  } else {
#pragma unroll
    for (int m = 0; m < C / 2; ++m) {
      const double2 t = *reinterpret_cast<const double2*>(row + m * 2 * L + 2 * j);
      v[2 * m] = t.x; v[2 * m + 1] = t.y;
    }
  }
}
template <int L, int C>
__device__ inline void store_row(double* __restrict__ row, int j, const double (&v)[C]) {
  if constexpr (C == 1) {
    row[0] = v[0];
  } else {
#pragma unroll
    for (int m = 0; m < C / 2; ++m) {
      double2 t; t.x = v[2 * m]; t.y = v[2 * m + 1];
      *reinterpret_cast<double2*>(row + m * 2 * L + 2 * j) = t;
    }
  }
}

/* the same row through non-temporal stores: for rows nobody reads again soon */
template <int L, int C>
__device__ inline void store_row_nt(double* __restrict__ row, int j, const double (&v)[C]) {
  if constexpr (C == 1) {
    __builtin_nontemporal_store(v[0], row);
  } else {
    typedef double d2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int m = 0; m < C / 2; ++m) {
      d2v t; t.x = v[2 * m]; t.y = v[2 * m + 1];
      __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(row + m * 2 * L + 2 * j));
    }
  }
}

/* The read-mostly model tables (prior descriptors, data vector) are staged in LDS once per
 * workgroup: every group of a block reads the same ld entries, so this turns ~10 dependent
 * global loads per component into broadcast LDS reads and keeps them out of the VGPR budget. */
template <int LD>
struct ModelLds {
  abz_tables tab;              /* log / sincos tables of the sampler (8 KB) */
  abz_prior_dim prior[LD];
  double y[LD];
};

/* two phases so the global loads can be issued early (before the wave's dependent loads)
 * and the LDS writes + barrier placed right before the first use */
template <int SIM, int LD, int BLOCK = ABZ_BLOCK>
struct ModelStage {
  static constexpr int W = LD * (int)(sizeof(abz_prior_dim) / 8);
  static constexpr int NW = (W + BLOCK - 1) / BLOCK;
  static constexpr int NU = (int)(sizeof(abz_tables) / 16);                /* 16-byte pieces of the tables (7 KB: 448) */
  static constexpr int NT = (NU + BLOCK - 1) / BLOCK;              /* ... per thread; the last round is partial */
  static_assert(sizeof(abz_tables) % 16 == 0, "the tables are staged in 16-byte pieces");
  uint64_t w[NW];
  double2 tb[NT];
  double y;
  __device__ inline void load(const HotModel& M) {
    const double2* __restrict__ tsrc = reinterpret_cast<const double2*>(M.tables);
#pragma unroll
    for (int q = 0; q < NT; ++q) {      /* every element is assigned (threads past the end re-read piece 0): a conditionally written
                                         * array stays in scratch memory -- 16 bytes per lane out to HBM and back, measured as +128 B
                                         * of fabric traffic per update (profiles/HISTORY.md, round 5) */
      const int t = threadIdx.x + q * BLOCK;
      tb[q] = tsrc[(NU % BLOCK == 0 || t < NU) ? t : 0];
    }
    const uint64_t* __restrict__ src = reinterpret_cast<const uint64_t*>(M.prior);
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int t = threadIdx.x + q * BLOCK;
      w[q] = t < W ? src[t] : 0ull;
    }
    y = 0.0;
    if constexpr (SIM == ABZ_SIM_MVN) {
      if ((int)threadIdx.x < LD && (int)threadIdx.x < M.d) y = M.data[threadIdx.x];
    } else if constexpr (SIM == ABZ_SIM_USER) {       /* the first ld entries of the user's data: abz_user_lanes::y */
      if ((int)threadIdx.x < LD && (int)threadIdx.x < M.n_data) y = M.data[threadIdx.x];
    }
  }
  __device__ inline void store(ModelLds<LD>& s) const {
    double2* tdst = reinterpret_cast<double2*>(&s.tab);
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int t = threadIdx.x + q * BLOCK;
      if (NU % BLOCK == 0 || t < NU) tdst[t] = tb[q];
    }
    uint64_t* dst = reinterpret_cast<uint64_t*>(s.prior);
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int t = threadIdx.x + q * BLOCK;
      if (t < W) dst[t] = w[q];
    }
    if ((int)threadIdx.x < LD) s.y[threadIdx.x] = y;
  }
};

/* the sampler tables alone (kernels that draw but neither evaluate the prior nor simulate) */
struct TabStage {
  static constexpr int NU = (int)(sizeof(abz_tables) / 16);
  static constexpr int NT = (NU + ABZ_BLOCK - 1) / ABZ_BLOCK;
  double2 tb[NT];
  __device__ inline void load(const HotModel& M) {
    const double2* __restrict__ tsrc = reinterpret_cast<const double2*>(M.tables);
#pragma unroll
    for (int q = 0; q < NT; ++q) {      /* every element is assigned (threads past the end re-read piece 0): a conditionally written
                                         * array stays in scratch memory -- 16 bytes per lane out to HBM and back, measured as +128 B
                                         * of fabric traffic per update (profiles/HISTORY.md, round 5) */
      const int t = threadIdx.x + q * ABZ_BLOCK;
      tb[q] = tsrc[(NU % ABZ_BLOCK == 0 || t < NU) ? t : 0];
    }
  }
  __device__ inline void store(abz_tables& s) const {
    double2* tdst = reinterpret_cast<double2*>(&s);
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int t = threadIdx.x + q * ABZ_BLOCK;
      if (NU % ABZ_BLOCK == 0 || t < NU) tdst[t] = tb[q];
    }
  }
};

/* kernels whose workgroups loop over tiles of ABZ_BLOCK threads (init, abcdemc sweep, blobs): tiles dealt round-robin to the
 * workgroups the launcher sized to what is resident at once (abz_persistent_grid), tables staged once per workgroup */
#define ABZ_TILE_LOOP(tile, ntiles) for (uint32_t tile = blockIdx.x; tile < (ntiles); tile += gridDim.x)

/* ---- abz_kernel_logpdf (abcdez_spec.h, types.jl:26-73) as the sweep evaluates it: the indicator kernels (the default, every
 * BASELINE configuration) are two compares; the Epanechnikov kernels' log(1 - (x/eps)^2) is kept OUT OF LINE so that its
 * dozen polynomial constants do not sit in registers across the tile loop of a sweep that never uses them.  Same function. */
__device__ __attribute__((noinline)) inline double kernel_logpdf_epa(double eps, double x) {
  const double t = x / eps;
  return abz_log(1.0 - t * t);
}
__device__ inline double kernel_logpdf_dev(int kind, double eps, double x) {
  if (!abz_kernel_insupport(kind, eps, x)) return ABZ_NINF;
  if (kind < ABZ_K_EPA) return 0.0;
  return kernel_logpdf_epa(eps, x);
}

/* ---- canonical per-particle tree sum ------------------------------------------- */
__device__ inline double shfl_xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }

template <int L, int C>
__device__ inline double group_tree_sum(const double (&x)[C]) {
  if constexpr (C == 1) {
    return x[0];
  } else {
    constexpr int M = C / 2;
    double s[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      double t = x[2 * m] + x[2 * m + 1];            /* level 0: (2t, 2t+1)            */
#pragma unroll
      for (int off = 1; off < L; off <<= 1) t = t + shfl_xor_f64(t, off);   /* over lanes */
      s[m] = t;
    }
#pragma unroll
    for (int st = 1; st < M; st <<= 1)               /* over m                         */
#pragma unroll
      for (int m = 0; m + st < M; m += 2 * st) s[m] = s[m] + s[m + st];
    return s[0];
  }
}

/* ---- correlated Normal prior (abcdez_spec.h, abz_model.mv): out_k = sum_{m <= k} mat[k][m] in_m for the lane's components k,
 * accumulated left to right (m ascending) with fma exactly as abz_mv_whiten1 / abz_mv_forward1 do.  The group's vector is spread
 * over its L lanes; component m = mm 2L + 2 jl + b lives in register 2 mm + b of lane jl, so walking (mm, jl, b) in that order
 * visits m in ascending order with compile-time register indices and one lane broadcast per component (ds_bpermute; a group sits
 * inside one wavefront).  No LDS array: the kernels of priors that are not correlated carry the branch but pay nothing for it
 * (round 3 exchanged the vector through a static 256 C 8-byte LDS row that every non-plain kernel was charged for). */
template <int L, int C>
__device__ inline void group_lower_matvec(const double* __restrict__ mat, int j, const double (&in)[C], double (&out)[C]) {
  constexpr int LD = L * C;
#pragma unroll
  for (int q = 0; q < C; ++q) out[q] = 0.0;
  const int lane0 = (int)(threadIdx.x & 63u) - j;          /* first lane of the group */
#pragma unroll
  for (int mm = 0; mm < (C + 1) / 2; ++mm) {
#pragma unroll 1
    for (int jl = 0; jl < L; ++jl) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (2 * mm + b >= C) continue;
        const int m = Lay<L, C>::comp(jl, mm, b);
        const double xm = L == 1 ? in[2 * mm + b] : __shfl(in[2 * mm + b], lane0 + jl, 64);
#pragma unroll
        for (int q = 0; q < C; ++q) {
          const int k = Lay<L, C>::comp(j, q / 2, q & 1);
          const double t = abz_fma(mat[(size_t)k * LD + m], xm, out[q]);
          out[q] = m <= k ? t : out[q];
        }
      }
    }
  }
}

/* ---- push_p + log prior of the lane's components (priors.jl:40-46, types.jl:20-23) */
/* PLAIN: every real dimension is a continuous Normal and the padding descriptors are all-zero (abcdez_ctx_create checks):
 * push_p is the identity and the log-density is abz_prior_logpdf1's Normal branch without the family dispatch -- the
 * same operations, so the same bits (tests/test_gpu_packed.py compares both against the oracle's generic evaluation) */
/* mv (generic instantiation only): the maps of a correlated Normal prior -- the per-dimension log-densities are then taken of
 * the whitened components z = W (theta - mu) (abcdez_spec.h) */
template <int L, int C, bool PLAIN = false>
__device__ inline double group_logprior(const abz_prior_dim* pd /* LDS, ld entries */, int j, const double (&p)[C],
                                        double (&pp)[C], const double* __restrict__ mv = nullptr,
                                        const double* __restrict__ ext = nullptr /* records of the wrapper families, global memory */) {
  double lp[C];
  if constexpr (!PLAIN) {
    if (mv) {
      constexpr int LD = L * C;
      double df[C], z[C];
#pragma unroll
      for (int q = 0; q < C; ++q) {
        pp[q] = p[q];                                                    /* continuous: push_p is the identity */
        df[q] = p[q] - mv[Lay<L, C>::comp(j, q / 2, q & 1)];
      }
      group_lower_matvec<L, C>(mv + LD, j, df, z);
#pragma unroll
      for (int q = 0; q < C; ++q) lp[q] = abz_prior_logpdf1(&pd[Lay<L, C>::comp(j, q / 2, q & 1)], z[q]);
      return group_tree_sum<L, C>(lp);
    }
  }
#pragma unroll
  for (int q = 0; q < C; ++q) {
    const abz_prior_dim* d = &pd[Lay<L, C>::comp(j, q / 2, q & 1)];
    if constexpr (PLAIN) {
      pp[q] = p[q];
      const double z = (p[q] - d->p0) * d->c1;
      lp[q] = abz_fma(-0.5 * z, z, d->c0);
    } else {
      pp[q] = abz_push_p(d, p[q]);
      lp[q] = abz_prior_logpdf1x(d, pp[q], ext);
    }
  }
  return group_tree_sum<L, C>(lp);
}

/* ---- user-supplied simulator (ABZ_SIM_USER): compiled at run time with hiprtc together with these
 * headers (abz_jit.hip).  The user source defines abz_user_dist; randomness comes from abz_user_rng,
 * which hands out consecutive Philox blocks of the particle's (index, epoch, purpose) stream, so a
 * user simulator is as reproducible and launch-shape independent as the built-in ones.            */
struct abz_user_rng {
  uint64_t seed;
  uint32_t i, epoch, purpose, sub;
  const abz_tables* T;
  __device__ inline abz_u64x2 block() { return abz_rng(seed, i, epoch, sub++, purpose); }
  __device__ inline double uniform() { return abz_u01_co(block().w0); }              /* [0, 1)          */
  __device__ inline uint64_t bits() { return block().w0; }
  __device__ inline void normal_pair(double& z0, double& z1) { abz_normal_pair(block(), T, &z0, &z1); }
  __device__ inline double normal() { double a, b; normal_pair(a, b); return a; }
  /* the same draws ADDRESSED by a sub-index instead of taken in sequence: what a cooperative simulator (below) uses, keyed by
   * the component it draws for, so that its results do not depend on how many lanes share the row */
  __device__ inline abz_u64x2 block_at(uint32_t s) const { return abz_rng(seed, i, epoch, s, purpose); }
  __device__ inline double uniform_at(uint32_t s) const { return abz_u01_co(block_at(s).w0); }
  __device__ inline void normal_pair_at(uint32_t s, double& z0, double& z1) const { abz_normal_pair(block_at(s), T, &z0, &z1); }
};
/* one thread sees the whole row: length(prior) <= 16 */
__device__ double abz_user_dist(const double* theta, int d, const double* data, int n_data, const double* sim_p,
                                abz_user_rng& rng);
/* COOPERATIVE form, rows of 17 to 256 parameters: L lanes of a wavefront own one particle, as in the built-in d-dimensional Normal
 * simulator.  Every lane of the group calls
 *     abz_user_dist_lanes(theta, g, d, data, n_data, sim_p, rng)
 * with ITS C = ABZ_USER_C components in theta[0 .. C) (push_p-cast); g.comp(q) is the index in the row of theta[q] (indices >= d are
 * padding: theta = 0), g.y[k] is data[k] for k < ld read from LDS instead of global memory, g.sum(v) adds v[0 .. C) over all L x C
 * entries of the group in one canonical tree -- the same value on every lane and for every L -- and the function returns the
 * distance, the same value on every lane.  Draws are addressed
 * (rng.normal_pair_at(k, ...), rng.uniform_at(k)): key them by component (g.comp(q) / 2 for a pair), never by lane.  ABZ_USER_L and
 * ABZ_USER_C are compile-time constants of the translation unit. */
struct abz_user_lanes {
  int L, C, j;
  const double* y;        /* the first ld entries of `data` (zero beyond n_data), staged in LDS by the library: y[comp(q)] is the datum that
                           * belongs to theta[q] when the data vector is indexed like the parameters */
  __device__ inline int comp(int q) const { return C == 1 ? 0 : (q >> 1) * 2 * L + 2 * j + (q & 1); }
  __device__ inline double sum(const double* v) const;
};
__device__ double abz_user_dist_lanes(const double* theta, const abz_user_lanes& g, int d, const double* data, int n_data,
                                      const double* sim_p, abz_user_rng& rng);
#if defined(ABZ_USER_L) && defined(ABZ_USER_C)     /* the run-time translation unit of a user simulator (abz_jit.hip) */
__device__ inline double abz_user_lanes::sum(const double* v) const {
  double x[ABZ_USER_C];
#pragma unroll
  for (int q = 0; q < ABZ_USER_C; ++q) x[q] = v[q];
  return group_tree_sum<ABZ_USER_L, ABZ_USER_C>(x);
}
#endif

/* ---- simulators = dist!(theta, ve); arithmetic fixed by abcdez_spec.h (ABZ_SIM_*) -- */
/* BLOB = true additionally writes the simulated data behind the distance to blob[] (this lane's C entries in
 * row layout for L > 1, the whole blob for L = 1); only abz_blob_eval instantiates it -- the sweeps never do. */
__device__ void abz_user_blob(const double* theta, int d, const double* data, int n_data, const double* sim_p,
                              abz_user_rng& rng, double* blob, int n_blob);

/* ---- Lotka-Volterra (ABZ_SIM_LV; BASELINE.json configs[3]): one observation and one interval of RK4 steps, shared by sim_dist and by
 * the round-by-round second phase of the two-phase sweep (abz_kernels.h) -- the same operations in the same order either way */
struct LvConst { double h, h2, h6, sn; int steps, nobs; };
__device__ inline LvConst lv_const(const HotModel& M) {
  LvConst k;
  k.h = M.sim_p[2]; k.h2 = 0.5 * k.h; k.h6 = k.h / 6.0; k.sn = M.sim_p[4];
  /* the trip count as an integer kernel argument (scalar register): taken from the f64 parameter it would be born in a vector
   * register -- there is no scalar f64 -> i32 conversion -- and the compiler would keep the loop counter there, two of the loop's
   * 32 vector instructions */
  k.steps = M.sim_i[0]; k.nobs = M.n_data / 2;
  return k;
}
/* observation jo: (x, y) + N(0, noise^2) against data[2 jo], data[2 jo + 1]; the squared errors join the running sum */
template <bool BLOB>
__device__ inline void lv_observe(const HotModel& M, const abz_tables* T, const LvConst& k, uint32_t i, uint32_t epoch, uint32_t purpose,
                                  int jo, double x, double y, double& acc, double* blob) {
  double z0, z1;
  abz_normal_pair(abz_rng(M.seed, i, epoch, (uint32_t)jo, purpose), T, &z0, &z1);
  const double ox = abz_fma(k.sn, z0, x), oy = abz_fma(k.sn, z1, y);
  if constexpr (BLOB) { if (2 * jo + 1 < ABZ_MAX_BLOB) { blob[2 * jo] = ox; blob[2 * jo + 1] = oy; } }
  const double ex = ox - M.data[2 * jo];
  const double ey = oy - M.data[2 * jo + 1];
  acc = abz_fma(ex, ex, acc);
  acc = abz_fma(ey, ey, acc);
}
/* the RK4 steps between two observations */
__device__ inline void lv_advance(const LvConst& k, double a, double b, double c, double e, double& x, double& y) {
  const double h = k.h, h2 = k.h2, h6 = k.h6;
  for (int s = 0; s < k.steps; ++s) {
    const double k1x = x * abz_fma(-b, y, a), k1y = y * abz_fma(e, x, -c);
    const double xa = abz_fma(h2, k1x, x), ya = abz_fma(h2, k1y, y);
    const double k2x = xa * abz_fma(-b, ya, a), k2y = ya * abz_fma(e, xa, -c);
    const double xb = abz_fma(h2, k2x, x), yb = abz_fma(h2, k2y, y);
    const double k3x = xb * abz_fma(-b, yb, a), k3y = yb * abz_fma(e, xb, -c);
    const double xc = abz_fma(h, k3x, x), yc = abz_fma(h, k3y, y);
    const double k4x = xc * abz_fma(-b, yc, a), k4y = yc * abz_fma(e, xc, -c);
    x = abz_fma(h6, abz_fma(2.0, k2x, k1x) + abz_fma(2.0, k3x, k4x), x);   /* 2 k is exact: the same bits as (k1 + 2 k2) + (2 k3 + k4) */
    y = abz_fma(h6, abz_fma(2.0, k2y, k1y) + abz_fma(2.0, k3y, k4y), y);
  }
}

/* FULL: d == ld (no padding components), so the MVN simulator skips its `k < d` selects -- same values */
template <int SIM, int L, int C, bool BLOB = false, bool FULL = false>
__device__ inline double sim_dist(const HotModel& M, const abz_tables* T, int j, const double (&th)[C],
                                  const double* y /* LDS, ld */, uint32_t i, uint32_t epoch, uint32_t purpose,
                                  double* blob = nullptr) {
  const uint64_t seed = M.seed;
  if constexpr (SIM == ABZ_SIM_NORMAL1D) {
    double z0, z1;
    abz_normal_pair(abz_rng(seed, i, epoch, 0, purpose), T, &z0, &z1);
    const double x = abz_fma(M.sim_p[0], z0, th[0]);
    if constexpr (BLOB) blob[0] = x;
    return __builtin_fabs(x - M.data[0]);
  } else if constexpr (SIM == ABZ_SIM_MVN) {
    const double sg = M.sim_p[0];
    const int d = M.d;
    double sq[C];
    if constexpr (C == 1) {
      double z0, z1;
      abz_normal_pair(abz_rng(seed, i, epoch, 0, purpose), T, &z0, &z1);
      const double x = abz_fma(sg, z0, th[0]);
      if constexpr (BLOB) blob[0] = x;
      const double e = x - y[0];
      sq[0] = e * e;
    } else {
#pragma unroll
      for (int m = 0; m < C / 2; ++m) {
        double z[2];
        abz_normal_pair(abz_rng(seed, i, epoch, (uint32_t)(m * L + j), purpose), T, &z[0], &z[1]);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int k = Lay<L, C>::comp(j, m, c);
          double v = 0.0;
          if constexpr (BLOB) blob[2 * m + c] = 0.0;
          if (FULL || k < d) {
            const double x = abz_fma(sg, z[c], th[2 * m + c]);
            if constexpr (BLOB) blob[2 * m + c] = x;
            const double e = x - y[k];
            v = e * e;
          }
          sq[2 * m + c] = v;
        }
      }
    }
    return abz_sqrt(group_tree_sum<L, C>(sq));
  } else if constexpr (SIM == ABZ_SIM_DIRAC) {
    const double x = th[0] * th[0] + 1.0;
    if constexpr (BLOB) blob[0] = x;
    return __builtin_fabs(x - M.sim_p[0]);
  } else if constexpr (SIM == ABZ_SIM_QUAD2D) {
    double n1, n2;
    abz_normal_pair(abz_rng(seed, i, epoch, 0, purpose), T, &n1, &n2);
    const double u = abz_u01_co(abz_rng(seed, i, epoch, 1, purpose).w0);
    const double a = (th[0] + n1 * 0.01) - th[1] * th[1];
    const double b = (th[1] - 1.0) + n2 * 0.01;
    const double r = 50.0 * (a * a) + b * b;
    if constexpr (BLOB) { blob[0] = a; blob[1] = b; }
    return (u < M.sim_p[0]) ? ABZ_INF : r;
  } else if constexpr (SIM == ABZ_SIM_MIXTURE) {
    double n1, n2;
    abz_normal_pair(abz_rng(seed, i, epoch, 0, purpose), T, &n1, &n2);
    const uint64_t coin = abz_rng(seed, i, epoch, 1, purpose).w0 >> 63;
    const double x = th[0] + (coin ? n2 : n1 * 0.1);
    if constexpr (BLOB) blob[0] = x;
    return __builtin_fabs(x - M.sim_p[0]);
  } else if constexpr (SIM == ABZ_SIM_NORMDU) {
    double n1, n2;
    abz_normal_pair(abz_rng(seed, i, epoch, 0, purpose), T, &n1, &n2);
    const double x = (th[0] * th[0] + th[1]) * (th[0] + n1 * 0.01);
    if constexpr (BLOB) blob[0] = x;
    return __builtin_fabs(x - M.sim_p[0]);
  } else if constexpr (SIM == ABZ_SIM_WIENER) {
    const double f = 0.95 + 0.1 * abz_u01_co(abz_rng(seed, i, epoch, 0, purpose).w0);
    double acc = 0.0;
    const int n = M.n_data;
    for (int t = 0; t < n; ++t) {
      const double dt = (double)t;
      const double v = abz_sqrt(th[0] * th[0] * dt * dt + th[1] * th[1] * dt) * f;
      if constexpr (BLOB) { if (t < ABZ_MAX_BLOB) blob[t] = v; }
      acc += __builtin_fabs(v - M.data[t]);
    }
    return acc / (double)n;
  } else if constexpr (SIM == ABZ_SIM_LV) {
    const LvConst k = lv_const(M);
    double x = M.sim_p[0], y = M.sim_p[1];
    double acc = 0.0;
    for (int jo = 0; jo < k.nobs; ++jo) {
      lv_observe<BLOB>(M, T, k, i, epoch, purpose, jo, x, y, acc, blob);
      if (jo + 1 == k.nobs) break;
      lv_advance(k, th[0], th[1], th[2], th[3], x, y);
    }
    return abz_sqrt(acc);
  } else if constexpr (SIM == ABZ_SIM_USER) {
    if constexpr (L > 1) {       /* rows of 17 .. 64 parameters: the cooperative form, the row spread over the group's lanes */
      static_assert(!BLOB, "user simulators on lane groups carry no blobs");
      abz_user_rng rng{seed, i, epoch, purpose, 0u, T};
      const abz_user_lanes g{L, C, j, y};
      return abz_user_dist_lanes(th, g, M.d, M.data, M.n_data, M.sim_p, rng);
    } else {
    abz_user_rng rng{seed, i, epoch, purpose, 0u, T};
#ifdef ABZ_USER_HAS_BLOB
    if constexpr (BLOB) {       /* the user's blob function re-runs the simulation on a fresh copy of the same stream */
      abz_user_rng rng2{seed, i, epoch, purpose, 0u, T};
      abz_user_blob(th, M.d, M.data, M.n_data, M.sim_p, rng2, blob, M.n_blob);
    }
#endif
    return abz_user_dist(th, M.d, M.data, M.n_data, M.sim_p, rng);
    }
  } else if constexpr (SIM == ABZ_SIM_SOCKS) {
    double ns = th[0];
    if (!(ns >= 0.0)) return ABZ_NAN;
    if (ns > 2147483647.0) ns = 2147483647.0;
    const uint32_t n_socks = (uint32_t)ns;
    const uint32_t n_pairs = (uint32_t)abz_rint(th[1] * abz_floor((double)n_socks * 0.5));
    const uint32_t n_want = (uint32_t)M.sim_p[2];
    const uint32_t m = n_socks < n_want ? n_socks : n_want;
    uint32_t pos[16];
    for (uint32_t t = 0; t < m; ++t) {
      const abz_u64x2 w = abz_rng(seed, i, epoch, t >> 1, purpose);
      uint32_t j = abz_randint((t & 1) ? w.w1 : w.w0, n_socks - t);
      uint32_t at = 0;
      while (at < t && j >= pos[at]) { ++j; ++at; }
      for (uint32_t q = t; q > at; --q) pos[q] = pos[q - 1];
      pos[at] = j;
    }
    uint32_t uniq = 0;
    for (uint32_t t = 0; t < m; ++t) {
      const uint32_t id = pos[t] < 2 * n_pairs ? pos[t] >> 1 : pos[t] - n_pairs;
      const uint32_t idp = t ? (pos[t - 1] < 2 * n_pairs ? pos[t - 1] >> 1 : pos[t - 1] - n_pairs) : 0xFFFFFFFFu;
      uniq += (t == 0) || (id != idp);
    }
    const double pairs = (double)(m - uniq), odds = (double)uniq - (double)(m - uniq);
    if constexpr (BLOB) { blob[0] = pairs; blob[1] = odds; }
    return __builtin_fabs(pairs - M.sim_p[0]) + __builtin_fabs(odds - M.sim_p[1]);
  } else {
    return ABZ_NAN;
  }
}

/* ---- per-particle scalar draws of one sweep: donor ranks, gamma jitter, log(accept uniform).
 * With L >= 4 lanes per particle the three Philox blocks are evaluated by lanes 0, 1, 2 of
 * the group in ONE pass of the instruction stream (the purpose tag is the only difference),
 * the Box-Muller radius of the jitter and the accept test share one log evaluation, and the
 * results are broadcast inside the group.  Same values as the straightforward evaluation. */
/* In two steps, so that a kernel can issue the donors' loads before the sampler tables are staged: words() is pure
 * integer work (Philox + the donor ranks), finish() the table-driven part (jitter normal, log of the accept uniform). */
template <int L>
struct ParticleDraws {
  abz_u64x2 w, wa;        /* L >= 4: w = this lane's block (donor | jitter | accept); L < 4: w = jitter block, wa = accept block */
  __device__ inline void words(uint64_t seed, uint32_t i, uint32_t sweep, int j, uint32_t n_pool, uint32_t ri, uint32_t* ra,
                               uint32_t* rb) {
    if constexpr (L >= 4) {
      const uint32_t purpose = j == 0 ? (uint32_t)ABZ_RNG_DONOR : (j == 1 ? (uint32_t)ABZ_RNG_JITTER : (uint32_t)ABZ_RNG_ACCEPT);
      w = abz_rng(seed, i, sweep, 0, purpose);
      uint32_t a_, b_;
      abz_donor_ranks(w, n_pool, ri, &a_, &b_);                 /* meaningful on lane 0 */
      *ra = __shfl(a_, 0, L);
      *rb = __shfl(b_, 0, L);
    } else {
      abz_donor_ranks(abz_rng(seed, i, sweep, 0, ABZ_RNG_DONOR), n_pool, ri, ra, rb);
      w = abz_rng(seed, i, sweep, 0, ABZ_RNG_JITTER);
      wa = abz_rng(seed, i, sweep, 0, ABZ_RNG_ACCEPT);
    }
  }
  __device__ inline void finish(const abz_tables* T, double gamma0, double gsig, double* g, double* log_u) const {
    if constexpr (L >= 4) {
      const double lg = abz_log_tab(abz_u01_open(w.w0), T);     /* lane 1: BM radius, lane 2: accept */
      double sn, cs;
      abz_sincos2pi_tab_w(w.w1, T, &sn, &cs);
      const double z0 = abz_sqrt_pn(-2.0 * lg) * cs;            /* meaningful on lane 1 */
      const double g_ = gamma0 * (1.0 + z0 * gsig);
      *g = __shfl(g_, 1, L);
      *log_u = __shfl(lg, 2, L);
    } else {
      double z0, z1;
      abz_normal_pair(w, T, &z0, &z1);
      *g = gamma0 * (1.0 + z0 * gsig);
      /* the SAME function as the L >= 4 branch and the oracle (abz_log_tab): the accept variate must not
       * depend on the lane shape -- the polynomial abz_log differs from it in ~19 % of arguments by an ulp */
      *log_u = abz_log_tab(abz_u01_open(wa.w0), T);
    }
  }
};
template <int L>
__device__ inline void particle_draws(const abz_tables* T, uint64_t seed, uint32_t i, uint32_t sweep, int j,
                                      uint32_t n_pool, uint32_t ri,
                                      double gamma0, double gsig, uint32_t* ra, uint32_t* rb, double* g,
                                      double* log_u) {
  ParticleDraws<L> d;
  d.words(seed, i, sweep, j, n_pool, ri, ra, rb);
  d.finish(T, gamma0, gsig, g, log_u);
}

/* ---- block-level integer counters: every thread counts over its tiles in two registers; at the end of the block a wave
 * reduction -> LDS -> two agent-scope atomic ADDS per block into one of ABZ_CSLOTS cumulative slots (abz_hotmodel.h).
 * Same-address atomics serialise (10^5 of them on ONE address per launch were the bottleneck of the first build, ~15 ns
 * each); spread over 256 lines they overlap the kernel, and because the slots are cumulative nothing has to be zeroed or
 * reduced by another launch.  Ends with no barrier pending; must be reached by every thread of the block.       */
template <int BLOCK = ABZ_BLOCK>
__device__ inline void block_count2(unsigned int x, unsigned int y, unsigned long long* __restrict__ cslots, uint32_t cls) {
  __shared__ unsigned int s_cnt[2][BLOCK / 64];
#pragma unroll
  for (int off = 32; off; off >>= 1) { x += __shfl_xor(x, off, 64); y += __shfl_xor(y, off, 64); }
  if ((threadIdx.x & 63) == 0) { s_cnt[0][threadIdx.x >> 6] = x; s_cnt[1][threadIdx.x >> 6] = y; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long xs = 0, ys = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) { xs += s_cnt[0][w]; ys += s_cnt[1][w]; }
    unsigned long long* s = cslots + (size_t)(blockIdx.x & (ABZ_CSLOTS - 1)) * ABZ_CSTRIDE + cls;
    if (xs) (void)__hip_atomic_fetch_add(s, xs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ys) (void)__hip_atomic_fetch_add(s + 1, ys, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

/* order-preserving u64 image of a double (negative values, -0.0 included, sort below the positive ones) */
__device__ inline unsigned long long f64_order_key(double x) {
  const unsigned long long u = abz_d2u(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__device__ inline double f64_from_order_key_dev(unsigned long long k) {
  return (k >> 63) ? abz_u2d(k & 0x7FFFFFFFFFFFFFFFull) : abz_u2d(~k);
}

#endif /* ABZ_DEVICE_H */
