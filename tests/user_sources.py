"""HIP source texts of user-supplied simulators used by the tests and by tools/user_sim_ab.py: built-in simulators restated as a
user would write them, in the three forms of include/abcdez_hip.h (abcdez_ctx_create_user).  Restating a built-in is what makes
them testable: the results must equal the built-in's -- hence the oracle's -- bit for bit."""

# the d-dimensional Normal simulator (x = theta + sigma z, Euclidean distance to the data; BASELINE.json configs[2] at d = 32) in the
# COOPERATIVE form: the row spread over the lanes of a wavefront, 8 components per lane.  Same operations as abz_device.h's
# sim_dist<ABZ_SIM_MVN>: one Box-Muller pair per pair of components, addressed by the pair's index in the row.
USER_MVN_LANES = """
__device__ double abz_user_dist_lanes(const double* th, const abz_user_lanes& g, int d, const double* data, int n_data,
                                      const double* p, abz_user_rng& rng) {
  double sq[ABZ_USER_C];
#pragma unroll
  for (int m = 0; m < ABZ_USER_C / 2; ++m) {
    const int k0 = g.comp(2 * m);                 /* the pair (k0, k0 + 1) of the row lives in th[2 m], th[2 m + 1] */
    double z[2];
    rng.normal_pair_at((uint32_t)(k0 >> 1), z[0], z[1]);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double v = 0.0;
      if (k0 + c < d) { const double x = abz_fma(p[0], z[c], th[2 * m + c]); const double e = x - g.y[k0 + c]; v = e * e; }
      sq[2 * m + c] = v;
    }
  }
  return abz_sqrt(g.sum(sq));
}
"""

# Lotka-Volterra by classical RK4 (BASELINE.json configs[3]) in the STAGED form: one round = one observation and the RK4 interval
# behind it; state = (x, y, running sum of squared errors).  The sum only grows, so its square root is a lower bound of the final
# distance after every round -- and the distance itself after the last.  Same operations as abz_device.h's lv_observe / lv_advance.
USER_LV_ROUNDS = """
#define ABZ_USER_ROUNDS %(rounds)d
#define ABZ_USER_STATE 3
__device__ double abz_user_round(const double* th, int d, const double* data, int n_data, const double* p, abz_user_rng& rng,
                                 int round, double* st) {
  const double a = th[0], b = th[1], c = th[2], e = th[3];
  const double h = p[2], h2 = 0.5 * h, h6 = h / 6.0, sn = p[4];
  const int steps = (int)p[3], nobs = n_data / 2, per = (nobs + ABZ_USER_ROUNDS - 1) / ABZ_USER_ROUNDS;
  double x = round == 0 ? p[0] : st[0], y = round == 0 ? p[1] : st[1], acc = st[2];
  for (int jo = round * per; jo < (round + 1) * per && jo < nobs; ++jo) {
    double z0, z1; rng.normal_pair(z0, z1);
    const double ex = abz_fma(sn, z0, x) - data[2 * jo], ey = abz_fma(sn, z1, y) - data[2 * jo + 1];
    acc = abz_fma(ex, ex, acc); acc = abz_fma(ey, ey, acc);
    if (jo + 1 == nobs) break;
    for (int s = 0; s < steps; ++s) {
      const double k1x = x * abz_fma(-b, y, a), k1y = y * abz_fma(e, x, -c);
      const double xa = abz_fma(h2, k1x, x), ya = abz_fma(h2, k1y, y);
      const double k2x = xa * abz_fma(-b, ya, a), k2y = ya * abz_fma(e, xa, -c);
      const double xb = abz_fma(h2, k2x, x), yb = abz_fma(h2, k2y, y);
      const double k3x = xb * abz_fma(-b, yb, a), k3y = yb * abz_fma(e, xb, -c);
      const double xc = abz_fma(h, k3x, x), yc = abz_fma(h, k3y, y);
      const double k4x = xc * abz_fma(-b, yc, a), k4y = yc * abz_fma(e, xc, -c);
      x = abz_fma(h6, abz_fma(2.0, k2x, k1x) + abz_fma(2.0, k3x, k4x), x);
      y = abz_fma(h6, abz_fma(2.0, k2y, k1y) + abz_fma(2.0, k3y, k4y), y);
    }
  }
  st[0] = x; st[1] = y; st[2] = acc;
  return abz_sqrt(acc);
}
"""

# the same model as ONE opaque call (tests/test_gpu_fullsize.py: USER_LV): every simulated proposal runs all its steps
USER_LV = """
__device__ double abz_user_dist(const double* th, int d, const double* data, int n_data, const double* p, abz_user_rng& rng) {
  const double a = th[0], b = th[1], c = th[2], e = th[3];
  double x = p[0], y = p[1];
  const double h = p[2], h2 = 0.5 * h, h6 = h / 6.0, sn = p[4];
  const int steps = (int)p[3], nobs = n_data / 2;
  double acc = 0.0;
  for (int jo = 0; jo < nobs; ++jo) {
    double z0, z1; rng.normal_pair(z0, z1);
    const double ex = abz_fma(sn, z0, x) - data[2 * jo], ey = abz_fma(sn, z1, y) - data[2 * jo + 1];
    acc = abz_fma(ex, ex, acc); acc = abz_fma(ey, ey, acc);
    if (jo + 1 == nobs) break;
    for (int s = 0; s < steps; ++s) {
      const double k1x = x * abz_fma(-b, y, a), k1y = y * abz_fma(e, x, -c);
      const double xa = abz_fma(h2, k1x, x), ya = abz_fma(h2, k1y, y);
      const double k2x = xa * abz_fma(-b, ya, a), k2y = ya * abz_fma(e, xa, -c);
      const double xb = abz_fma(h2, k2x, x), yb = abz_fma(h2, k2y, y);
      const double k3x = xb * abz_fma(-b, yb, a), k3y = yb * abz_fma(e, xb, -c);
      const double xc = abz_fma(h, k3x, x), yc = abz_fma(h, k3y, y);
      const double k4x = xc * abz_fma(-b, yc, a), k4y = yc * abz_fma(e, xc, -c);
      x = abz_fma(h6, abz_fma(2.0, k2x, k1x) + abz_fma(2.0, k3x, k4x), x);
      y = abz_fma(h6, abz_fma(2.0, k2y, k1y) + abz_fma(2.0, k3y, k4y), y);
    }
  }
  return abz_sqrt(acc);
}
"""

# 9 to 16 parameters, one lane per particle (rows of 16 doubles).  USER_MVN16 is the d-dimensional Normal simulator again (tree sum of
# the squared errors: equal to the built-in, whatever lane shape that runs in).  USER_SEQ16 / USER_SEQ16_ROUNDS are ONE model in the
# opaque and in the staged form: the squared errors accumulated pair by pair in order -- a running sum that only grows, so its square
# root after any number of pairs is a lower bound of the distance.
USER_MVN16 = """
__device__ double abz_user_dist(const double* th, int d, const double* data, int n_data, const double* p, abz_user_rng& rng) {
  double sq[16];
  for (int m = 0; m < 8; ++m) {
    double z[2];
    rng.normal_pair(z[0], z[1]);
    for (int c = 0; c < 2; ++c) {
      const int k = 2 * m + c;
      double v = 0.0;
      if (k < d) { const double x = abz_fma(p[0], z[c], th[k]); const double e = x - data[k]; v = e * e; }
      sq[k] = v;
    }
  }
  return abz_sqrt(abz_tree_sum_small(sq, 16));
}
"""

_SEQ16_PAIR = """
__device__ inline double seq16_pair(const double* th, int d, const double* data, const double* p, abz_user_rng& rng, int m, double acc) {
  double z[2];
  rng.normal_pair(z[0], z[1]);
  for (int c = 0; c < 2; ++c) {
    const int k = 2 * m + c;
    if (k < d) { const double e = abz_fma(p[0], z[c], th[k]) - data[k]; acc = abz_fma(e, e, acc); }
  }
  return acc;
}
"""

USER_SEQ16 = _SEQ16_PAIR + """
__device__ double abz_user_dist(const double* th, int d, const double* data, int n_data, const double* p, abz_user_rng& rng) {
  double acc = 0.0;
  for (int m = 0; m < 8; ++m) acc = seq16_pair(th, d, data, p, rng, m, acc);
  return abz_sqrt(acc);
}
"""

USER_SEQ16_ROUNDS = """
#define ABZ_USER_ROUNDS %(rounds)d
#define ABZ_USER_STATE 1
""" + _SEQ16_PAIR + """
__device__ double abz_user_round(const double* th, int d, const double* data, int n_data, const double* p, abz_user_rng& rng,
                                 int round, double* st) {
  const int per = 8 / ABZ_USER_ROUNDS;
  double acc = st[0];
  for (int m = round * per; m < (round + 1) * per; ++m) acc = seq16_pair(th, d, data, p, rng, m, acc);
  st[0] = acc;
  return abz_sqrt(acc);
}
"""
