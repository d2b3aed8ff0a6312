/*
 * abcdez_spec.h -- arithmetic specification shared by the HIP kernels and the CPU oracle.
 *
 * Everything here is built only from IEEE-754 binary64 operations that are
 * correctly rounded on both x86-64 (SSE2) and gfx950 (+ - * / fma sqrt, integer
 * ops, bit casts), so the same inputs give bit-identical outputs on host and
 * device.  Both sides must be compiled with -ffp-contract=off; every fused
 * multiply-add below is an explicit abz_fma().
 *
 * The reference (ABCdeZ.jl) takes these pieces from Julia's stdlib / Distributions.jl
 * (SURVEY.md section 8c): rand/randn (src/abcdez_smc.jl:128,145), log (smc:145,
 * types:36,61), exp (smc:79), prior logpdf (priors.jl:40-46), push_p (types.jl:20-23),
 * the four ABC kernels (types.jl:26-73).  Julia's task-local Xoshiro stream is not
 * reproducible on a GPU, so the build's RNG contract is a counter-based Philox4x32-10 (below).
 *
 * Plain C99 / C++ / HIP.  No dependencies.
 */
#ifndef ABCDEZ_SPEC_H
#define ABCDEZ_SPEC_H

#include <stdint.h>

#include "abcdez_tables.h"

#if defined(__HIPCC__)
#define ABZ_HD __host__ __device__ static inline
#else
#define ABZ_HD static inline
#endif
/* The WRAPPER prior families (ABZ_PRIOR_TRUNCATED, ABZ_PRIOR_MIXTURE) in DEVICE code are compiled only into translation units that
 * ask for them (-DABZ_PRIOR_WRAP=1): inlined into every sweep kernel their loops cost the kernels of ALL non-Normal priors a third of
 * their occupancy (94 -> 132 registers and scratch memory on the Lotka-Volterra sweep, measured).  The library compiles the sweep
 * kernels of a model that has such factors at run time (csrc/abz_jit.hip), like those of a user-supplied simulator; its statically
 * compiled initial-population and test-hook kernels carry them.  Host code (the oracle, the library's validation) always has them. */
#if !defined(__HIP_DEVICE_COMPILE__) || defined(ABZ_PRIOR_WRAP)
#define ABZ_HAVE_PRIOR_WRAP 1
#else
#define ABZ_HAVE_PRIOR_WRAP 0
#endif

/* ------------------------------------------------------------------ bit casts */
ABZ_HD uint64_t abz_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
ABZ_HD double abz_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
ABZ_HD double abz_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
ABZ_HD double abz_sqrt(double a) { return __builtin_sqrt(a); }

#define ABZ_INF (abz_u2d(0x7FF0000000000000ull))
#define ABZ_NINF (abz_u2d(0xFFF0000000000000ull))
#define ABZ_NAN (abz_u2d(0x7FF8000000000000ull))

ABZ_HD int abz_isfinite(double x) { return ((abz_d2u(x) >> 52) & 0x7FF) != 0x7FF; }
ABZ_HD int abz_isnan(double x) { return x != x; }

/* ------------------------------------------------------------------ Philox4x32-10
 * Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11).
 * counter = (c0,c1,c2,c3), key = (k0,k1).  R = ABZ_PHILOX_ROUNDS = 10: Random123's default and SURVEY.md section 7's RNG
 * contract.  (Round 3 shipped 7 rounds -- the paper's smallest Crush-resistant count -- for 0-1.5 % of sweep time; the
 * counters here are maximally structured (position, sweep, sub-index, purpose tag), which is the case the extra rounds
 * are the margin for, so the default is 10 again.  A build may override it with -DABZ_PHILOX_ROUNDS=7 on BOTH sides
 * (library and oracle); the library reports the count it was built with, abcdez_abi_layout / abcdez_rng_rounds, and
 * checkpoints carry it.)  tests/test_spec_math.py pins the round function against Random123's published 10-round known
 * answers and a pure-Python Philox for every round count.                                   */
typedef struct { uint32_t v[4]; } abz_u32x4;

#ifndef ABZ_PHILOX_ROUNDS
#define ABZ_PHILOX_ROUNDS 10
#endif
#define ABZ_PHILOX_M0 0xD2511F53u
#define ABZ_PHILOX_M1 0xCD9E8D57u
#define ABZ_PHILOX_W0 0x9E3779B9u
#define ABZ_PHILOX_W1 0xBB67AE85u

#if defined(__HIP_DEVICE_COMPILE__)
#define ABZ_XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)   /* one v_bitop3_b32 */
#else
#define ABZ_XOR3(a, b, c) ((a) ^ (b) ^ (c))
#endif

#define ABZ_PHILOX_ROUND()                                                     \
  {                                                                            \
    uint64_t p0 = (uint64_t)ABZ_PHILOX_M0 * c0;                                \
    uint64_t p1 = (uint64_t)ABZ_PHILOX_M1 * c2;                                \
    uint32_t n0 = ABZ_XOR3((uint32_t)(p1 >> 32), c1, k0);                      \
    uint32_t n1 = (uint32_t)p1;                                                \
    uint32_t n2 = ABZ_XOR3((uint32_t)(p0 >> 32), c3, k1);                      \
    uint32_t n3 = (uint32_t)p0;                                                \
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;                                        \
    k0 += ABZ_PHILOX_W0; k1 += ABZ_PHILOX_W1;                                  \
  }

ABZ_HD abz_u32x4 abz_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int r = 0; r < ABZ_PHILOX_ROUNDS; ++r) ABZ_PHILOX_ROUND()
  abz_u32x4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3; return o;
}
/* any round count (test hook: the published known answers are for 10 rounds) */
ABZ_HD abz_u32x4 abz_philox4x32_r(int rounds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < rounds; ++r) ABZ_PHILOX_ROUND()
  abz_u32x4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3; return o;
}

/* RNG addressing contract.  One Philox block = 128 bits = two 64-bit words.
 *   c0 = global particle (or stratum) index
 *   c1 = epoch: global sweep number (swarm), retry number (init), draw number (resample)
 *   c2 = sub-index inside the purpose (simulator draw block)
 *   c3 = purpose tag below
 *   key = 64-bit seed                                                              */
enum {
  ABZ_RNG_INIT_PRIOR = 1, /* prior draw, block m covers components 2m,2m+1        */
  ABZ_RNG_INIT_SIM = 2,   /* simulator noise during init                          */
  ABZ_RNG_DONOR = 3,      /* word0 -> donor a, word1 -> donor b                   */
  ABZ_RNG_JITTER = 4,     /* Box-Muller pair, first normal = gamma jitter         */
  ABZ_RNG_ACCEPT = 5,     /* word0 -> accept uniform                              */
  ABZ_RNG_SIM = 6,        /* simulator noise during sweeps                        */
  ABZ_RNG_BETTER = 7,     /* abcdemc "better particle" draw (mc:23)               */
  ABZ_RNG_STRATUM = 8,    /* stratified resampling uniform (smc:47)               */
  ABZ_RNG_INIT_AUX = 9    /* rejection samplers of the Beta / NegativeBinomial priors */
};

typedef struct { uint64_t w0, w1; } abz_u64x2;

ABZ_HD abz_u64x2 abz_rng(uint64_t seed, uint32_t idx, uint32_t epoch, uint32_t sub, uint32_t purpose) {
  abz_u32x4 r = abz_philox4x32(idx, epoch, sub, purpose, (uint32_t)seed, (uint32_t)(seed >> 32));
  abz_u64x2 o;
  o.w0 = ((uint64_t)r.v[1] << 32) | r.v[0];
  o.w1 = ((uint64_t)r.v[3] << 32) | r.v[2];
  return o;
}

/* uniform in (0,1): (k+1/2) 2^-52, k = top 52 bits.  Never 0 or 1 -> log() finite.
 * Built as [1,2) mantissa fill minus (1 - 2^-53); the subtraction is exact.            */
ABZ_HD double abz_u01_open(uint64_t w) {
  return abz_u2d(0x3FF0000000000000ull | (w >> 12)) - 0x1.fffffffffffffp-1;
}
/* uniform in [0,1) with 52 bits: k 2^-52 (the Box-Muller angle) */
ABZ_HD double abz_u01_52(uint64_t w) { return abz_u2d(0x3FF0000000000000ull | (w >> 12)) - 1.0; }
/* uniform in [0,1): k 2^-53, k = top 53 bits (what Julia's rand() returns in law). */
ABZ_HD double abz_u01_co(uint64_t w) { return (double)(w >> 11) * 0x1p-53; }

/* unbiased-to-2^-64 integer in [0,n): floor(w * n / 2^64) */
ABZ_HD uint64_t abz_mulhi64(uint64_t a, uint64_t b) {
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
}
ABZ_HD uint32_t abz_randint(uint64_t w, uint32_t n) { return (uint32_t)abz_mulhi64(w, (uint64_t)n); }

/* ------------------------------------------------------------------ log
 * Argument reduction x = 2^k m, m in [sqrt(1/2), sqrt(2)); f = m-1; s = f/(2+f);
 * log(1+f) = f - f^2/2 + s (f^2/2 + R(s^2)); degree-7 minimax R in s^2 (the
 * classic Sun/fdlibm coefficients).  < 1 ulp.                                      */
ABZ_HD double abz_log_core(uint64_t ux, int k) {   /* ux = bits of a positive normal double */
  /* bring mantissa into [sqrt(1/2), sqrt(2)) */
  uint32_t hx = (uint32_t)(ux >> 32);
  hx += 0x3FF00000u - 0x3FE6A09Eu;
  k += (int)(hx >> 20) - 0x3FF;
  hx = (hx & 0x000FFFFFu) + 0x3FE6A09Eu;
  double m = abz_u2d(((uint64_t)hx << 32) | (ux & 0xFFFFFFFFull));
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  double R = 1.479819860511658591e-01;
  R = abz_fma(R, z, 1.531383769920937332e-01);
  R = abz_fma(R, z, 1.818357216161805012e-01);
  R = abz_fma(R, z, 2.222219843214978396e-01);
  R = abz_fma(R, z, 2.857142874366239149e-01);
  R = abz_fma(R, z, 3.999999999940941908e-01);
  R = abz_fma(R, z, 6.666666666666735130e-01);
  R = R * z;
  double hfsq = 0.5 * f * f;
  double dk = (double)k;
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  return dk * ln2_hi - ((hfsq - abz_fma(s, hfsq + R, dk * ln2_lo)) - f);
}
ABZ_HD double abz_log(double x) {
  uint64_t ux = abz_d2u(x);
  int k = 0;
  if ((ux << 1) == 0) return ABZ_NINF;                /* +-0            */
  if (ux >> 63) return ABZ_NAN;                       /* negative       */
  if ((ux >> 52) == 0x7FF) return x;                  /* +Inf / NaN     */
  if ((ux >> 52) == 0) {                              /* subnormal      */
    x = x * 0x1p54; ux = abz_d2u(x); k = -54;
  }
  return abz_log_core(ux, k);
}

/* ------------------------------------------------------------------ exp
 * x = k ln2 + r, |r| <= ln2/2; exp(r) = 1 + 2r/(2 - c(r)) form with the degree-5
 * Remez polynomial in r^2.  < 1 ulp.                                               */
ABZ_HD double abz_exp(double x) {
  if (abz_isnan(x)) return x;
  if (x > 709.782712893383973096) return ABZ_INF;
  if (x < -745.13321910194110842) return 0.0;
  const double invln2 = 1.44269504088896338700e+00;
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  double t = x * invln2;
  int k = (int)(t + (x < 0.0 ? -0.5 : 0.5));
  double dk = (double)k;
  double hi = abz_fma(-dk, ln2_hi, x);
  double lo = dk * ln2_lo;
  double r = hi - lo;
  double rr = r * r;
  double P = 4.13813679705723846039e-08;
  P = abz_fma(P, rr, -1.65339022054652515390e-06);
  P = abz_fma(P, rr, 6.61375632143793436117e-05);
  P = abz_fma(P, rr, -2.77777777770155933842e-03);
  P = abz_fma(P, rr, 1.66666666666666019037e-01);
  double c = abz_fma(-P, rr, r);
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  /* scale by 2^k without ldexp: two-step to stay in range for subnormal results */
  if (k > 1000) { y = y * 0x1p1000; k -= 1000; }
  else if (k < -1000) { y = y * 0x1p-1000; k += 1000; }
  return y * abz_u2d((uint64_t)(int64_t)(k + 1023) << 52);
}

/* ------------------------------------------------------------------ sincos(2 pi u), u in [0,1)
 * n = round(4u); phi = 2 pi (u - n/4) in [-pi/4, pi/4] (the subtraction is exact);
 * degree-13 / degree-14 minimax kernels; quadrant fix-up.                          */
ABZ_HD void abz_sincos2pi(double u, double* sn, double* cs) {
  double t4 = u * 4.0;
  int n = (int)(t4 + 0.5);              /* 0..4 */
  double phi = (u - (double)n * 0.25) * 6.283185307179586477;
  double z = phi * phi;
  double S = 1.58969099521155010221e-10;
  S = abz_fma(S, z, -2.50507602534068634195e-08);
  S = abz_fma(S, z, 2.75573137070700676789e-06);
  S = abz_fma(S, z, -1.98412698298579493134e-04);
  S = abz_fma(S, z, 8.33333333332248946124e-03);
  S = abz_fma(S, z, -1.66666666666666324348e-01);
  double sp = abz_fma(phi * z, S, phi);
  double C = -1.13596475577881948265e-11;
  C = abz_fma(C, z, 2.08757232129817482790e-09);
  C = abz_fma(C, z, -2.75573143513906633035e-07);
  C = abz_fma(C, z, 2.48015872894767294178e-05);
  C = abz_fma(C, z, -1.38888888888741095749e-03);
  C = abz_fma(C, z, 4.16666666666666019037e-02);
  double cp = abz_fma(z * z, C, abz_fma(-0.5, z, 1.0));
  /* quadrant n&3: 0 (s,c)  1 (c,-s)  2 (-s,-c)  3 (-c,s); negation = sign-bit flip */
  const int swap = n & 1;
  const uint64_t s_neg = (uint64_t)((n >> 1) & 1) << 63;
  const uint64_t c_neg = (uint64_t)(((n + 1) >> 1) & 1) << 63;
  const double s0 = swap ? cp : sp;
  const double c0 = swap ? sp : cp;
  *sn = abz_u2d(abz_d2u(s0) ^ s_neg);
  *cs = abz_u2d(abz_d2u(c0) ^ c_neg);
}

/* ------------------------------------------------------------------ table-driven log / sincos for the sampler
 * (tables: abcdez_tables.h, generated by tools/gen_tables.py).  About half the instructions
 * of the polynomial versions above; the sweep kernel is VALU-bound on exactly this code.   */
static const abz_tables abz_tables_host = ABZ_TABLES_INIT;   /* host copy; kernels stage a device copy into LDS */

/* log(x), x positive normal.  x = 2^k z, z in [~sqrt(1/2), ~sqrt(2)); interval i of z from the
 * top 7 mantissa bits; r = z c_i - 1 (one fma), |r| < 3.9e-3; log z = T_i + log1p(r), Taylor to
 * r^7.  <= 2 ulp (tests/test_spec_math.py).                                                 */
ABZ_HD double abz_log_tab(double x, const abz_tables* T) {
  /* tmp = bits(x) - OFF; k = (int64) tmp >> 52; i = (tmp >> 45) & 127; z = bits(x) - (tmp & 0xFFF0...0) -- written on the two
   * 32-bit words (OFF's low word is zero, so neither subtraction borrows): the compilers otherwise carry a 64-bit
   * subtract and convert k as a 64-bit integer, 6 instructions per call on the device for the same values */
  const uint64_t ix = abz_d2u(x);
  const uint32_t xh = (uint32_t)(ix >> 32);
  const uint32_t th = xh - (uint32_t)(ABZ_LOG_TAB_OFF >> 32);
  const int k = (int)((int32_t)th >> 20);
  const uint32_t i = (th >> (20 - ABZ_LOG_TAB_BITS)) & (ABZ_LOG_TAB_N - 1);
  const double z = abz_u2d(((uint64_t)(xh - (th & 0xFFF00000u)) << 32) | (uint32_t)ix);
  const double* e = T->logt[i];
  const double r = abz_fma(z, e[0], -1.0);
  double p = 0x1.2492492492492p-3;            /*  1/7 */
  p = abz_fma(p, r, -0x1.5555555555555p-3);   /* -1/6 */
  p = abz_fma(p, r, 0x1.999999999999ap-3);    /*  1/5 */
  p = abz_fma(p, r, -0.25);
  p = abz_fma(p, r, 0x1.5555555555555p-2);    /*  1/3 */
  p = abz_fma(p, r, -0.5);
  const double dk = (double)k;
  const double hi = abz_fma(dk, 6.93147180369123816490e-01, e[1]);
  const double lo = abz_fma(dk, 1.90821492927058770002e-10, e[2]);
  return hi + (lo + abz_fma(r * r, p, r));
}

/* sincos(2 pi u), u = k 2^-52 in [0,1).  j = round(256 u); delta = 2 pi (u - j/256), |delta| <=
 * pi/256; Taylor to delta^7 / delta^6; rotate the table entry (sin, cos)(2 pi j / 256).     */
ABZ_HD void abz_sincos2pi_tab(double u, const abz_tables* T, double* sn, double* cs) {
  const double t = u * 256.0;
  const double tr = (t + 0x1.8p52) - 0x1.8p52;          /* nearest integer, 0..256 */
  const int j = (int)tr & (ABZ_SC_TAB_N - 1);
  const double dl = (t - tr) * 0x1.921fb54442d18p-6;    /* 2 pi / 256 */
  const double z = dl * dl;
  const double ps = abz_fma(abz_fma(-0x1.a01a01a01a01ap-13, z, 0x1.1111111111111p-7), z, -0x1.5555555555555p-3);
  const double pc = abz_fma(abz_fma(-0x1.6c16c16c16c17p-10, z, 0x1.5555555555555p-5), z, -0.5);
  const double sd = abz_fma(dl * z, ps, dl);            /* sin(delta)     */
  const double cm1 = z * pc;                            /* cos(delta) - 1 */
  const double S = T->sc[j][0], C = T->sc[j][1];
  *sn = S + abz_fma(S, cm1, C * sd);
  *cs = C + abz_fma(-S, sd, C * cm1);
}

/* the same for u = (w >> 12) 2^-52 taken straight from a random word: t = 256 u is built in [256, 512) and the integer
 * part is read off the bits of t + 0x1.8p52 -- two instructions fewer than abz_sincos2pi_tab(abz_u01_52(w)) for the same values
 * (256 u is exact either way; (t256 - 256) + 0x1.8p52 and t256 + (0x1.8p52 - 256) round the same real number) */
ABZ_HD void abz_sincos2pi_tab_w(uint64_t w, const abz_tables* T, double* sn, double* cs) {
  const double t256 = abz_u2d(0x4070000000000000ull | (w >> 12));
  const double t = t256 - 256.0;
  const double tc = t256 + (0x1.8p52 - 256.0);
  const double tr = tc - 0x1.8p52;                      /* nearest integer, 0..256 */
  const int j = (int)((uint32_t)abz_d2u(tc) & (ABZ_SC_TAB_N - 1));
  const double dl = (t - tr) * 0x1.921fb54442d18p-6;    /* 2 pi / 256 */
  const double z = dl * dl;
  const double ps = abz_fma(abz_fma(-0x1.a01a01a01a01ap-13, z, 0x1.1111111111111p-7), z, -0x1.5555555555555p-3);
  const double pc = abz_fma(abz_fma(-0x1.6c16c16c16c17p-10, z, 0x1.5555555555555p-5), z, -0.5);
  const double sd = abz_fma(dl * z, ps, dl);
  const double cm1 = z * pc;
  const double S = T->sc[j][0], C = T->sc[j][1];
  *sn = S + abz_fma(S, cm1, C * sd);
  *cs = C + abz_fma(-S, sd, C * cm1);
}

/* sqrt(x) for x in the normal range far from over/underflow (here: -2 log u in [2e-16, 74]).
 * Device: v_rsq_f64 seed + the two Goldschmidt/Newton steps and two residual corrections the
 * compiler's own correctly rounded expansion uses, minus its range scaling.  Host: sqrt().  */
ABZ_HD double abz_sqrt_pn(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
#else
  return __builtin_sqrt(x);
#endif
}

/* Box-Muller: one Philox block -> two independent N(0,1).  (randn, smc:128)       */
ABZ_HD void abz_normal_pair(abz_u64x2 w, const abz_tables* T, double* z0, double* z1) {
  const double u1 = abz_u01_open(w.w0);
  const double r = abz_sqrt_pn(-2.0 * abz_log_tab(u1, T));
  double sn, cs;
  abz_sincos2pi_tab_w(w.w1, T, &sn, &cs);                 /* == abz_sincos2pi_tab(abz_u01_52(w.w1), ...) */
  *z0 = r * cs;
  *z1 = r * sn;
}

/* round half to even == Julia round(Int, x) (types.jl:23) */
ABZ_HD double abz_rint(double x) {
  uint64_t u = abz_d2u(x);
  uint32_t e = (uint32_t)((u >> 52) & 0x7FF);
  if (e >= 0x3FF + 52) return x;        /* already integral, Inf or NaN */
  const double big = 0x1p52;
  double r = (u >> 63) ? (x - big) + big : (x + big) - big;
  /* keep the sign of zero results like rint() */
  if (r == 0.0) return (u >> 63) ? -0.0 : 0.0;
  return r;
}
ABZ_HD double abz_floor(double x) {
  double r = abz_rint(x);
  return (r > x) ? r - 1.0 : r;
}

/* ------------------------------------------------------------------ priors (priors.jl:18-61)
 * One descriptor per dimension; the Julia/Python host derives it from the
 * Distribution object (Factored = one per factor).                                 */
enum {
  ABZ_PRIOR_PAD = 0,        /* padding component: value 0, logpdf 0                 */
  ABZ_PRIOR_NORMAL = 1,     /* p0 = mu, p1 = sigma, c0 = -log(sigma) - log(2 pi)/2, c1 = 1/sigma */
  ABZ_PRIOR_UNIFORM = 2,    /* p0 = a, p1 = b (closed support), c0 = -log(b-a)      */
  ABZ_PRIOR_DUNIFORM = 3,   /* p0 = a, p1 = b integers, c0 = -log(b-a+1); discrete  */
  ABZ_PRIOR_BETA = 4,       /* p0 = alpha, p1 = beta, c0 = -log B(alpha, beta); support [0,1]       */
  ABZ_PRIOR_NEGBIN = 5,     /* p0 = r, p1 = p, c0 = r log p - lgamma(r), c1 = log(1-p); discrete k >= 0
                               (Distributions.NegativeBinomial: failures before the r-th success)   */
  /* further univariate families of Distributions.jl in its parametrisations (`prior::Distribution`, smc:165, 215; mc:102):     */
  ABZ_PRIOR_EXPONENTIAL = 6,  /* Exponential(theta): p0 = theta (scale), c0 = -log theta, c1 = 1/theta; x >= 0                  */
  ABZ_PRIOR_GAMMA = 7,        /* Gamma(alpha, theta): p0 = alpha, p1 = theta, c0 = -lgamma(alpha) - alpha log theta, c1 = 1/theta */
  ABZ_PRIOR_LOGNORMAL = 8,    /* LogNormal(mu, sigma): p0, p1, c0 = -log sigma - log(2 pi)/2, c1 = 1/sigma; x > 0               */
  ABZ_PRIOR_CAUCHY = 9,       /* Cauchy(mu, sigma): p0, p1, c0 = -log(pi sigma), c1 = 1/sigma                                   */
  ABZ_PRIOR_LAPLACE = 10,     /* Laplace(mu, theta): p0, p1, c0 = -log(2 theta), c1 = 1/theta                                   */
  ABZ_PRIOR_WEIBULL = 11,     /* Weibull(alpha, theta): p0 = shape, p1 = scale, c0 = log(alpha / theta), c1 = 1/theta; x >= 0   */
  ABZ_PRIOR_INVGAMMA = 12,    /* InverseGamma(alpha, theta): p0, p1, c0 = alpha log theta - lgamma(alpha); x > 0                */
  ABZ_PRIOR_TRUNCNORMAL = 13, /* truncated(Normal(mu, sigma), lo, hi): p0 = mu, p1 = sigma, c1 = lo, reserved = hi (either may be
                                 infinite), c0 = -log sigma - log(2 pi)/2 - log(Phi((hi-mu)/sigma) - Phi((lo-mu)/sigma))        */
  ABZ_PRIOR_LOGISTIC = 14,    /* Logistic(mu, theta): p0, p1, c0 = -log theta, c1 = 1/theta                                     */
  ABZ_PRIOR_TDIST = 15,       /* TDist(nu): p0 = nu, p1 = (nu+1)/2, c0 = lgamma((nu+1)/2) - lgamma(nu/2) - log(nu pi)/2, c1 = 1/nu */
  ABZ_PRIOR_PARETO = 16,      /* Pareto(alpha, theta): p0 = shape, p1 = scale, c0 = log alpha + alpha log theta; x >= theta     */
  ABZ_PRIOR_POISSON = 17,     /* Poisson(lambda): p0 = lambda, c0 = -lambda, c1 = log lambda; discrete k >= 0                    */
  ABZ_PRIOR_BINOMIAL = 18,    /* Binomial(n, p), 0 < p < 1: p0 = n, p1 = p, c0 = lgamma(n+1) + n log(1-p), c1 = log p - log(1-p);
                                 discrete 0 <= k <= n                                                                           */
  /* wrappers around the families above; their records live in the model's `ext` table (abz_model.ext, doubles):              */
  ABZ_PRIOR_TRUNCATED = 19,   /* truncated(parent, lo, hi) of ANY univariate parent above: p0 = offset in ext of the record
                                 [lo, hi, log(cdf(hi) - cdf(lo-)), parent descriptor (7 doubles)]; logpdf = parent's - log mass
                                 inside [lo, hi], -Inf outside; `discrete` = the parent's                                        */
  ABZ_PRIOR_MIXTURE = 20,     /* MixtureModel(components, weights) of univariate components: p0 = K, p1 = offset in ext of K
                                 records [log w_j, cumulative weight, component descriptor (7 doubles)]; logpdf = log-sum-exp of
                                 log w_j + logpdf_j; `discrete` = the components' (all alike)                                    */
  ABZ_PRIOR_AFFINE = 21,      /* mu + sigma * parent (Distributions' `d * sigma + mu`, LocationScale / AffineDistribution) of a continuous
                                 parent, sigma > 0: p0 = offset in ext of [mu, sigma, 1 / sigma, log sigma, parent descriptor (7)];
                                 logpdf(x) = parent((x - mu) / sigma) - log sigma                                                */
  ABZ_PRIOR_LAST = 21
};
#define ABZ_EXT_DESC 7        /* a descriptor stored as doubles: family, discrete, p0, p1, c0, c1, reserved */
#define ABZ_EXT_TRUNC (3 + ABZ_EXT_DESC)
#define ABZ_EXT_MIXC (2 + ABZ_EXT_DESC)
#define ABZ_EXT_AFFINE (4 + ABZ_EXT_DESC)
#define ABZ_MAX_MIX 16        /* components per mixture */

typedef struct {    /* 48 bytes = three 16-byte loads */
  int32_t family;
  int32_t discrete;   /* push_p rounds this component (types.jl:23) */
  double p0, p1, c0, c1, reserved;
} abz_prior_dim;

/* push_p for one component (types.jl:22-23) */
ABZ_HD double abz_push_p(const abz_prior_dim* pd, double x) { return pd->discrete ? abz_rint(x) : x; }

/* log Gamma(x), x > 0: shift to x >= 16 by the recurrence, then the Stirling series to x^-9
 * (truncation 1e-16).  Error < 3e-14 max(1, |lgamma|) (tests/test_spec_math.py).              */
ABZ_HD double abz_lgamma(double x) {
  double prod = 1.0;
  for (int it = 0; it < 16 && x < 16.0; ++it) { prod *= x; x += 1.0; }
  const double xi = 1.0 / x, x2 = xi * xi;
  double ser = 0x1.5555555555555p-4;                        /*  1/12   */
  {
    double t = -0x1.b951e2b18ff23p-11;                      /* -1/1188 */
    t = abz_fma(t, x2, 0x1.3813813813814p-11);              /*  1/1680 */
    t = abz_fma(t, x2, -0x1.a01a01a01a01ap-11);             /* -1/1260 */
    t = abz_fma(t, x2, 0x1.6c16c16c16c17p-9);               /*  1/360  */
    ser = abz_fma(-t, x2, ser);
  }
  const double lg = abz_fma(x - 0.5, abz_log(x), -x) + 0.91893853320467274178;
  return (lg + ser * xi) - abz_log(prod);
}

/* log-densities of the families beyond Normal / (Discrete)Uniform: Beta / NegativeBinomial (the Socks problem of
 * test/runtests.jl:425-491) and the further Distributions.jl families above.  x is already push_p-cast. */
ABZ_HD double abz_prior_logpdf_ext(const abz_prior_dim* pd, double x);
/* Normal / Uniform / DiscreteUniform / padding: branch-free so a wave with mixed families
 * does not serialise.  Normal: z = (x - mu) * (1/sigma), -z^2/2 + c0.                     */
ABZ_HD double abz_prior_logpdf_basic(const abz_prior_dim* pd, double x) {
  const int fam = pd->family;
  const double p0 = pd->p0, p1 = pd->p1, c0 = pd->c0, c1 = pd->c1;
  const double z = (x - p0) * c1;
  const double ln = abz_fma(-0.5 * z, z, c0);
  int inr = (x >= p0) & (x <= p1);
  if (fam == ABZ_PRIOR_DUNIFORM) inr &= (abz_rint(x) == x);
  const double lu = inr ? c0 : ABZ_NINF;
  return fam == ABZ_PRIOR_NORMAL ? ln : (fam == ABZ_PRIOR_PAD ? 0.0 : lu);
}
/* any base family -- what a wrapper's parent / component may be (no nesting: a wrapper inside a wrapper is NaN) */
ABZ_HD double abz_prior_logpdf_base(const abz_prior_dim* pd, double x) {
  if (pd->family >= ABZ_PRIOR_TRUNCATED) return ABZ_NAN;
  return pd->family >= ABZ_PRIOR_BETA ? abz_prior_logpdf_ext(pd, x) : abz_prior_logpdf_basic(pd, x);
}
ABZ_HD void abz_ext_desc(const double* rec, abz_prior_dim* out) {
  out->family = (int32_t)rec[0]; out->discrete = (int32_t)rec[1];
  out->p0 = rec[2]; out->p1 = rec[3]; out->c0 = rec[4]; out->c1 = rec[5]; out->reserved = rec[6];
}
/* the wrapper families: ext = abz_model.ext (NULL: the model has none, and a wrapper descriptor is malformed) */
ABZ_HD double abz_prior_logpdf_wrapped(const abz_prior_dim* pd, double x, const double* ext) {
  if (!ext) return ABZ_NAN;
  if (pd->family == ABZ_PRIOR_TRUNCATED) {
    const double* rec = ext + (size_t)pd->p0;
    if (!(x >= rec[0] && x <= rec[1])) return ABZ_NINF;
    abz_prior_dim par;
    abz_ext_desc(rec + 3, &par);
    return abz_prior_logpdf_base(&par, x) - rec[2];
  }
  if (pd->family == ABZ_PRIOR_AFFINE) {
    const double* rec = ext + (size_t)pd->p0;
    abz_prior_dim par;
    abz_ext_desc(rec + 4, &par);
    return abz_prior_logpdf_base(&par, (x - rec[0]) * rec[2]) - rec[3];
  }
  /* mixture: log-sum-exp around the largest term, terms added in component order.  Two passes over the components instead of an
   * array of terms: a dynamically indexed local array would live in scratch memory on the device -- for every kernel that merely
   * CONTAINS this branch */
  const int K = (int)pd->p0;
  const double* rec = ext + (size_t)pd->p1;
  double m = ABZ_NINF;
  int nan_term = 0;
  for (int j = 0; j < K && j < ABZ_MAX_MIX; ++j) {
    abz_prior_dim c;
    abz_ext_desc(rec + (size_t)j * ABZ_EXT_MIXC + 2, &c);
    const double t = rec[(size_t)j * ABZ_EXT_MIXC] + abz_prior_logpdf_base(&c, x);
    if (t != t) nan_term = 1;
    if (t > m) m = t;
  }
  if (nan_term) return ABZ_NAN;
  if (!(m > ABZ_NINF)) return ABZ_NINF;                            /* outside every component's support */
  double acc = 0.0;
  for (int j = 0; j < K && j < ABZ_MAX_MIX; ++j) {
    abz_prior_dim c;
    abz_ext_desc(rec + (size_t)j * ABZ_EXT_MIXC + 2, &c);
    acc += abz_exp((rec[(size_t)j * ABZ_EXT_MIXC] + abz_prior_logpdf_base(&c, x)) - m);
  }
  return m + abz_log(acc);
}
ABZ_HD double abz_prior_logpdf_ext(const abz_prior_dim* pd, double x) {
  const double p0 = pd->p0, p1 = pd->p1, c0 = pd->c0, c1 = pd->c1;
  switch (pd->family) {
    case ABZ_PRIOR_BETA: {
      if (!(x >= 0.0 && x <= 1.0)) return ABZ_NINF;
      const double a1 = p0 - 1.0, b1 = p1 - 1.0;
      const double t1 = a1 == 0.0 ? 0.0 : a1 * abz_log(x);
      const double t2 = b1 == 0.0 ? 0.0 : b1 * abz_log(1.0 - x);
      return (t1 + t2) + c0;
    }
    case ABZ_PRIOR_NEGBIN:
      if (!(x >= 0.0) || abz_rint(x) != x || x > 0x1p52) return ABZ_NINF;
      return ((abz_lgamma(x + p0) - abz_lgamma(x + 1.0)) + c0) + x * c1;
    default: break;
  }
  if (!abz_isfinite(x)) return ABZ_NINF;          /* +-Inf carries no density in any family below; NaN is outside every support */
  switch (pd->family) {
    case ABZ_PRIOR_EXPONENTIAL:
      return x >= 0.0 ? c0 - x * c1 : ABZ_NINF;
    case ABZ_PRIOR_GAMMA: {
      if (!(x >= 0.0)) return ABZ_NINF;
      const double a1 = p0 - 1.0;
      const double t1 = a1 == 0.0 ? 0.0 : a1 * abz_log(x);
      return (t1 - x * c1) + c0;
    }
    case ABZ_PRIOR_LOGNORMAL: {
      if (!(x > 0.0)) return ABZ_NINF;
      const double lx = abz_log(x), z = (lx - p0) * c1;
      return abz_fma(-0.5 * z, z, c0) - lx;
    }
    case ABZ_PRIOR_CAUCHY: {
      const double z = (x - p0) * c1;
      return c0 - abz_log(abz_fma(z, z, 1.0));
    }
    case ABZ_PRIOR_LAPLACE: {
      const double t = x - p0;
      return c0 - (t < 0.0 ? -t : t) * c1;
    }
    case ABZ_PRIOR_WEIBULL: {
      if (!(x >= 0.0)) return ABZ_NINF;
      const double a1 = p0 - 1.0, lt = abz_log(x * c1);
      const double t1 = a1 == 0.0 ? 0.0 : a1 * lt;
      return (c0 + t1) - abz_exp(p0 * lt);
    }
    case ABZ_PRIOR_INVGAMMA:
      if (!(x > 0.0)) return ABZ_NINF;
      return (c0 - (p0 + 1.0) * abz_log(x)) - p1 / x;
    case ABZ_PRIOR_TRUNCNORMAL: {
      if (!(x >= c1 && x <= pd->reserved)) return ABZ_NINF;
      const double z = (x - p0) / p1;
      return abz_fma(-0.5 * z, z, c0);
    }
    case ABZ_PRIOR_LOGISTIC: {
      double z = (x - p0) * c1;
      z = z < 0.0 ? -z : z;
      return (c0 - z) - 2.0 * abz_log(1.0 + abz_exp(-z));
    }
    case ABZ_PRIOR_TDIST:
      return c0 - p1 * abz_log(abz_fma(x * x, c1, 1.0));
    case ABZ_PRIOR_PARETO:
      if (!(x >= p1)) return ABZ_NINF;
      return c0 - (p0 + 1.0) * abz_log(x);
    case ABZ_PRIOR_POISSON:
      if (!(x >= 0.0) || abz_rint(x) != x || x > 0x1p52) return ABZ_NINF;
      return abz_fma(x, c1, c0) - abz_lgamma(x + 1.0);
    case ABZ_PRIOR_BINOMIAL:
      if (!(x >= 0.0 && x <= p0) || abz_rint(x) != x) return ABZ_NINF;
      return ((c0 - abz_lgamma(x + 1.0)) - abz_lgamma((p0 - x) + 1.0)) + x * c1;
    default:
      return ABZ_NINF;
  }
}

/* logpdf of one (already pushed) component */
ABZ_HD double abz_prior_logpdf1x(const abz_prior_dim* pd, double x, const double* ext) {
  const int fam = pd->family;
#if ABZ_HAVE_PRIOR_WRAP
  if (fam >= ABZ_PRIOR_TRUNCATED) return abz_prior_logpdf_wrapped(pd, x, ext);
#else
  if (fam >= ABZ_PRIOR_TRUNCATED) return ABZ_NAN;       /* (a kernel compiled without the wrapper families is never launched on such a model) */
#endif
  if (fam >= ABZ_PRIOR_BETA) return abz_prior_logpdf_ext(pd, x);
  return abz_prior_logpdf_basic(pd, x);
}
/* a model without wrapper families (every caller that has no ext table at hand) */
ABZ_HD double abz_prior_logpdf1(const abz_prior_dim* pd, double x) { return abz_prior_logpdf1x(pd, x, (const double*)0); }

/* one prior draw for component pair (2m, 2m+1) uses one Philox block:
 *   normal: Box-Muller pair (z0 -> even component, z1 -> odd component)
 *   (d)uniform: w0 -> even component, w1 -> odd component                          */
ABZ_HD double abz_prior_draw1(const abz_prior_dim* pd, uint64_t w, double z) {
  switch (pd->family) {
    case ABZ_PRIOR_NORMAL: return abz_fma(pd->p1, z, pd->p0);
    case ABZ_PRIOR_UNIFORM: return abz_fma(pd->p1 - pd->p0, abz_u01_co(w), pd->p0);
    case ABZ_PRIOR_DUNIFORM: return pd->p0 + abz_floor(abz_u01_co(w) * (pd->p1 - pd->p0 + 1.0));
    default: return 0.0;
  }
}

/* Rejection / inversion samplers of the extended families; only the initial population uses
 * them (smc:242, init.jl:15).  Random numbers: purpose ABZ_RNG_INIT_AUX, sub-index
 * (component k) * 4096 + stream * 1024 + 2 * attempt (+1 for the attempt's uniform).         */
ABZ_HD double abz_gamma_draw(double shape, uint64_t seed, uint32_t i, uint32_t retry, uint32_t base,
                             const abz_tables* T) {
  /* Marsaglia & Tsang (2000); shape < 1 is boosted through Gamma(shape+1) U^(1/shape) */
  const double k = shape < 1.0 ? shape + 1.0 : shape;
  const double d = k - 1.0 / 3.0, c = 1.0 / abz_sqrt(9.0 * d);
  double g = d;
  for (uint32_t a = 0; a < 500; ++a) {
    double z0, z1;
    abz_normal_pair(abz_rng(seed, i, retry, base + 2 * a, ABZ_RNG_INIT_AUX), T, &z0, &z1);
    const double t = 1.0 + c * z0;
    if (!(t > 0.0)) continue;
    const double v = t * t * t;
    const double u = abz_u01_open(abz_rng(seed, i, retry, base + 2 * a + 1, ABZ_RNG_INIT_AUX).w0);
    if (abz_log_tab(u, T) < 0.5 * z0 * z0 + d - d * v + d * abz_log(v)) { g = d * v; break; }
  }
  if (shape < 1.0) {
    const double u = abz_u01_open(abz_rng(seed, i, retry, base + 1001, ABZ_RNG_INIT_AUX).w0);
    g = g * abz_exp(abz_log_tab(u, T) / shape);
  }
  return g;
}
ABZ_HD double abz_prior_draw_ext(const abz_prior_dim* pd, uint64_t seed, uint32_t i, uint32_t retry, uint32_t k,
                                 const abz_tables* T) {
  const uint32_t base = k * 4096u;
  const double p0 = pd->p0, p1 = pd->p1;
  switch (pd->family) {
    case ABZ_PRIOR_BETA: {
      const double x = abz_gamma_draw(p0, seed, i, retry, base, T);
      const double y = abz_gamma_draw(p1, seed, i, retry, base + 1024u, T);
      return x / (x + y);
    }
    case ABZ_PRIOR_NEGBIN: {
      /* NegativeBinomial(r, p) by inversion: P(0) = p^r, P(k+1) = P(k) (k+r)/(k+1) (1-p) */
      const double r = p0, q = 1.0 - p1;
      const double u = abz_u01_co(abz_rng(seed, i, retry, base, ABZ_RNG_INIT_AUX).w0);
      double P = abz_exp(r * abz_log(p1)), cum = P, kk = 0.0;
      for (int it = 0; it < 100000 && u >= cum; ++it) {
        P = P * ((kk + r) / (kk + 1.0)) * q;
        kk += 1.0;
        cum += P;
      }
      return kk;
    }
    default: break;
  }
  const abz_u64x2 w = abz_rng(seed, i, retry, base, ABZ_RNG_INIT_AUX);
  switch (pd->family) {
    case ABZ_PRIOR_EXPONENTIAL:                       /* inversion: -theta log U */
      return -p0 * abz_log(abz_u01_open(w.w0));
    case ABZ_PRIOR_GAMMA:
      return p1 * abz_gamma_draw(p0, seed, i, retry, base, T);
    case ABZ_PRIOR_LOGNORMAL: {
      double z0, z1;
      abz_normal_pair(w, T, &z0, &z1);
      return abz_exp(abz_fma(p1, z0, p0));
    }
    case ABZ_PRIOR_CAUCHY: {                          /* the ratio of two independent standard normals */
      double z0, z1;
      abz_normal_pair(w, T, &z0, &z1);
      return z1 == 0.0 ? p0 : abz_fma(p1, z0 / z1, p0);
    }
    case ABZ_PRIOR_LAPLACE: {                         /* an exponential with a random sign */
      const double e = -abz_log(abz_u01_open(w.w0));
      return (w.w1 >> 63) ? p0 - p1 * e : abz_fma(p1, e, p0);
    }
    case ABZ_PRIOR_WEIBULL: {                         /* theta E^(1/alpha), E exponential */
      const double e = -abz_log(abz_u01_open(w.w0));
      return p1 * abz_exp(abz_log(e) / p0);
    }
    case ABZ_PRIOR_INVGAMMA:
      return p1 / abz_gamma_draw(p0, seed, i, retry, base, T);
    case ABZ_PRIOR_TRUNCNORMAL: {                     /* rejection from the parent Normal; hosts refuse a mass below 1 % */
      const double lo = pd->c1, hi = pd->reserved;
      for (uint32_t a = 0; a < 4096u; ++a) {
        double z0, z1;
        abz_normal_pair(abz_rng(seed, i, retry, base + a, ABZ_RNG_INIT_AUX), T, &z0, &z1);
        const double x0 = abz_fma(p1, z0, p0), x1 = abz_fma(p1, z1, p0);
        if (x0 >= lo && x0 <= hi) return x0;
        if (x1 >= lo && x1 <= hi) return x1;
      }
      return p0 < lo ? lo : (p0 > hi ? hi : p0);
    }
    case ABZ_PRIOR_LOGISTIC: {                        /* inversion: mu + theta log(U / (1 - U)) */
      const double u = abz_u01_open(w.w0);
      return abz_fma(p1, abz_log(u) - abz_log(1.0 - u), p0);
    }
    case ABZ_PRIOR_TDIST: {                           /* Z / sqrt(chi2_nu / nu), chi2_nu = 2 Gamma(nu / 2) */
      double z0, z1;
      abz_normal_pair(w, T, &z0, &z1);
      const double g = abz_gamma_draw(0.5 * p0, seed, i, retry, base + 1024u, T);
      return z0 * abz_sqrt(p0 / (2.0 * g));
    }
    case ABZ_PRIOR_PARETO:                            /* theta U^(-1/alpha) */
      return p1 * abz_exp(-abz_log(abz_u01_open(w.w0)) / p0);
    case ABZ_PRIOR_POISSON: {                         /* inversion from 0: P(0) = exp(-lambda), P(k+1) = P(k) lambda / (k+1) */
      const double u = abz_u01_co(w.w0);
      double P = abz_exp(pd->c0), cum = P, kk = 0.0;
      for (int it = 0; it < 100000 && u >= cum; ++it) {
        kk += 1.0;
        P = P * (p0 / kk);
        cum += P;
        if (kk > p0 && P < 0x1p-70) break;            /* past the mode with terms that can no longer move the sum */
      }
      return kk;
    }
    case ABZ_PRIOR_BINOMIAL: {                        /* inversion from the thinner end: k successes or n - k */
      const int flip = p1 > 0.5;
      const double pp = flip ? 1.0 - p1 : p1, qq = 1.0 - pp, ratio = pp / qq;
      const double u = abz_u01_co(w.w0);
      double P = abz_exp(p0 * abz_log(qq)), cum = P, kk = 0.0;
      for (int it = 0; it < 100000 && u >= cum && kk < p0; ++it) {
        P = P * ((p0 - kk) / (kk + 1.0)) * ratio;
        kk += 1.0;
        cum += P;
        if (kk > p0 * pp && P < 0x1p-70) break;
      }
      return flip ? p0 - kk : kk;
    }
    default:
      return 0.0;
  }
}

/* one draw of any BASE family with its random numbers taken at (epoch, component k, sub-index sub ...) of the purpose
 * ABZ_RNG_INIT_AUX: what the wrapper families below draw their parents / components with */
ABZ_HD double abz_prior_draw_base(const abz_prior_dim* pd, uint64_t seed, uint32_t i, uint32_t epoch, uint32_t k, uint32_t sub,
                                  const abz_tables* T) {
  if (pd->family >= ABZ_PRIOR_TRUNCATED) return ABZ_NAN;
  if (pd->family >= ABZ_PRIOR_BETA) return abz_prior_draw_ext(pd, seed, i, epoch, k, T);
  const abz_u64x2 w = abz_rng(seed, i, epoch, k * 4096u + sub, ABZ_RNG_INIT_AUX);
  double z0, z1;
  abz_normal_pair(w, T, &z0, &z1);
  return abz_prior_draw1(pd, w.w0, z0);
}
/* truncated(parent, lo, hi): rejection from the parent (hosts refuse an interval holding less than 1 % of the parent's mass);
 * attempt a draws at epoch retry | (a + 1) << 20 (the retry number of abcde_init! stays below 2^20).
 * MixtureModel: the component by inversion of the cumulative weights with one uniform, then that component's own sampler. */
ABZ_HD double abz_prior_draw_extx(const abz_prior_dim* pd, uint64_t seed, uint32_t i, uint32_t retry, uint32_t k,
                                  const abz_tables* T, const double* ext) {
  if (pd->family < ABZ_PRIOR_TRUNCATED) return abz_prior_draw_ext(pd, seed, i, retry, k, T);
#if !ABZ_HAVE_PRIOR_WRAP
  return ABZ_NAN;
#else
  if (!ext) return ABZ_NAN;
  if (pd->family == ABZ_PRIOR_TRUNCATED) {
    const double* rec = ext + (size_t)pd->p0;
    abz_prior_dim par;
    abz_ext_desc(rec + 3, &par);
    for (uint32_t a = 0; a < 4095u; ++a) {
      const double x = abz_prior_draw_base(&par, seed, i, retry | ((a + 1u) << 20), k, 3000u, T);
      if (x >= rec[0] && x <= rec[1]) return x;
    }
    return abz_isfinite(rec[0]) ? rec[0] : rec[1];
  }
  if (pd->family == ABZ_PRIOR_AFFINE) {
    const double* rec = ext + (size_t)pd->p0;
    abz_prior_dim par;
    abz_ext_desc(rec + 4, &par);
    return abz_fma(rec[1], abz_prior_draw_base(&par, seed, i, retry, k, 3000u, T), rec[0]);
  }
  const int K = (int)pd->p0;
  const double* rec = ext + (size_t)pd->p1;
  const double u = abz_u01_co(abz_rng(seed, i, retry, k * 4096u + 4000u, ABZ_RNG_INIT_AUX).w0);
  int j = 0;
  while (j + 1 < K && !(u < rec[(size_t)j * ABZ_EXT_MIXC + 1])) ++j;
  abz_prior_dim c;
  abz_ext_desc(rec + (size_t)j * ABZ_EXT_MIXC + 2, &c);
  return abz_prior_draw_base(&c, seed, i, retry, k, 4001u, T);
#endif
}

/* ------------------------------------------------------------------ ABC kernels (types.jl:26-73) */
enum {
  ABZ_K_INDICATOR = 0,         /* Indicator0toeps        0 <= x <= eps   (types:34) */
  ABZ_K_INDICATOR_STRICT = 1,  /* IndicatorStrict0toeps  0 <= x <  eps   (types:46) */
  ABZ_K_EPA = 2,               /* Epa0toeps              0 <= x <= eps   (types:59) */
  ABZ_K_EPA_STRICT = 3         /* EpaStrict0toeps        0 <= x <  eps   (types:71) */
};

ABZ_HD int abz_kernel_insupport(int kind, double eps, double x) {
  if (!(0.0 <= x)) return 0;
  return (kind & 1) ? (x < eps) : (x <= eps);
}
ABZ_HD double abz_kernel_pdf(int kind, double eps, double x) {
  if (!abz_kernel_insupport(kind, eps, x)) return 0.0;
  if (kind < ABZ_K_EPA) return 1.0;
  double t = x / eps;
  return 1.0 - t * t;
}
ABZ_HD double abz_kernel_logpdf(int kind, double eps, double x) {
  if (!abz_kernel_insupport(kind, eps, x)) return ABZ_NINF;
  if (kind < ABZ_K_EPA) return 0.0;
  double t = x / eps;
  return abz_log(1.0 - t * t);
}

/* ------------------------------------------------------------------ donors
 * Uniform over alive particles other than i (and a) WITHOUT rejection
 * (same law as the rejection loops at smc:119-126): draw a rank in a set with the
 * forbidden ranks removed.  ri = rank of i in the sorted alive list.               */
ABZ_HD void abz_donor_ranks(abz_u64x2 w, uint32_t n_alive, uint32_t ri, uint32_t* ra, uint32_t* rb) {
  uint32_t a = abz_randint(w.w0, n_alive - 1);
  if (a >= ri) a += 1;
  uint32_t lo = a < ri ? a : ri, hi = a < ri ? ri : a;
  uint32_t b = abz_randint(w.w1, n_alive - 2);
  if (b >= lo) b += 1;
  if (b >= hi) b += 1;
  *ra = a; *rb = b;
}

/* ------------------------------------------------------------------ abcdemc's "better particle" (mc:20-24)
 * s = rand(rng, (1:nparticles)[Ds .<= Ds[i]]): uniform over the candidate set {j : Ds[j] <= Ds[i]} (i itself belongs to it).
 * Two formulations of the same law:
 *   by rank       s = order[randint(cnt_i)] over an enumeration of the set (needs the particles that draw sorted by distance);
 *   by rejection  uniform j over ALL particles until Ds[j] <= Ds[i]: the first hit is uniform over the set.  No sort, but
 *                 N / |set| trials on average.
 * The rule (a function of the generation's input distances only, so that every implementation takes the same branch): a
 * generation draws by rejection iff at least 1 / ABZ_MC_REJECT_DIV = 1 / 16 of its particles lie at or below eps_target.  Whoever
 * draws has Ds[i] > eps_pop >= eps_target, so its candidate set contains every particle at or below eps_target: then at least
 * N / 16 candidates, at most 16 trials on average (measured on MI355X at N = 2^20: 16 trials of two random 8-byte reads cost
 * half of what the rank pass costs, profiles/r04_mc1d_reject_div_ab.txt) -- and #(Ds > eps_target) never grows again (mc:54: a particle at or below eps_target accepts only dp <=
 * eps_target, the others only dp <= max(eps_pop, Ds[i])), so a run switches once.
 * Trial 2t is word 0, trial 2t + 1 word 1 of block t of the purpose ABZ_RNG_BETTER.  After 2 * ABZ_MC_REJECT_BLOCKS misses
 * (probability (15/16)^1024 < 1e-28 under the rule) the particle keeps itself, which is a member of its own candidate set.          */
#define ABZ_MC_REJECT_BLOCKS 512u
#ifndef ABZ_MC_REJECT_DIV
#define ABZ_MC_REJECT_DIV 16u          /* part of the spec: library and oracle must agree (a build flag for experiments only) */
#endif
ABZ_HD int abz_mc_draws_by_rejection(uint64_t n_above_target, uint64_t n) {
  return n_above_target <= n && (uint64_t)ABZ_MC_REJECT_DIV * (n - n_above_target) >= n;
}
ABZ_HD uint32_t abz_mc_better_by_rejection(uint64_t seed, uint32_t i, uint32_t sweep, const double* delta, uint32_t n, double di,
                                           int* exhausted) {
  *exhausted = 0;
  for (uint32_t t = 0; t < ABZ_MC_REJECT_BLOCKS; ++t) {
    const abz_u64x2 w = abz_rng(seed, i, sweep, t, ABZ_RNG_BETTER);
    const uint32_t j0 = abz_randint(w.w0, n), j1 = abz_randint(w.w1, n);
    const double d0 = delta[j0], d1 = delta[j1];          /* both reads in flight together */
    if (d0 <= di) return j0;
    if (d1 <= di) return j1;
  }
  *exhausted = 1;     /* under the rule: never.  An implementation may treat it as proof that the rule's premise did not hold. */
  return i;
}

/* ------------------------------------------------------------------ resampling fixed point (smc:15-56)
 * The reference walks a sequentially accumulated fp cumsum; a parallel scan cannot
 * reproduce its roundings, so the spec accumulates in exact integers:
 * total mass S = N * 2^b; weight i -> rint(W_i * N * 2^b); stratum s draws
 * R = s * 2^b + (b random bits); pick the smallest i with cumsum_i > R (clamped to the last
 * positive-weight index).  b = 40 fraction bits for N <= 2^23, fewer above so that S < 2^64
 * (N = 2^25, an 8-GPU population of 2^22 per GPU: b = 38).                             */
ABZ_HD int abz_stratum_bits(uint32_t n) {
  int lg = 0;                               /* ceil(log2(n)) */
  while (lg < 32 && ((uint64_t)1 << lg) < (uint64_t)n) ++lg;
  return lg <= 23 ? 40 : 63 - lg;
}
ABZ_HD uint64_t abz_weight_fix(double w, uint32_t n) {
  const int b = abz_stratum_bits(n);
  double v = w * (double)n * abz_u2d((uint64_t)(1023 + b) << 52);     /* * 2^b */
  if (!(v > 0.0)) return 0;           /* zero, negative or NaN weights carry no mass */
  if (v >= 0x1p63) return (uint64_t)1 << 63;
  return (uint64_t)abz_rint(v);
}
ABZ_HD uint64_t abz_stratum_point(uint64_t seed, uint32_t n, uint32_t s, uint32_t draw) {
  const int b = abz_stratum_bits(n);
  abz_u64x2 w = abz_rng(seed, s, draw, 0, ABZ_RNG_STRATUM);
  return ((uint64_t)s << b) | (w.w0 >> (64 - b));
}

/* ------------------------------------------------------------------ canonical summation tree
 * Per-particle sums over components (logpdf, squared distance) use the pure
 * pairwise binary tree over the component index, zero-padded to a power of two:
 * sum(x[0..P)) = sum(x[0..P/2)) + sum(x[P/2..P)).  The lane-group kernels and the
 * oracle both evaluate exactly this tree (addition is commutative, so the order in
 * which sibling nodes become available does not matter).                           */
ABZ_HD double abz_tree_sum_small(const double* x, int p2) { /* p2 = power of two <= 64 */
  double t[64];
  for (int k = 0; k < p2; ++k) t[k] = x[k];
  for (int st = 1; st < p2; st <<= 1)
    for (int k = 0; k < p2; k += 2 * st) t[k] = t[k] + t[k + st];
  return t[0];
}

/* Population-level fp sums (wnorm, sum W^2) use a fixed tile tree:
 * tile = 2048 consecutive elements; slot t (0..255) owns elements m*512+2t+c,
 * m = 0..3, c = 0..1, summed as ((e00+e01)+(e10+e11))+((e20+e21)+(e30+e31));
 * the 256 slots are reduced by the pairwise binary tree over t; tile partials are
 * reduced recursively by the same rule.  Missing elements are +0.0.                */
#define ABZ_TILE 2048

/* ------------------------------------------------------------------ model descriptor
 * What the reference passes as (prior, dist!, varexternal, rng) (smc:215, mc:102).
 * An arbitrary Julia closure cannot run on the device, so dist! is one of the
 * built-in simulators below, selected by id, with its constants in sim_p / data.   */
#define ABZ_MAX_D 256    /* the reference has no upper limit on length(prior) (src/abcdez_smc.jl:234 is a lower bound on nparticles) */
#define ABZ_DEAD 0xFFFFFFFFu

enum {
  /* x ~ N(theta0, sim_p[0]);  dist = |x - data[0]|
   * (examples/minimal_example.jl:10-24, test/runtests.jl:137)                     */
  ABZ_SIM_NORMAL1D = 0,
  /* x_k = theta_k + sim_p[0] z_k, k < d;  dist = sqrt(tree_sum((x_k - data[k])^2))
   * (BASELINE.json config 3; Philox block m feeds components 2m, 2m+1)            */
  ABZ_SIM_MVN = 1,
  /* deterministic: dist = |theta0^2 + 1 - sim_p[0]|   (test/runtests.jl:495-497)  */
  ABZ_SIM_DIRAC = 2,
  /* dist = 50 (x + 0.01 n1 - y^2)^2 + (y - 1 + 0.01 n2)^2, replaced by +Inf with
   * probability sim_p[0]                               (test/runtests.jl:603,614) */
  ABZ_SIM_QUAD2D = 3,
  /* x = theta0 + (coin ? 0.1 n1 : n2); dist = |x - sim_p[0]| (test/runtests.jl:582-583) */
  ABZ_SIM_MIXTURE = 4,
  /* x = (n^2 + du)(n + 0.01 n1); dist = |x - sim_p[0]|  (test/runtests.jl:524-525) */
  ABZ_SIM_NORMDU = 5,
  /* rms_t = sqrt(mu^2 t^2 + sigma^2 t) (0.95 + 0.1 u), t = 0..n_data-1;
   * dist = sum_t |rms_t - data[t]| / n_data            (test/runtests.jl:537-546) */
  ABZ_SIM_WIENER = 6,
  /* Lotka-Volterra, classical RK4 (BASELINE.json config 4):
   * x' = a x - b x y, y' = -c y + e x y, theta = (a,b,c,e), (x0,y0) = (sim_p[0], sim_p[1]),
   * step sim_p[2], sim_p[3] steps between observations, n_data/2 observation times
   * (the first at t = 0), additive N(0, sim_p[4]^2) noise on every observed value;
   * dist = sqrt(sum over the n_data values (obs - data)^2), summed in order        */
  ABZ_SIM_LV = 7,
  /* "Tiny data, ABC and the socks of Karl Broman" (test/runtests.jl:427-437): theta = (n_socks, prop_pairs);
   * n_pairs = round(prop_pairs floor(n_socks/2)), n_odd = n_socks - 2 n_pairs; pick min(n_socks, sim_p[2])
   * socks without replacement (sequential uniform draws, Philox block t/2 word t%2 for pick t);
   * dist = |pairs picked - sim_p[0]| + |odd socks picked - sim_p[1]|                               */
  ABZ_SIM_SOCKS = 8,
  /* user-supplied device function compiled at run time (abcdez_ctx_create_user): the whole row in one thread for d <= 16,
   * spread over the lanes of a wavefront beyond (include/abcdez_hip.h)                     */
  ABZ_SIM_USER = 9
};

/* ---- blobs (second return value of dist!, src/abcdez_smc.jl:137,148; docs/src/index.md:298-324).
 * A blob here is the simulated data behind a particle's current distance.  It is a pure function of the
 * (push_p-cast) parameters and of the random numbers of ONE simulator call, and those are addressed by
 * (origin particle, epoch, init-or-sweep) -- so the population carries that 8-byte STAMP with each distance
 * (set on accept / at init, gathered on resampling exactly like Ds and blobs in smc:96-99) and the blobs are
 * rebuilt from the stamps when the result is read.  The rebuilt distance must equal the stored one bit for bit. */
#define ABZ_MAX_BLOB 64
#define ABZ_STAMP_INIT 0x8000000000000000ull
ABZ_HD uint64_t abz_stamp(uint32_t origin, uint32_t epoch, int init) {
  return (uint64_t)origin | ((uint64_t)(epoch & 0x7FFFFFFFu) << 32) | (init ? ABZ_STAMP_INIT : 0ull);
}
ABZ_HD uint32_t abz_stamp_origin(uint64_t st) { return (uint32_t)st; }
ABZ_HD uint32_t abz_stamp_epoch(uint64_t st) { return (uint32_t)(st >> 32) & 0x7FFFFFFFu; }
ABZ_HD int abz_stamp_is_init(uint64_t st) { return (int)(st >> 63); }
/* doubles per blob of the built-in simulators: the simulated datum / data vector (-1: declared by the user) */
ABZ_HD int abz_sim_blob_size(int sim_id, int d, int n_data) {
  switch (sim_id) {
    case ABZ_SIM_NORMAL1D: case ABZ_SIM_DIRAC: case ABZ_SIM_MIXTURE: case ABZ_SIM_NORMDU: return 1;
    case ABZ_SIM_QUAD2D: case ABZ_SIM_SOCKS: return 2;
    case ABZ_SIM_MVN: return d;
    case ABZ_SIM_WIENER: case ABZ_SIM_LV: return n_data;
    default: return -1;
  }
}

typedef struct abz_model {
  int32_t d;        /* length(prior)                                               */
  int32_t ld;       /* row stride of theta in doubles: smallest power of two >= d  */
  int32_t sim_id;   /* ABZ_SIM_*                                                   */
  int32_t abck;     /* ABZ_K_*  (ABCk keyword, smc:218)                            */
  uint64_t seed;    /* Philox key                                                  */
  int32_t n_data;
  int32_t n_blob;   /* doubles per blob (0 = blobs off); must equal abz_sim_blob_size() for the built-in simulators */
  double sim_p[8];
  const double* data; /* n_data doubles: host memory for the oracle, device memory for the HIP library */
  abz_prior_dim prior[ABZ_MAX_D]; /* entries d..ld-1 are ABZ_PRIOR_PAD             */
  const double* mv;   /* NULL, or the maps of a correlated Normal prior (below): host memory for the oracle; the HIP library
                         copies them to the device at context creation                                                    */
  const double* ext;  /* NULL, or n_ext doubles: the records of the wrapper families (ABZ_PRIOR_TRUNCATED / _MIXTURE); host memory
                         for the oracle, copied to the device by the HIP library                                           */
  int32_t n_ext;
  int32_t reserved0;
} abz_model;

/* ---- a multivariate prior that is not a product (a Distributions.MvNormal in the `prior` position, src/abcdez_types.jl:16,21:
 * particles are vectors, push_p casts every element to float).  theta = mu + L z with z ~ N(0, I) and Sigma = L L^T, so
 *   logpdf(theta) = sum_k [ -z_k^2 / 2 - log L_kk - log(2 pi) / 2 ],   z = W (theta - mu),  W = L^-1 (lower triangular)
 * i.e. the SAME per-component tree as a product of standard Normals, over the whitened components: the per-dimension
 * descriptors of such a model are Normal(0, 1) with c0 = -log L_kk - log(2 pi) / 2, and mv = [mu[ld] | W[ld][ld] | L[ld][ld]]
 * (row-major, zero above the diagonal and in the padding).  Sums run left to right over m <= k with fma.                     */
ABZ_HD double abz_mv_whiten1(const double* mv, int ld, int k, const double* theta) {
  const double* mu = mv;
  const double* W = mv + ld + (size_t)k * ld;
  double z = 0.0;
  for (int m = 0; m <= k; ++m) z = abz_fma(W[m], theta[m] - mu[m], z);
  return z;
}
ABZ_HD double abz_mv_forward1(const double* mv, int ld, int k, const double* z) {
  const double* L = mv + ld + (size_t)ld * ld + (size_t)k * ld;
  double t = 0.0;
  for (int m = 0; m <= k; ++m) t = abz_fma(L[m], z[m], t);
  return mv[k] + t;
}
#define ABZ_MV_DOUBLES(ld) ((size_t)(ld) + 2u * (size_t)(ld) * (size_t)(ld))

#endif /* ABCDEZ_SPEC_H */
