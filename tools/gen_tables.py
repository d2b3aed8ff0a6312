#!/usr/bin/env python3
"""Generates the lookup tables of the sampler (include/abcdez_tables.h, include/abcdez_tables_data.h):

* the table of the table-driven log (the accept test log(rand) < w, src/abcdez_smc.jl:145);
* the piecewise-polynomial inverse normal CDF behind randn (src/abcdez_smc.jl:128): one 64-bit random word ->
  sign bit + U = 2^-(j+1) (1 + t) in (0, 1), j = the binade of U (leading zeros of the 63 remaining bits), t in [0, 1)
  cut into ABZ_ICDF_SUB sub-intervals; on each, |x| = Q^-1(U / 2) (upper-tail quantile) is a degree-7 polynomial in the local
  coordinate tau in [-1/2, 1/2): interpolation at the Chebyshev nodes, computed with mpmath at 200 bits, coefficients
  rounded to binary64.  Measured against mpmath the polynomial is within 2e-16 of |x| relative to max(|x|, 1/4)
  (tests/test_spec_math.py); the table makes randn cost ~20 vector instructions and no log / sqrt / sincos.

Values are written as hex floats so host and device read bit-identical constants.

    python tools/gen_tables.py          (about 15 s)
"""
import os
import struct
import sys

import mpmath

mpmath.mp.prec = 200
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "include", "abcdez_tables.h")
OUT_DATA = os.path.join(ROOT, "include", "abcdez_tables_data.h")

LOG_BITS = 7
LOG_N = 1 << LOG_BITS
LOG_OFF = 0x3FE6A09E00000000      # ~sqrt(1/2): z = 2^-k x lands in [OFF, 2 OFF)

ICDF_SUB_BITS = 5
ICDF_SUB = 1 << ICDF_SUB_BITS     # sub-intervals per binade
ICDF_BINADES = 64                 # j = 0 .. 63 (63 = all 63 bits zero)
ICDF_DEG = 7
ICDF_HOT_BINADES = 16             # binades kept in LDS by the kernels (P(deeper) = 2^-16 per draw): 32 KB, four workgroups per CU
ICDF_ROWS = ICDF_BINADES * ICDF_SUB


def d2u(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def u2d(u):
    return struct.unpack("<d", struct.pack("<Q", u))[0]


def rnd(x):
    """mpf -> nearest double"""
    return float(mpmath.mpf(x))


def hexf(x):
    return float(x).hex()


def log_table():
    rows = []
    one = d2u(1.0)
    for i in range(LOG_N):
        # bit patterns iz with ((iz - OFF) >> 45) & 127 == i inside the octave starting at OFF
        lo_bits = LOG_OFF + (i << 45)
        hi_bits = LOG_OFF + ((i + 1) << 45) - 1
        # iz = ix - (tmp & 0xfff<<52) maps back into [OFF, OFF + 2^52): patterns are contiguous per i
        zlo, zhi = u2d(lo_bits), u2d(hi_bits)
        if lo_bits <= one <= hi_bits:
            c = 1.0                      # the interval that contains 1.0: r = z - 1 exactly, T = 0
        else:
            mid = (mpmath.mpf(zlo) + mpmath.mpf(zhi)) / 2
            c = rnd(1 / mid)
        T = -mpmath.log(mpmath.mpf(c))
        thi = rnd(T)
        tlo = rnd(T - mpmath.mpf(thi))
        rmax = max(abs(mpmath.mpf(zlo) * c - 1), abs(mpmath.mpf(zhi) * c - 1))
        rows.append((c, thi, tlo, float(rmax)))
    return rows


SQ2 = mpmath.sqrt(2)
SQ2PI = mpmath.sqrt(2 * mpmath.pi)


def upper_quantile(p):
    """x >= 0 with P(Z > x) = p for p in (0, 1/2], by Newton on erfc at working precision"""
    if p >= mpmath.mpf(1) / 2:
        return mpmath.mpf(0)
    # start: the leading asymptotic form for small p, a crude central guess otherwise; Newton converges from either
    if p < mpmath.mpf("0.02"):
        t = mpmath.sqrt(-2 * mpmath.log(p))
        x = t - (mpmath.mpf("2.515517") + mpmath.mpf("0.802853") * t + mpmath.mpf("0.010328") * t * t) / (
            1 + mpmath.mpf("1.432788") * t + mpmath.mpf("0.189269") * t * t + mpmath.mpf("0.001308") * t ** 3)
    else:
        x = (mpmath.mpf(1) / 2 - p) * SQ2PI          # tangent at 0: left of the root, Newton then climbs monotonically
    for _ in range(200):
        q = mpmath.erfc(x / SQ2) / 2
        phi = mpmath.exp(-x * x / 2) / SQ2PI
        dx = (q - p) / phi
        x = x + dx
        if abs(dx) < mpmath.mpf(2) ** -150 * max(abs(x), mpmath.mpf(1)):
            break
    else:
        raise RuntimeError("upper_quantile did not converge for p = %s" % p)
    return x


def icdf_magnitude(j, i, tau):
    """|x| for binade j, sub-interval i, local coordinate tau (mpf in [-1/2, 1/2])"""
    t = (i + mpmath.mpf(1) / 2 + tau) / ICDF_SUB
    U = mpmath.ldexp(1 + t, -(j + 1))
    return upper_quantile(U / 2)


def icdf_table(progress=True):
    n = ICDF_DEG + 1
    nodes = [mpmath.cos(mpmath.pi * (2 * k + 1) / (2 * n)) / 2 for k in range(n)]     # Chebyshev nodes on [-1/2, 1/2]
    A = mpmath.matrix(n, n)
    for r in range(n):
        for c in range(n):
            A[r, c] = nodes[r] ** c
    Ainv = A ** -1
    rows = []
    for j in range(ICDF_BINADES):
        for i in range(ICDF_SUB):
            ys = mpmath.matrix([icdf_magnitude(j, i, t) for t in nodes])
            co = Ainv * ys
            rows.append([rnd(co[c]) for c in range(n)])
        if progress:
            print("icdf binade %d / %d" % (j + 1, ICDF_BINADES), file=sys.stderr, flush=True)
    return rows


def main():
    lt = log_table()
    rmax = max(r[3] for r in lt)
    ic = icdf_table()
    with open(OUT, "w") as f:
        f.write("/* GENERATED by tools/gen_tables.py (mpmath, 200 bits, rounded to binary64) -- do not edit.\n")
        f.write(" * log table: z in [~sqrt(1/2), ~sqrt(2)) split into %d intervals by the top %d mantissa bits of\n" % (LOG_N, LOG_BITS))
        f.write(" * (bits(z) - 0x%016X); per interval c ~ 1/centre, T = -log(c) as hi + lo.  max |z c - 1| = %.3e.\n" % (LOG_OFF, rmax))
        f.write(" * inverse normal CDF: %d binades x %d sub-intervals, degree %d in the local coordinate; row r = binade * %d +\n" % (ICDF_BINADES, ICDF_SUB, ICDF_DEG, ICDF_SUB))
        f.write(" * sub-interval; stored PIECE-MAJOR: piece q of row r = coefficients (2q, 2q+1) at [q][r], so that the 16 lanes an\n")
        f.write(" * LDS read serves together hit 16-byte slots r mod 16 -- spread evenly whatever the rows.  The data is in\n")
        f.write(" * abcdez_tables_data.h (host translation units only). */\n")
        f.write("#ifndef ABCDEZ_TABLES_H\n#define ABCDEZ_TABLES_H\n\n#include <stdint.h>\n\n")
        f.write("#define ABZ_LOG_TAB_BITS %d\n#define ABZ_LOG_TAB_N %d\n#define ABZ_LOG_TAB_OFF 0x%016XULL\n" % (LOG_BITS, LOG_N, LOG_OFF))
        f.write("#define ABZ_ICDF_SUB_BITS %d\n#define ABZ_ICDF_SUB %d\n#define ABZ_ICDF_BINADES %d\n" % (ICDF_SUB_BITS, ICDF_SUB, ICDF_BINADES))
        f.write("#define ABZ_ICDF_ROWS %d\n#define ABZ_ICDF_HOT_BINADES %d\n#define ABZ_ICDF_HOT_ROWS %d\n#define ABZ_ICDF_PIECES %d\n\n" % (
            ICDF_ROWS, ICDF_HOT_BINADES, ICDF_HOT_BINADES * ICDF_SUB, (ICDF_DEG + 1) // 2))
        f.write("typedef struct __attribute__((aligned(16))) { double x, y; } abz_f64x2;\n\n")
        f.write("typedef struct {\n  double logt[ABZ_LOG_TAB_N][4];                           /* c, T_hi, T_lo, pad */\n")
        f.write("  abz_f64x2 icdf_hot[ABZ_ICDF_PIECES][ABZ_ICDF_HOT_ROWS];   /* the first ABZ_ICDF_HOT_BINADES binades (what a kernel keeps in LDS) */\n")
        f.write("  const abz_f64x2* icdf_all;                               /* [ABZ_ICDF_PIECES][ABZ_ICDF_ROWS]: every binade (device: global memory) */\n")
        f.write("  uint64_t pad_;\n} abz_tables;\n\n")
        f.write("#endif\n")
    with open(OUT_DATA, "w") as f:
        f.write("/* GENERATED by tools/gen_tables.py -- do not edit.  The sampler tables' DATA: included by host translation units\n")
        f.write(" * that need the values (the oracle, abz_api.hip which uploads them); kernels get them through a pointer. */\n")
        f.write("#ifndef ABCDEZ_TABLES_DATA_H\n#define ABCDEZ_TABLES_DATA_H\n\n#include \"abcdez_tables.h\"\n\n")
        f.write("#define ABZ_LOGT_INIT { \\\n")
        for c, thi, tlo, _ in lt:
            f.write("    { %s, %s, %s, 0.0 }, \\\n" % (hexf(c), hexf(thi), hexf(tlo)))
        f.write("  }\n\n")
        f.write("static const abz_f64x2 abz_icdf_all_data[ABZ_ICDF_PIECES][ABZ_ICDF_ROWS] = {\n")
        for q in range((ICDF_DEG + 1) // 2):
            f.write("  {\n")
            for r in range(ICDF_ROWS):
                f.write("    { %s, %s },\n" % (hexf(ic[r][2 * q]), hexf(ic[r][2 * q + 1])))
            f.write("  },\n")
        f.write("};\n\n")
        f.write("/* host copy: the hot part stays zero (host code reads every row through icdf_all) */\n")
        f.write("static const abz_tables abz_tables_host = { ABZ_LOGT_INIT, {{{0.0, 0.0}}}, &abz_icdf_all_data[0][0], 0ull };\n\n")
        f.write("#endif\n")
    print("wrote", OUT, OUT_DATA, "max |r| =", rmax)


if __name__ == "__main__":
    main()
