/* abz_dispatch.h -- (simulator, lanes, comps-per-lane) -> template instantiation. */
#ifndef ABZ_DISPATCH_H
#define ABZ_DISPATCH_H

#include <type_traits>

#include "abz_ctx.h"
#include "abz_device.h"

template <int V> using IC = std::integral_constant<int, V>;

/* Shapes.  Component-parallel simulators (MVN) may spread a row over L lanes;
 * the others need the whole row in one thread (L = 1, ld <= 8).                   */
#define ABZ_FOR_EACH_SHAPE(X)                                                            \
  X(ABZ_SIM_MVN, 1, 1) X(ABZ_SIM_MVN, 1, 2) X(ABZ_SIM_MVN, 1, 4) X(ABZ_SIM_MVN, 2, 2)    \
  X(ABZ_SIM_MVN, 2, 4) X(ABZ_SIM_MVN, 4, 2) X(ABZ_SIM_MVN, 4, 4) X(ABZ_SIM_MVN, 8, 2)    \
  X(ABZ_SIM_MVN, 8, 4) X(ABZ_SIM_MVN, 4, 8) X(ABZ_SIM_MVN, 16, 2) X(ABZ_SIM_MVN, 16, 4)  \
  X(ABZ_SIM_MVN, 2, 16) X(ABZ_SIM_MVN, 1, 8) X(ABZ_SIM_MVN, 1, 16) X(ABZ_SIM_MVN, 4, 16) X(ABZ_SIM_MVN, 2, 8) X(ABZ_SIM_MVN, 8, 8)  \
  X(ABZ_SIM_MVN, 8, 16) X(ABZ_SIM_MVN, 8, 32)     /* rows of 128 and 256 doubles */                                      \
  X(ABZ_SIM_NORMAL1D, 1, 1) X(ABZ_SIM_DIRAC, 1, 1) X(ABZ_SIM_MIXTURE, 1, 1)              \
  X(ABZ_SIM_QUAD2D, 1, 2) X(ABZ_SIM_NORMDU, 1, 2) X(ABZ_SIM_WIENER, 1, 2)                \
  X(ABZ_SIM_LV, 1, 4) X(ABZ_SIM_SOCKS, 1, 2)

template <class F>
static inline bool abz_dispatch(int sim, int L, int C, F&& f) {
#define ABZ_X(S, LL, CC)                                   \
  if (sim == S && L == LL && C == CC) {                    \
    f(IC<S>{}, IC<LL>{}, IC<CC>{});                        \
    return true;                                           \
  }
  ABZ_FOR_EACH_SHAPE(ABZ_X)
#undef ABZ_X
  return false;
}

/* simulator-independent kernels: dispatch on the lane-group shape / the row width only */
template <class F>
static inline bool abz_dispatch_lc(int L, int C, F&& f) {
#define ABZ_Y(LL, CC) if (L == LL && C == CC) { f(IC<LL>{}, IC<CC>{}); return true; }
  ABZ_Y(1, 1) ABZ_Y(1, 2) ABZ_Y(1, 4) ABZ_Y(1, 8) ABZ_Y(1, 16) ABZ_Y(2, 2) ABZ_Y(2, 4) ABZ_Y(2, 8) ABZ_Y(2, 16)
  ABZ_Y(4, 2) ABZ_Y(4, 4) ABZ_Y(4, 8) ABZ_Y(4, 16) ABZ_Y(8, 2) ABZ_Y(8, 4) ABZ_Y(8, 8) ABZ_Y(16, 2) ABZ_Y(16, 4)
  ABZ_Y(8, 16) ABZ_Y(8, 32)
#undef ABZ_Y
  return false;
}
template <class F>
static inline bool abz_dispatch_ld(int ld, F&& f) {
#define ABZ_Z(V) if (ld == V) { f(IC<V>{}); return true; }
  ABZ_Z(1) ABZ_Z(2) ABZ_Z(4) ABZ_Z(8) ABZ_Z(16) ABZ_Z(32) ABZ_Z(64) ABZ_Z(128) ABZ_Z(256)
#undef ABZ_Z
  return false;
}

static inline unsigned abz_grid(uint64_t threads) { return (unsigned)((threads + ABZ_BLOCK - 1) / ABZ_BLOCK); }

#endif
