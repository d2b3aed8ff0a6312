// sweep_variants.hip -- times compile-time variants of the d = 32 sweep kernel against each other and against the
// arithmetic-free access pattern, in ONE process on ONE GPU, interleaved rounds (same data, same launch sizes):
//   bash tools/build_sweep_variants.sh && tools/sweep_variants [n_alive] [rounds]
// The variants are the library's kernel body compiled with different knobs (tools/sweep_variant_kernel.hip) and the same kernel
// of other commits from git worktrees (tools/sweep_variant_tree.hip); both lists live in tools/build_sweep_variants.sh, which
// generates variants.inc / trees.inc.  Output: one JSON line per variant.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "abz_kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Variant { const char* name; const char* what; int (*occ)(); void (*launch)(const SmcPackedArgs*, unsigned, hipStream_t); };
static int g_ncu = 256;
// kernels of this tree (tools/sweep_variant_kernel.hip): the library's argument struct, grid = one tile per workgroup
#define V(name, what) extern "C" int sweep_occ_##name(); extern "C" void sweep_launch_##name(const SmcPackedArgs*, unsigned, hipStream_t);
#include "variants.inc"
#undef V
// kernels of other commits (tools/sweep_variant_tree.hip, git worktrees): their own grid, argument struct and tables
#define T(name, what) extern "C" int sweep_occ_##name(); \
  extern "C" void sweep_launch_##name##_raw(const uint32_t*, uint32_t*, double*, double*, double*, double*, unsigned long long*, const void*, const double*, \
                                            double, double, double, uint32_t, uint32_t, int, hipStream_t); \
  static void sweep_launch_##name(const SmcPackedArgs* a, unsigned, hipStream_t st) { \
    sweep_launch_##name##_raw(a->bits, a->bits_out, a->slot0, a->slot1, a->logpi, a->delta, a->cslots, a->hm.prior, a->hm.data, a->eps, a->gamma0, \
                              a->gsig, a->n_alive, a->sweep, g_ncu, st); }
#include "trees.inc"
#undef T
#define V(name, what) {#name, what, sweep_occ_##name, sweep_launch_##name},
#define T(name, what) {#name, what, sweep_occ_##name, sweep_launch_##name},
static const Variant variants[] = {
#include "variants.inc"
#include "trees.inc"
};
#undef V
#undef T

// the arithmetic-free access pattern (tools/layout_bench.hip, variant P; tools/liblayout_bench.so)
extern "C" int layout_bench_packed(uint32_t N, uint32_t M, int wf, int occ, int reps, int inner, double* ms_mean, double* ms_best);

__host__ __device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// population: rows ~ 0.5 + 0.75 N(0,1) (a mid-anneal posterior), both slots filled, random slot bits, exact log-priors
__global__ void fill_rows(double* s0, double* s1, double* logpi, uint32_t N) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double lp = 0.0;
  for (int k = 0; k < 32; ++k) {
    const uint32_t h1 = hash32(i * 64u + 2u * k + 1u), h2 = hash32(i * 64u + 2u * k + 2u + 0x9e3779b9u);
    const double u1 = (h1 + 0.5) * 2.3283064365386963e-10, u2 = (h2 + 0.5) * 2.3283064365386963e-10;
    const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    const double x = 0.5 + 0.75 * z;
    s0[(size_t)i * 32 + k] = x; s1[(size_t)i * 32 + k] = x;
    lp += -0.5 * x * x - 0.9189385332046727;
  }
  logpi[i] = lp;
}
__global__ void fill_bits(uint32_t* b, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) b[i] = hash32(i + 12345u); }
__global__ void fill_f64(double* p, uint32_t n, double v) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }

static unsigned long long read_counter(const unsigned long long* d_cs, int cls) {
  std::vector<unsigned long long> h(ABZ_CSLOTS * ABZ_CSTRIDE);
  CK(hipMemcpy(h.data(), d_cs, h.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t = 0;
  for (int s = 0; s < ABZ_CSLOTS; ++s) t += h[(size_t)s * ABZ_CSTRIDE + cls];
  return t;
}

int main(int argc, char** argv) {
  const uint32_t N = 1u << 22;
  const uint32_t n_alive = argc > 1 ? (uint32_t)atoll(argv[1]) : 2965608u;
  const int rounds = argc > 2 ? atoi(argv[2]) : 12;
  const double target_acc = argc > 3 ? atof(argv[3]) : 0.147;
  const int inner = argc > 4 ? atoi(argv[4]) : 1;    // launches back to back between the two events (sustained clocks, no idle gaps)
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount; g_ncu = ncu;

  double *s0, *s1, *logpi, *delta, *logpi0; uint32_t *bits, *bits_out; unsigned long long* cs;
  CK(hipMalloc(&s0, (size_t)N * 256)); CK(hipMalloc(&s1, (size_t)N * 256));
  CK(hipMalloc(&logpi, (size_t)N * 8)); CK(hipMalloc(&logpi0, (size_t)N * 8)); CK(hipMalloc(&delta, (size_t)N * 8));
  CK(hipMalloc(&bits, N / 8)); CK(hipMalloc(&bits_out, N / 8));
  CK(hipMalloc(&cs, ABZ_CSLOTS * ABZ_CSTRIDE * 8)); CK(hipMemset(cs, 0, ABZ_CSLOTS * ABZ_CSTRIDE * 8));
  fill_rows<<<N / 256, 256>>>(s0, s1, logpi0, N);
  fill_bits<<<N / 32 / 256, 256>>>(bits, N / 32);
  CK(hipDeviceSynchronize());

  // model: 32 x Normal(0,1) priors, MVN simulator sigma 1, data = 1-vector, strict indicator kernel
  abz_prior_dim hp[32]; memset(hp, 0, sizeof(hp));
  for (int k = 0; k < 32; ++k) { hp[k].family = ABZ_PRIOR_NORMAL; hp[k].p0 = 0.0; hp[k].p1 = 1.0; hp[k].c0 = -0.9189385332046727; hp[k].c1 = 1.0; }
  double hy[32]; for (int k = 0; k < 32; ++k) hy[k] = 1.0;
  abz_prior_dim* dp; double* dy; abz_tables* dtab;
  CK(hipMalloc(&dp, sizeof(hp))); CK(hipMemcpy(dp, hp, sizeof(hp), hipMemcpyHostToDevice));
  CK(hipMalloc(&dy, sizeof(hy))); CK(hipMemcpy(dy, hy, sizeof(hy), hipMemcpyHostToDevice));
  CK(hipMalloc(&dtab, sizeof(abz_tables))); CK(hipMemcpy(dtab, &abz_tables_host, sizeof(abz_tables), hipMemcpyHostToDevice));

  SmcPackedArgs a; memset(&a, 0, sizeof(a));
  a.hm.seed = 1; a.hm.prior = dp; a.hm.data = dy; a.hm.tables = dtab;
  a.hm.sim_p[0] = 1.0; a.hm.d = 32; a.hm.abck = ABZ_K_INDICATOR_STRICT; a.hm.n_data = 32; a.hm.n_blob = 0;
  a.bits = bits; a.bits_out = bits_out; a.slot0 = s0; a.slot1 = s1; a.logpi = logpi; a.delta = delta;
  a.cslots = cs; a.flags = nullptr; a.stamp = nullptr; a.stop = nullptr;
  a.gamma0 = 2.38 / sqrt(64.0); a.gsig = 1e-5;
  a.n_alive = n_alive; a.r_lo = 0; a.n_work = n_alive; a.sweep = 7; a.c_cls = ABZ_C_NACC;

  const int nv = (int)(sizeof(variants) / sizeof(variants[0]));
  auto grid_of = [&](const Variant&) { return (unsigned)(((uint64_t)n_alive * 4 + ABZ_BLOCK - 1) / ABZ_BLOCK); };   // this tree: one tile per workgroup
  auto reset = [&]() { CK(hipMemcpy(logpi, logpi0, (size_t)N * 8, hipMemcpyDeviceToDevice)); };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  // eps with the target acceptance (bisection on variant 0).  Every distance starts INSIDE the kernel's support (K(di) = 0, as for every
  // alive particle of a real run): the prior ratio then takes part in the accept test, and the two-phase body skips the simulator
  // for the proposals it has already rejected -- with distances outside the support (rounds 3-4: 1e9) every in-support proposal
  // would be accepted on its distance alone and nothing could be skipped
  double lo = 5.0, hi = 14.0, acc = 0.0;
  for (int it = 0; it < 14; ++it) {
    a.eps = 0.5 * (lo + hi);
    reset(); fill_f64<<<N / 256, 256>>>(delta, N, 0.0);
    const unsigned long long c0 = read_counter(cs, ABZ_C_NACC);
    variants[0].launch(&a, grid_of(variants[0]), 0); CK(hipDeviceSynchronize());
    acc = (double)(read_counter(cs, ABZ_C_NACC) - c0) / n_alive;
    if (acc > target_acc) hi = a.eps; else lo = a.eps;
  }
  fprintf(stderr, "eps = %.4f acceptance = %.4f n_alive = %u\n", a.eps, acc, n_alive);

  std::vector<std::vector<float>> ms(nv);
  std::vector<float> pat[3];                         // pattern at no cap / 4 / 5 waves per SIMD, measured between the rounds
  for (int rd = 0; rd < rounds + 1; ++rd) {
    if (rd > 0 && rd % 3 == 1) {
      const int caps[3] = {0, 4, 5};
      for (int c = 0; c < 3; ++c) {
        double mm = 0, mb = 0;
        if (layout_bench_packed(N, n_alive, (int)std::lround(acc * 100), caps[c], inner > 1 ? 1 : 3, inner, &mm, &mb) == 0) pat[c].push_back((float)mm);
      }
    }
    for (int v = 0; v < nv; ++v) {
      reset(); fill_f64<<<N / 256, 256>>>(delta, N, 0.0); CK(hipDeviceSynchronize());
      const unsigned g = grid_of(variants[v]);
      CK(hipEventRecord(e0, 0));
      // variants named serp*: every other launch walks the prefix from its end (SmcPackedArgs.rev), as the library alternates them
      const bool serp = strncmp(variants[v].name, "serp", 4) == 0;
      for (int k = 0; k < inner; ++k) { a.rev = serp ? (uint32_t)(k & 1) : 0u; variants[v].launch(&a, g, 0); }
      a.rev = 0u;
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (rd > 0) ms[v].push_back(t / inner);    // round 0 = warm-up (code objects, caches)
    }
  }
  for (int v = 0; v < nv; ++v) {
    std::sort(ms[v].begin(), ms[v].end());
    const float med = ms[v][ms[v].size() / 2], mn = ms[v][0];
    printf("{\"variant\": \"%s\", \"what\": \"%s\", \"occupancy_blocks_per_cu\": %d, \"grid\": %u, \"median_ms\": %.4f, \"min_ms\": %.4f, "
           "\"updates_per_s_median\": %.4g, \"read_frac_median\": %.4f, \"n_alive\": %u, \"acceptance\": %.4f}\n",
           variants[v].name, variants[v].what, variants[v].occ(), grid_of(variants[v]), med, mn, n_alive / (med * 1e-3),
           n_alive / (med * 1e-3) * 785.0 / 8e12, n_alive, acc);
  }
  const char* capname[3] = {"none", "4", "5"};
  for (int c = 0; c < 3; ++c) {
    if (pat[c].empty()) continue;
    std::sort(pat[c].begin(), pat[c].end());
    const float med = pat[c][pat[c].size() / 2];
    printf("{\"variant\": \"pattern\", \"what\": \"arithmetic-free access pattern (layout_bench P), single launches, waves per SIMD cap %s\", "
           "\"median_ms\": %.4f, \"min_ms\": %.4f, \"updates_per_s_median\": %.4g, \"read_frac_median\": %.4f}\n",
           capname[c], med, pat[c][0], n_alive / (med * 1e-3), n_alive / (med * 1e-3) * 785.0 / 8e12);
  }
  return 0;
}
