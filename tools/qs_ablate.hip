// Round-4 ablation of the select's first pass (qs_hist_kernel, abz_population.hip): where do its ~23 us go?  The kernel streams
// 2.8 M (distance, flag) pairs + 1.4 M dead distances (36 MB: ~7 us at the part's streaming rate) with one 1024-thread block per CU.
// Variants of a faithful copy, each launched behind a small dependent kernel (as in the pipeline), 30 launches each:
//   0 full | 1 no global flush | 2 no LDS atomics (bins computed, not counted) | 3 no dead tail | 4 no block reductions at the end
//   5 loads only (no binning at all) | 6 full, 512 blocks of 512 threads | 7 full, 1024 blocks of 256 threads
//   hipcc --offload-arch=gfx950 -O3 -o qs_ablate qs_ablate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define BINS 2048
__device__ inline unsigned long long okey(double x) { unsigned long long u = __double_as_longlong(x); return (u >> 63) ? ~u : (u | 0x8000000000000000ull); }
template <int FAT>
__device__ inline unsigned long long fat_min(unsigned long long v, unsigned long long* s_w) {
  for (int off = 32; off; off >>= 1) { const unsigned long long a = __shfl_xor(v, off, 64); v = a < v ? a : v; }
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) for (int w = 1; w < FAT / 64; ++w) v = s_w[w] < v ? s_w[w] : v;
  return v;
}
template <int FAT, int VAR>
__global__ __launch_bounds__(FAT) void hist_k(const double* __restrict__ delta, const uint8_t* __restrict__ alive, int64_t N,
                                              const unsigned long long* __restrict__ st, uint32_t* __restrict__ hist,
                                              unsigned long long* __restrict__ bmin, int64_t n_all, unsigned long long* __restrict__ ball) {
  __shared__ uint32_t s_h[BINS];
  __shared__ unsigned long long s_w[FAT / 64];
  unsigned long long alo = ~0ull, ahi = 0ull;
  for (int b = threadIdx.x; b < BINS; b += FAT) s_h[b] = 0;
  const unsigned long long klo = st[0];
  const int shift = (int)st[1];
  __syncthreads();
  unsigned long long lo = ~0ull;
  constexpr bool CONTIG = VAR >= 8;        /* VAR 10: contiguous + no flag loads (the packed prefix is all alive) */        /* block b streams ONE contiguous range of the prefix (and one of the tail) */
  const int64_t nall = VAR == 3 ? N : n_all;
  const int64_t per = ((N + gridDim.x - 1) / gridDim.x + FAT - 1) / FAT * FAT, tper = ((nall - N + gridDim.x - 1) / gridDim.x + FAT - 1) / FAT * FAT;
  const int64_t stride = CONTIG ? (int64_t)FAT : (int64_t)gridDim.x * FAT;
  constexpr int UM = 12, UT = 8;
  int64_t k0 = CONTIG ? (int64_t)blockIdx.x * per + threadIdx.x : (int64_t)blockIdx.x * FAT + threadIdx.x;
  int64_t t0 = N + (CONTIG ? (int64_t)blockIdx.x * tper + threadIdx.x : (int64_t)blockIdx.x * FAT + threadIdx.x);
  const int64_t kend = CONTIG ? (((int64_t)blockIdx.x + 1) * per < N ? ((int64_t)blockIdx.x + 1) * per : N) : N;
  const int64_t tend = CONTIG ? (N + ((int64_t)blockIdx.x + 1) * tper < nall ? N + ((int64_t)blockIdx.x + 1) * tper : nall) : nall;
  while (k0 < kend || t0 < tend) {
    unsigned long long key[UM], tkey[UT];
    uint8_t al[UM];
#pragma unroll
    for (int u = 0; u < UM; ++u) { const int64_t k = k0 + u * stride; const bool in = k < kend; key[u] = in ? okey(delta[k]) : 0ull; al[u] = VAR == 10 ? (uint8_t)in : (in ? alive[k] : (uint8_t)0); }
#pragma unroll
    for (int u = 0; u < UT; ++u) { const int64_t k = t0 + u * stride; tkey[u] = k < tend ? okey(delta[k]) : 0ull; }
#pragma unroll
    for (int u = 0; u < UM; ++u) {
      if (k0 + u * stride < kend) { alo = key[u] < alo ? key[u] : alo; ahi = key[u] > ahi ? key[u] : ahi; }
      if (al[u]) {
        if (VAR != 5) {
          unsigned long long b = (key[u] - klo) >> shift; b = b < BINS ? b : BINS - 1;
          if (VAR != 2) atomicAdd(&s_h[(uint32_t)b], 1u); else lo ^= b;
        }
        lo = key[u] < lo ? key[u] : lo;
      }
    }
#pragma unroll
    for (int u = 0; u < UT; ++u) if (t0 + u * stride < tend) { alo = tkey[u] < alo ? tkey[u] : alo; ahi = tkey[u] > ahi ? tkey[u] : ahi; }
    k0 += UM * stride; t0 += UT * stride;
  }
  __syncthreads();
  if (VAR != 1 && VAR != 5) for (int b = threadIdx.x; b < BINS; b += FAT) if (s_h[b]) atomicAdd(&hist[b], s_h[b]);
  if (VAR == 4) { if (lo == 1 || alo == 1 || ahi == 1) bmin[blockIdx.x] = lo; return; }
  lo = fat_min<FAT>(lo, s_w);
  if (threadIdx.x == 0) bmin[blockIdx.x] = lo;
  __syncthreads();
  alo = fat_min<FAT>(alo, s_w);
  __syncthreads();
  ahi = ~fat_min<FAT>(~ahi, s_w);
  if (threadIdx.x == 0) { ball[blockIdx.x] = alo; ball[gridDim.x + blockIdx.x] = ahi; }
}
__global__ void dep_k(unsigned long long* st, uint32_t* hist) { if (threadIdx.x == 0) st[2] += 1; for (int b = threadIdx.x; b < BINS; b += 256) hist[b] = 0; }

int main() {
  const int64_t NALL = 1 << 22, N = 2900000;
  std::vector<double> h(NALL);
  uint32_t s = 12345;
  for (int64_t i = 0; i < NALL; ++i) { double a = 0; for (int q = 0; q < 12; ++q) { s = s * 1664525u + 1013904223u; a += (s >> 8) * (1.0 / 16777216.0); } h[i] = 6.0 + (a - 6.0) * 0.7; if (i < N && h[i] > 7.2) h[i] = 7.2 - (h[i] - 7.2); }
  double *delta; uint8_t* alive; unsigned long long *st, *bmin, *ball; uint32_t* hist;
  CK(hipMalloc(&delta, NALL * 8)); CK(hipMalloc(&alive, NALL)); CK(hipMalloc(&st, 64)); CK(hipMalloc(&bmin, 1024 * 8)); CK(hipMalloc(&ball, 2048 * 8)); CK(hipMalloc(&hist, BINS * 4));
  CK(hipMemcpy(delta, h.data(), NALL * 8, hipMemcpyHostToDevice)); CK(hipMemset(alive, 1, NALL)); CK(hipMemset(hist, 0, BINS * 4));
  double dmin = 1e300, dmax = -1e300; for (int64_t i = 0; i < N; ++i) { dmin = fmin(dmin, h[i]); dmax = fmax(dmax, h[i]); }
  auto hk = [](double x) { unsigned long long u; memcpy(&u, &x, 8); return (u >> 63) ? ~u : (u | 0x8000000000000000ull); };
  unsigned long long klo = hk(dmin), khi = hk(dmax), range = khi - klo; int bits = 0; while ((range >> bits) != 0) ++bits;
  unsigned long long hst[8] = {klo, (unsigned long long)(bits > 11 ? bits - 11 : 0), 0, 0, 0, 0, 0, 0};
  CK(hipMemcpy(st, hst, 64, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[11] = {"0 full (as shipped)", "1 no global flush", "2 no LDS atomics", "3 no dead tail", "4 no block reductions", "5 loads only",
                          "6 full, 512 blocks x 512 threads", "7 full, 1024 blocks x 256 threads", "8 full, contiguous range per block", "9 contiguous, 512 blocks x 512 threads", "10 contiguous, no flag loads"};
  for (int rep = 0; rep < 2; ++rep)
  for (int var = 0; var < 11; ++var) {
    auto launch = [&] {
      hipLaunchKernelGGL(dep_k, dim3(1), dim3(256), 0, 0, st, hist);
#define L(V) hipLaunchKernelGGL((hist_k<1024, V>), dim3(256), dim3(1024), 0, 0, delta, alive, N, st, hist, bmin, NALL, ball)
      switch (var) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break; case 5: L(5); break;
        case 6: hipLaunchKernelGGL((hist_k<512, 0>), dim3(512), dim3(512), 0, 0, delta, alive, N, st, hist, bmin, NALL, ball); break;
        case 7: hipLaunchKernelGGL((hist_k<256, 0>), dim3(1024), dim3(256), 0, 0, delta, alive, N, st, hist, bmin, NALL, ball); break;
        case 8: L(8); break; case 10: L(10); break;
        case 9: hipLaunchKernelGGL((hist_k<512, 8>), dim3(512), dim3(512), 0, 0, delta, alive, N, st, hist, bmin, NALL, ball); break; }
    };
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 30; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"variant\": \"%s\", \"rep\": %d, \"us_per_pair\": %.2f}\n", names[var], rep, ms * 1000.0 / 30);
  }
  // the dependent kernel alone
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(dep_k, dim3(1), dim3(256), 0, 0, st, hist);
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int r = 0; r < 60; ++r) hipLaunchKernelGGL(dep_k, dim3(1), dim3(256), 0, 0, st, hist);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"variant\": \"the small dependent kernel alone\", \"us_per_launch\": %.2f}\n", ms * 1000.0 / 60);
  return 0;
}
