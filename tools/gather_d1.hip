// Round-4 experiment: can a d = 1 sweep's donor gather be served with less line traffic?
// configs[4] moves 5.7x its algorithmic bytes because every random 8-byte donor read is a line request of its own, and the sweep
// kernel is bound by the rate of the CU's vector-memory address unit.  The donors of a proposal are uniform
// over the alive set (src/abcdez_smc.jl:119-126) -- that must not change.  What may change is the ORDER in which a tile's requests
// reach the memory system: a wave- or workgroup-cooperative gather that sorts the requests by address (so that lanes of one
// vector-memory instruction ask for neighbouring lines / the same line / the same page), loads them in that order and hands the
// values back through LDS.
//   V0  direct: own row (coalesced) + two random 8-byte reads per position + one coalesced write        (the shipped pattern)
//   V1  the 128 requests of a WAVE sorted by address in LDS (bitonic, 28 steps), loaded in sorted order, values routed back in LDS
//   V2  the 512 requests of a WORKGROUP sorted (bitonic, 45 steps with barriers), same
//   V3  V0 with the donor table in 4-byte floats (what a narrower row would buy: not shippable -- the state is fp64 -- a bound)
//   V4  the lines a wave asks for, counted: distinct 64-byte lines among its 128 requests (no memory access; explains V1 / V2)
//   hipcc --offload-arch=gfx950 -O3 -o gather_d1 gather_d1.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__host__ __device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <typename T>
__global__ __launch_bounds__(256) void v0(const T* __restrict__ s0, uint32_t n, double* __restrict__ w1, const double* __restrict__ own) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint32_t a = hash32(r * 2 + 1) % n, b = hash32(r * 2 + 2) % n;
  w1[r] = own[r] + ((double)s0[a] - (double)s0[b]);
}

// requests of a group of G threads (2 per thread) sorted by address in LDS; G = 64: wave-private, no workgroup barrier
template <int G>
__global__ __launch_bounds__(256) void vsort(const double* __restrict__ s0, uint32_t n, double* __restrict__ w1, const double* __restrict__ own) {
  __shared__ uint32_t s_key[512];
  __shared__ uint16_t s_slot[512];
  __shared__ double s_val[512];
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;         /* n is a multiple of 256 */
  const uint32_t a = hash32(r * 2 + 1) % n, b = hash32(r * 2 + 2) % n;
  const int g0 = (threadIdx.x / G) * G;                       /* first thread of the group */
  const int t = threadIdx.x - g0;
  uint32_t* key = s_key + 2 * g0; uint16_t* slot = s_slot + 2 * g0; double* val = s_val + 2 * g0;
  key[2 * t] = a; slot[2 * t] = (uint16_t)(2 * t); key[2 * t + 1] = b; slot[2 * t + 1] = (uint16_t)(2 * t + 1);
  auto sync = [] { if (G == 64) __builtin_amdgcn_wave_barrier(); else __syncthreads(); };
  sync();
  constexpr int NREQ = 2 * G;
  for (int k = 2; k <= NREQ; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;       /* thread t owns the pair (i, i | j) */
      const bool up = (i & k) == 0;
      const uint32_t ki = key[i], kp = key[p];
      if ((ki > kp) == up) { key[i] = kp; key[p] = ki; const uint16_t si = slot[i]; slot[i] = slot[p]; slot[p] = si; }
      sync();
    }
  const double x0 = s0[key[2 * t]], x1 = s0[key[2 * t + 1]];  /* ascending addresses across the lanes */
  val[slot[2 * t]] = x0; val[slot[2 * t + 1]] = x1;
  sync();
  w1[r] = own[r] + (val[2 * t] - val[2 * t + 1]);
}

__global__ __launch_bounds__(256) void v4(uint32_t n, unsigned long long* __restrict__ out) {
  __shared__ uint32_t s_key[512];
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  const int g0 = (threadIdx.x / 64) * 64, t = threadIdx.x - g0;
  uint32_t* key = s_key + 2 * g0;
  key[2 * t] = (hash32(r * 2 + 1) % n) >> 3; key[2 * t + 1] = (hash32(r * 2 + 2) % n) >> 3;     /* 64-byte line of an 8-byte row */
  __builtin_amdgcn_wave_barrier();
  uint32_t distinct = 0;
  for (int q = 0; q < 2; ++q) {
    const uint32_t mine = key[2 * t + q];
    bool first = true;
    for (int z = 0; z < 2 * t + q; ++z) if (key[z] == mine) { first = false; break; }
    distinct += first;
  }
  for (int off = 32; off; off >>= 1) distinct += __shfl_xor(distinct, off, 64);
  if (t == 0) atomicAdd(out, (unsigned long long)distinct);
}

int main() {
  const uint32_t N = 1u << 23, M = 3u * (N / 4);             /* configs[4]: 2^23 particles, 3/4 alive in the bench window */
  double *q0, *w1, *own; float* f0; unsigned long long* cnt;
  CK(hipMalloc(&q0, (size_t)N * 8)); CK(hipMalloc(&w1, (size_t)N * 8)); CK(hipMalloc(&own, (size_t)N * 8)); CK(hipMalloc(&f0, (size_t)N * 4));
  CK(hipMalloc(&cnt, 8)); CK(hipMemset(cnt, 0, 8));
  CK(hipMemset(q0, 0, (size_t)N * 8)); CK(hipMemset(own, 0, (size_t)N * 8)); CK(hipMemset(f0, 0, (size_t)N * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = M / 256;
  const char* names[4] = {"V0 direct gather (shipped pattern)", "V1 wave-sorted requests (128, LDS bitonic)", "V2 workgroup-sorted requests (512, LDS bitonic)",
                          "V3 direct gather from a 4-byte table (bound, not shippable)"};
  for (int rep = 0; rep < 2; ++rep)
  for (int var = 0; var < 4; ++var) {
    auto launch = [&] {
      if (var == 0) hipLaunchKernelGGL(v0<double>, dim3(grid), dim3(256), 0, 0, q0, M, w1, own);
      if (var == 1) hipLaunchKernelGGL(vsort<64>, dim3(grid), dim3(256), 0, 0, q0, M, w1, own);
      if (var == 2) hipLaunchKernelGGL(vsort<256>, dim3(grid), dim3(256), 0, 0, q0, M, w1, own);
      if (var == 3) hipLaunchKernelGGL(v0<float>, dim3(grid), dim3(256), 0, 0, f0, M, w1, own);
    };
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 20; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("{\"variant\": \"%s\", \"rep\": %d, \"prefix\": %u, \"ms\": %.4f, \"particles_per_s\": %.4e}\n", names[var], rep, M, ms, M / (ms * 1e-3));
  }
  hipLaunchKernelGGL(v4, dim3(grid), dim3(256), 0, 0, M, cnt);
  unsigned long long h = 0; CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
  printf("{\"variant\": \"V4 distinct 64-byte lines per wave of 128 requests\", \"prefix\": %u, \"mean_distinct_lines\": %.3f, \"of\": 128}\n", M, (double)h / (M / 64));
  return 0;
}
