// Throughput of global (agent-scope) atomics at RANDOM addresses of an L2-sized table -- the question behind a
// counting-sort formulation of abcdemc's rank pass (csrc/abz_sort.hip).   hipcc --offload-arch=gfx950 -O3 -o atomic_bench atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__host__ __device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k_add(uint32_t* t, uint32_t mask, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&t[hash32(i) & mask], 1u);
}
__global__ void k_add_ret(uint32_t* t, uint32_t mask, uint32_t n, uint32_t* out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t p = atomicAdd(&t[hash32(i) & mask], 1u); out[p & (n - 1)] = i; }
}
__global__ void k_plain(uint32_t* t, uint32_t mask, uint32_t n, uint32_t* out) {   // same traffic without atomics
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t p = t[hash32(i) & mask]; out[(p + i) & (n - 1)] = i; }
}
int main() {
  const uint32_t n = 1u << 20;
  uint32_t *t, *out;
  CK(hipMalloc(&t, (size_t)(1u << 24) * 4)); CK(hipMalloc(&out, (size_t)n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (uint32_t logm : {8u, 12u, 16u, 20u, 24u}) {
    const uint32_t mask = (1u << logm) - 1;
    float ms[3] = {0, 0, 0};
    for (int v = 0; v < 3; ++v) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(t, 0, (size_t)(1u << 24) * 4));
        CK(hipEventRecord(e0));
        if (v == 0) hipLaunchKernelGGL(k_add, dim3(n / 256), dim3(256), 0, 0, t, mask, n);
        if (v == 1) hipLaunchKernelGGL(k_add_ret, dim3(n / 256), dim3(256), 0, 0, t, mask, n, out);
        if (v == 2) hipLaunchKernelGGL(k_plain, dim3(n / 256), dim3(256), 0, 0, t, mask, n, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1)); best = m < best ? m : best;
      }
      ms[v] = best;
    }
    printf("{\"elements\": %u, \"table_entries\": %u, \"atomic_add_us\": %.1f, \"atomic_add_returning_plus_scatter_us\": %.1f, \"plain_load_plus_scatter_us\": %.1f}\n",
           n, 1u << logm, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3);
  }
  return 0;
}
