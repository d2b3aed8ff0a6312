// The d = 1 sweep's memory accesses with the arithmetic removed: own 8-byte row + log-prior + distance (coalesced), two random
// 8-byte donor rows, (A) with the three slot-bit look-ups of the packed population (own word coalesced, two random 4-byte words of a
// 1 MB bitmap) and only accepted rows written, (B) without bit look-ups, every row written to the other buffer (double buffer).
//   hipcc --offload-arch=gfx950 -O3 -o pattern_d1 pattern_d1.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__host__ __device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ __launch_bounds__(256) void k(const double* __restrict__ s0, const double* __restrict__ s1, const uint32_t* __restrict__ bits,
                                         uint32_t* __restrict__ bits_out, uint32_t n, double* __restrict__ w0, double* __restrict__ w1,
                                         double* __restrict__ st, int wfrac) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint32_t a = hash32(r * 2 + 1) % n, b = hash32(r * 2 + 2) % n;
  double acc = st[r] + st[n + r];
  if (MODE == 0) {
    const uint32_t bi = (bits[r >> 5] >> (r & 31)) & 1u, ba = (bits[a >> 5] >> (a & 31)) & 1u, bb = (bits[b >> 5] >> (b & 31)) & 1u;
    const double v = (bi ? s1 : s0)[r] + ((ba ? s1 : s0)[a] - (bb ? s1 : s0)[b]);
    const bool wr = (hash32(r * 7 + 3) % 100) < (uint32_t)wfrac;
    if (wr) { (bi ? w0 : w1)[r] = v; st[r] = v * 1e-300; st[n + r] = 2.0; }
    const unsigned long long m = __ballot(wr);
    if ((threadIdx.x & 31) == 0) bits_out[r >> 5] = bits[r >> 5] ^ (uint32_t)(m >> (threadIdx.x & 32));
  } else {
    const double v = s0[r] + (s0[a] - s0[b]);
    const bool wr = (hash32(r * 7 + 3) % 100) < (uint32_t)wfrac;
    w1[r] = wr ? v : s0[r];
    if (wr) { st[r] = v * 1e-300; st[n + r] = 2.0; }
  }
  if (acc == 1.2345e300) st[r] = acc;
}
int main() {
  const uint32_t N = 1u << 23;
  double *q0, *q1, *st; uint32_t *bits, *bo;
  CK(hipMalloc(&q0, (size_t)N * 8)); CK(hipMalloc(&q1, (size_t)N * 8)); CK(hipMalloc(&st, (size_t)N * 16));
  CK(hipMemset(q0, 0, (size_t)N * 8)); CK(hipMemset(q1, 0, (size_t)N * 8)); CK(hipMemset(st, 0, (size_t)N * 16));
  CK(hipMalloc(&bits, N / 8)); CK(hipMalloc(&bo, N / 8));
  { std::vector<uint32_t> h(N / 32); for (uint32_t i = 0; i < N / 32; ++i) h[i] = hash32(i * 977 + 5); CK(hipMemcpy(bits, h.data(), N / 8, hipMemcpyHostToDevice)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (uint32_t M : {N, 3u * (N / 4), N / 2}) for (int mode = 0; mode < 2; ++mode) {
    const unsigned grid = (M + 255) / 256;
    auto launch = [&] {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, q0, q1, bits, bo, M, q0, q1, st, 40);
      else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, q0, q1, bits, bo, M, q0, q1, st, 40);
    };
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 20; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("{\"variant\": \"%s\", \"prefix\": %u, \"ms\": %.4f, \"particles_per_s\": %.4e}\n",
           mode == 0 ? "A packed: slot-bit look-ups, 40% of the rows written" : "B double buffer: no look-ups, every row written", M, ms, M / (ms * 1e-3));
  }
  return 0;
}
