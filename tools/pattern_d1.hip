// The d = 1 sweep's memory accesses with the arithmetic removed: own 8-byte row + log-prior + distance (coalesced), two random
// 8-byte donor rows, (A) with the three slot-bit look-ups of the packed population (own word coalesced, two random 4-byte words of a
// 1 MB bitmap) and only accepted rows written, (B) without bit look-ups, every row written to the other buffer (double buffer).
// NOTE (round 3): in (A) and (B) the compiler sinks the two donor loads under the "row written" branch -- only wfrac % of the lanes
// gather donors -- so they overstate what the real sweep can reach (it needs both donors for every position).  (C) below gathers them
// unconditionally and is the pattern to compare the d = 1 kernels with.
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

// (C) = (B) + what the real d = 1 sweep does besides: 8 KB of sampler tables staged into LDS by every workgroup (two 16-byte loads
// per thread, a barrier), the own bitmap word read and one word written per 32 positions, log-prior / distance written for the
// accepted 40 %.  (D), (E) = (C) with P = 2 / 4 CONSECUTIVE positions per thread: the coalesced accesses become 16- / 32-byte
// accesses per lane (half / a quarter as many vector-memory instructions per update), the random donor reads stay 8 bytes each.
template <int P, int STAGE = 1, int BITS = 1, int LOOP = 0, int BAR = 1, int LDSR = 1>
__global__ __launch_bounds__(256) void k2(const double* __restrict__ s0, const uint32_t* __restrict__ bits, uint32_t* __restrict__ bits_out,
                                          uint32_t n, double* __restrict__ w1, double* __restrict__ st, const double2* __restrict__ tab, int wfrac) {
  __shared__ double2 s_tab[512];
  if (STAGE) for (int q = threadIdx.x; q < 512; q += 256) s_tab[q] = tab[q];
  if (LOOP) __syncthreads();
  const uint32_t ntiles = (n / P + 255) / 256;
  for (uint32_t tile = blockIdx.x; tile < (LOOP ? ntiles : blockIdx.x + 1); tile += gridDim.x) {
  const uint32_t r0 = (tile * 256 + threadIdx.x) * P;
  if (r0 >= n) { if (!LOOP && BAR) __syncthreads(); return; }          /* (n is a multiple of 256 * P in this tool) */
  double own[P], lp[P], dl[P];
  uint32_t a[P], b[P];
#pragma unroll
  for (int p = 0; p < P; ++p) { a[p] = hash32((r0 + p) * 2 + 1) % n; b[p] = hash32((r0 + p) * 2 + 2) % n; }
  if constexpr (P == 1) { own[0] = s0[r0]; lp[0] = st[r0]; dl[0] = st[n + r0]; }
  else if constexpr (P == 2) {
    const double2 o = *(const double2*)(s0 + r0), l = *(const double2*)(st + r0), d = *(const double2*)(st + n + r0);
    own[0] = o.x; own[1] = o.y; lp[0] = l.x; lp[1] = l.y; dl[0] = d.x; dl[1] = d.y;
  } else {
    const double4 o = *(const double4*)(s0 + r0), l = *(const double4*)(st + r0), d = *(const double4*)(st + n + r0);
    own[0] = o.x; own[1] = o.y; own[2] = o.z; own[3] = o.w; lp[0] = l.x; lp[1] = l.y; lp[2] = l.z; lp[3] = l.w;
    dl[0] = d.x; dl[1] = d.y; dl[2] = d.z; dl[3] = d.w;
  }
  const uint32_t wi = BITS ? bits[r0 >> 5] : 0u;
  double da[P], db[P];
#pragma unroll
  for (int p = 0; p < P; ++p) { da[p] = s0[a[p]]; db[p] = s0[b[p]]; }
  if (!LOOP && BAR) __syncthreads();
  double v[P];
  bool wr[P];
  uint32_t m = 0;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const double2 t = LDSR ? s_tab[(a[p] ^ b[p]) & 511] : make_double2(1.0, 2.0);
    v[p] = own[p] + (da[p] - db[p]) * t.x + lp[p] * 1e-300 + dl[p] * 1e-300 + t.y * 1e-300;
    wr[p] = (hash32((r0 + p) * 7 + 3) % 100) < (uint32_t)wfrac;
    m |= (uint32_t)wr[p] << p;
  }
  if constexpr (P == 1) w1[r0] = wr[0] ? v[0] : own[0];
  else if constexpr (P == 2) *(double2*)(w1 + r0) = make_double2(wr[0] ? v[0] : own[0], wr[1] ? v[1] : own[1]);
  else *(double4*)(w1 + r0) = make_double4(wr[0] ? v[0] : own[0], wr[1] ? v[1] : own[1], wr[2] ? v[2] : own[2], wr[3] ? v[3] : own[3]);
#pragma unroll
  for (int p = 0; p < P; ++p) if (wr[p]) { st[r0 + p] = v[p] * 1e-300; st[n + r0 + p] = 2.0; }
  /* one bitmap word per 32 positions: 32 / P lanes share a word */
  uint32_t word = m << ((r0 & 31u));
  for (int off = 1; off < 32 / P; off <<= 1) word |= __shfl_xor(word, off, 64);
  if (BITS && (r0 & 31u) == 0) bits_out[r0 >> 5] = wi ^ word;
  }
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
  double2* tab; CK(hipMalloc(&tab, 8192)); CK(hipMemset(tab, 0, 8192));
  for (int var = 5; var < 9; ++var) {
    const uint32_t M = 3u * (N / 4);
    const unsigned grid = (M + 255) / 256;
    auto launch = [&] {
      if (var == 5) hipLaunchKernelGGL((k2<1, 0, 0, 0, 0, 1>), dim3(grid), dim3(256), 0, 0, q0, bits, bo, M, q1, st, tab, 40);
      if (var == 6) hipLaunchKernelGGL((k2<1, 0, 0, 0, 1, 0>), dim3(grid), dim3(256), 0, 0, q0, bits, bo, M, q1, st, tab, 40);
      if (var == 7) hipLaunchKernelGGL((k2<1, 0, 0, 0, 0, 0>), dim3(grid), dim3(256), 0, 0, q0, bits, bo, M, q1, st, tab, 40);
      if (var == 8) hipLaunchKernelGGL((k2<1, 1, 1, 0, 1, 0>), dim3(grid), dim3(256), 0, 0, q0, bits, bo, M, q1, st, tab, 40);
    };
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 20; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    const char* names[4] = {"C5 no barrier, LDS read", "C6 barrier, no LDS read", "C7 no barrier, no LDS read", "C8 staging + bitmap + barrier, no LDS read"};
    printf("{\"variant\": \"%s\", \"prefix\": %u, \"ms\": %.4f, \"particles_per_s\": %.4e}\n", names[var - 5], M, ms, M / (ms * 1e-3));
  }
  for (int var = 0; var < 5; ++var) {
    const uint32_t M = 3u * (N / 4);
    const unsigned grid = var == 4 ? 256u * 8u : (M + 255) / 256;
    auto launch = [&] {
      if (var == 0) hipLaunchKernelGGL((k2<1, 0, 0, 0>), dim3(grid), dim3(256), 0, 0, q0, bits, bo, M, q1, st, tab, 40);
      if (var == 1) hipLaunchKernelGGL((k2<1, 1, 0, 0>), dim3(grid), dim3(256), 0, 0, q0, bits, bo, M, q1, st, tab, 40);
      if (var == 2) hipLaunchKernelGGL((k2<1, 0, 1, 0>), dim3(grid), dim3(256), 0, 0, q0, bits, bo, M, q1, st, tab, 40);
      if (var == 3) hipLaunchKernelGGL((k2<1, 1, 1, 0>), dim3(grid), dim3(256), 0, 0, q0, bits, bo, M, q1, st, tab, 40);
      if (var == 4) hipLaunchKernelGGL((k2<1, 1, 1, 1>), dim3(grid), dim3(256), 0, 0, q0, bits, bo, M, q1, st, tab, 40);
    };
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 20; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    const char* names[5] = {"C0 neither staging nor bitmap (LDS read of an unstaged table)", "C1 table staging only", "C2 bitmap word only", "C3 staging + bitmap",
                            "C4 staging once per looping workgroup (2048 workgroups) + bitmap"};
    printf("{\"variant\": \"%s\", \"prefix\": %u, \"ms\": %.4f, \"particles_per_s\": %.4e}\n", names[var], M, ms, M / (ms * 1e-3));
  }
  for (uint32_t M : {N, 3u * (N / 4)}) for (int P : {1, 2, 4}) {
    const unsigned grid = (M / P + 255) / 256;
    auto launch = [&] {
      if (P == 1) hipLaunchKernelGGL(k2<1>, dim3(grid), dim3(256), 0, 0, q0, bits, bo, M, q1, st, tab, 40);
      else if (P == 2) hipLaunchKernelGGL(k2<2>, dim3(grid), dim3(256), 0, 0, q0, bits, bo, M, q1, st, tab, 40);
      else hipLaunchKernelGGL(k2<4>, dim3(grid), dim3(256), 0, 0, q0, bits, bo, M, q1, st, tab, 40);
    };
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 20; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("{\"variant\": \"%s\", \"positions_per_thread\": %d, \"prefix\": %u, \"ms\": %.4f, \"particles_per_s\": %.4e}\n",
           "C double buffer + table staging, bitmap word, state written for 40%", P, M, ms, M / (ms * 1e-3));
  }
  {
    const uint32_t M = 3u * (N / 4);
    const unsigned grid = (M + 255) / 256;
    for (int rep = 0; rep < 2; ++rep) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, q0, q1, bits, bo, M, q0, q1, st, 40);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, q0, q1, bits, bo, M, q0, q1, st, 40);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
      printf("{\"variant\": \"B again, after the C variants\", \"prefix\": %u, \"ms\": %.4f, \"particles_per_s\": %.4e}\n", M, ms, M / (ms * 1e-3));
      if (rep == 0) { CK(hipMemset(q0, 0, (size_t)N * 8)); CK(hipMemset(q1, 0, (size_t)N * 8)); CK(hipMemset(st, 0, (size_t)N * 16)); }
    }
  }
  return 0;
}
