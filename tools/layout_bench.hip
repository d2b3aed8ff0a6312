// layout_bench.hip -- evidence for the theta layout decision.
// Measures the DE-donor access pattern alone (N particles, d = 32 doubles each, two uniformly
// random donors per particle, result reduced so nothing is optimised away) for
//   (A) row-major rows  f64[N][32], 4 lanes x 8 components per particle (the shipped layout)
//   (B) component-major f64[32][N] ("SoA"), one thread per particle
//   (C) row-major rows staged through LDS by the whole workgroup before use
// Build: hipcc --offload-arch=gfx950 -O3 tools/layout_bench.hip -o tools/layout_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int D = 32;

__host__ __device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// (A) 4 lanes per particle, lane j loads doubles [2j,2j+1] and [8+2j, 8+2j+1] ... (2 x 16 B per row)
__global__ __launch_bounds__(256) void k_rows(const double* __restrict__ th, uint32_t N, double* __restrict__ out) {
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x, i = gid >> 2; const int j = gid & 3;
  if (i >= N) return;
  const uint32_t a = hash32(i * 2 + 1) % N, b = hash32(i * 2 + 2) % N;
  double acc = 0;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double2 o = *(const double2*)(th + (size_t)i * D + m * 8 + 2 * j);
    const double2 x = *(const double2*)(th + (size_t)a * D + m * 8 + 2 * j);
    const double2 y = *(const double2*)(th + (size_t)b * D + m * 8 + 2 * j);
    acc += o.x + o.y + (x.x - y.x) + (x.y - y.y);
  }
  if (acc == 1.2345e300) out[gid] = acc;
}
// (A2) as (A) but the two donor rows are found through a random 4-byte look-up in a 16 MB index table
//      (what the sweep does when some particles are dead: alive_idx[rank])
__global__ __launch_bounds__(256) void k_rows_idx(const double* __restrict__ th, const uint32_t* __restrict__ idx, uint32_t N,
                                                  double* __restrict__ out) {
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x, i = gid >> 2; const int j = gid & 3;
  if (i >= N) return;
  const uint32_t a = idx[hash32(i * 2 + 1) % N], b = idx[hash32(i * 2 + 2) % N];
  double acc = 0;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double2 o = *(const double2*)(th + (size_t)i * D + m * 8 + 2 * j);
    const double2 x = *(const double2*)(th + (size_t)a * D + m * 8 + 2 * j);
    const double2 y = *(const double2*)(th + (size_t)b * D + m * 8 + 2 * j);
    acc += o.x + o.y + (x.x - y.x) + (x.y - y.y);
  }
  if (acc == 1.2345e300) out[gid] = acc;
}
// (A3) as (A) plus the sweep's writes: one 256-B row + 16 B per particle
__global__ __launch_bounds__(256) void k_rows_rw(const double* __restrict__ th, const uint32_t* __restrict__ idx, uint32_t N,
                                                 double* __restrict__ nth, double* __restrict__ nlp, int use_idx, int wfrac) {
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x, i = gid >> 2; const int j = gid & 3;
  if (i >= N) return;
  uint32_t a = hash32(i * 2 + 1) % N, b = hash32(i * 2 + 2) % N;
  if (use_idx) { a = idx[a]; b = idx[b]; }
  const bool wr = (hash32(i * 7 + 3) % 100) < (uint32_t)wfrac;
  double acc = 0;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double2 o = *(const double2*)(th + (size_t)i * D + m * 8 + 2 * j);
    const double2 x = *(const double2*)(th + (size_t)a * D + m * 8 + 2 * j);
    const double2 y = *(const double2*)(th + (size_t)b * D + m * 8 + 2 * j);
    double2 r; r.x = o.x + (x.x - y.x); r.y = o.y + (x.y - y.y);
    acc += r.x + r.y;                                   /* the reads happen whether or not the row is written */
    if (wr) *(double2*)(nth + (size_t)i * D + m * 8 + 2 * j) = r;
  }
  acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);   /* every lane's loads are live */
  if (j == 0) { nlp[i] = acc; nlp[N + i] = 2.0; }
}
// (A6) the row store's pattern (what bench.py's kernel does): own row + two donors found through the alive list, the
//      accepted fraction written to the particle's OTHER slot, 16 B of state read always and rewritten in place only
//      on accept, 4 B of new alive list per particle
__global__ __launch_bounds__(256) void k_rows_store(const double* __restrict__ th, const uint32_t* __restrict__ idx, uint32_t N,
                                                    double* __restrict__ nth, double* __restrict__ st, uint32_t* __restrict__ aout,
                                                    int wfrac, int use_idx, int flags = 3) {
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x, r = gid >> 2; const int j = gid & 3;
  if (r >= N) return;
  const uint32_t i = use_idx ? idx[r] : r;
  uint32_t a = hash32(r * 2 + 1) % N, b = hash32(r * 2 + 2) % N;
  if (use_idx) { a = idx[a]; b = idx[b]; }
  const bool wr = (hash32(r * 7 + 3) % 100) < (uint32_t)wfrac;
  double acc = (flags & 1) ? st[i] + st[N + i] : 0.0;      /* flags bit 0: read the 16 B of state */
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double2 o = *(const double2*)(th + (size_t)i * D + m * 8 + 2 * j);
    const double2 x = *(const double2*)(th + (size_t)a * D + m * 8 + 2 * j);
    const double2 y = *(const double2*)(th + (size_t)b * D + m * 8 + 2 * j);
    double2 v; v.x = o.x + (x.x - y.x); v.y = o.y + (x.y - y.y);
    acc += v.x + v.y;
    if (wr) *(double2*)(nth + (size_t)i * D + m * 8 + 2 * j) = v;
  }
  acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
  if (j == 0) {
    if (wr) { st[i] = acc * 1e-300; st[N + i] = 2.0; }
    if (flags & 2) aout[r] = wr ? (i | 0x80000000u) : i;    /* flags bit 1: write the 4-B alive-list entry */
    else if (acc == 1.2345e300) aout[r] = 1u;
  }
}
// (A14) as (A6) but the two slots of a particle are ADJACENT in memory (store[N][2][32]): the accepted row is written
//       512 bytes-aligned next to the own row that was just read (same DRAM page) instead of 1 GiB away
__global__ __launch_bounds__(256) void k_rows_store_il(const double* __restrict__ st2, const uint32_t* __restrict__ idx, uint32_t N,
                                                       double* __restrict__ wr2, double* __restrict__ st, uint32_t* __restrict__ aout,
                                                       int wfrac) {
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x, r = gid >> 2; const int j = gid & 3;
  if (r >= N) return;
  const uint32_t i = idx[r];
  const uint32_t a = idx[hash32(r * 2 + 1) % N], b = idx[hash32(r * 2 + 2) % N];
  const uint32_t si = hash32(i * 5 + 1) & 1u, sa = hash32(a * 5 + 1) & 1u, sb = hash32(b * 5 + 1) & 1u;   /* current slots */
  const bool wr = (hash32(r * 7 + 3) % 100) < (uint32_t)wfrac;
  double acc = st[i] + st[N + i];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double2 o = *(const double2*)(st2 + ((size_t)i * 2 + si) * D + m * 8 + 2 * j);
    const double2 x = *(const double2*)(st2 + ((size_t)a * 2 + sa) * D + m * 8 + 2 * j);
    const double2 y = *(const double2*)(st2 + ((size_t)b * 2 + sb) * D + m * 8 + 2 * j);
    double2 v; v.x = o.x + (x.x - y.x); v.y = o.y + (x.y - y.y);
    acc += v.x + v.y;
    if (wr) *(double2*)(wr2 + ((size_t)i * 2 + (si ^ 1u)) * D + m * 8 + 2 * j) = v;
  }
  acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
  if (j == 0) {
    if (wr) { st[i] = acc * 1e-300; st[N + i] = 2.0; }
    aout[r] = wr ? (i | 0x80000000u) : i;
  }
}
// (P) the PACKED population's pattern (abz_kernels.h, smc_swarm_packed_body): alive rank = position, so the own row
//     streams and the donors are addressed directly; the only indirection is one bit per position (current slot) in an
//     L2-resident bitmap; the accepted fraction is written to the position's other slot, 16 B of state read always and
//     rewritten in place on accept, one bitmap word per 32 positions written
__global__ __launch_bounds__(256) void k_packed(const double* __restrict__ s0, const double* __restrict__ s1,
                                                const uint32_t* __restrict__ bits, uint32_t* __restrict__ bits_out, uint32_t N,
                                                double* __restrict__ w0, double* __restrict__ w1, double* __restrict__ st, int wfrac) {
  __shared__ uint32_t s_acc[2];
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x, r = gid >> 2; const int j = gid & 3;
  if (threadIdx.x < 2) s_acc[threadIdx.x] = 0;
  __syncthreads();
  if (r < N) {
    const uint32_t a = hash32(r * 2 + 1) % N, b = hash32(r * 2 + 2) % N;
    const uint32_t bi = (bits[r >> 5] >> (r & 31)) & 1u, ba = (bits[a >> 5] >> (a & 31)) & 1u, bb = (bits[b >> 5] >> (b & 31)) & 1u;
    const bool wr = (hash32(r * 7 + 3) % 100) < (uint32_t)wfrac;
    double acc = st[r] + st[N + r];
    const double* ri = (bi ? s1 : s0) + (size_t)r * D;
    const double* ra = (ba ? s1 : s0) + (size_t)a * D;
    const double* rb = (bb ? s1 : s0) + (size_t)b * D;
    double* ro = (bi ? w0 : w1) + (size_t)r * D;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const double2 o = *(const double2*)(ri + m * 8 + 2 * j);
      const double2 x = *(const double2*)(ra + m * 8 + 2 * j);
      const double2 y = *(const double2*)(rb + m * 8 + 2 * j);
      double2 v; v.x = o.x + (x.x - y.x); v.y = o.y + (x.y - y.y);
      acc += v.x + v.y;
      if (wr) *(double2*)(ro + m * 8 + 2 * j) = v;
    }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
    if (j == 0 && wr) {
      st[r] = acc * 1e-300; st[N + r] = 2.0;
      atomicOr(&s_acc[(threadIdx.x >> 2) >> 5], 1u << ((threadIdx.x >> 2) & 31));
    }
  }
  __syncthreads();
  if (threadIdx.x < 2) { const uint32_t w = blockIdx.x * 2 + threadIdx.x; if (w * 32 < N) bits_out[w] = bits[w] ^ s_acc[threadIdx.x]; }
}
// (Q) as (P), with non-temporal hints: mode bit 0 = the accepted row is stored non-temporally, bit 1 = the own row (a pure
//     stream) is loaded non-temporally, bit 2 = the donor rows too
typedef double d2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k_packed_nt(const double* __restrict__ s0, const double* __restrict__ s1,
                                                   const uint32_t* __restrict__ bits, uint32_t* __restrict__ bits_out, uint32_t N,
                                                   double* __restrict__ w0, double* __restrict__ w1, double* __restrict__ st, int wfrac) {
  __shared__ uint32_t s_acc[2];
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x, r = gid >> 2; const int j = gid & 3;
  if (threadIdx.x < 2) s_acc[threadIdx.x] = 0;
  __syncthreads();
  if (r < N) {
    const uint32_t a = hash32(r * 2 + 1) % N, b = hash32(r * 2 + 2) % N;
    const uint32_t bi = (bits[r >> 5] >> (r & 31)) & 1u, ba = (bits[a >> 5] >> (a & 31)) & 1u, bb = (bits[b >> 5] >> (b & 31)) & 1u;
    const bool wr = (hash32(r * 7 + 3) % 100) < (uint32_t)wfrac;
    double acc = st[r] + st[N + r];
    const double* ri = (bi ? s1 : s0) + (size_t)r * D;
    const double* ra = (ba ? s1 : s0) + (size_t)a * D;
    const double* rb = (bb ? s1 : s0) + (size_t)b * D;
    double* ro = (bi ? w0 : w1) + (size_t)r * D;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const d2v o = (MODE & 2) ? __builtin_nontemporal_load((const d2v*)(ri + m * 8 + 2 * j)) : *(const d2v*)(ri + m * 8 + 2 * j);
      const d2v x = (MODE & 4) ? __builtin_nontemporal_load((const d2v*)(ra + m * 8 + 2 * j)) : *(const d2v*)(ra + m * 8 + 2 * j);
      const d2v y = (MODE & 4) ? __builtin_nontemporal_load((const d2v*)(rb + m * 8 + 2 * j)) : *(const d2v*)(rb + m * 8 + 2 * j);
      const d2v v = o + (x - y);
      acc += v.x + v.y;
      if (wr) { if (MODE & 1) __builtin_nontemporal_store(v, (d2v*)(ro + m * 8 + 2 * j)); else *(d2v*)(ro + m * 8 + 2 * j) = v; }
    }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
    if (j == 0 && wr) {
      st[r] = acc * 1e-300; st[N + r] = 2.0;
      atomicOr(&s_acc[(threadIdx.x >> 2) >> 5], 1u << ((threadIdx.x >> 2) & 31));
    }
  }
  __syncthreads();
  if (threadIdx.x < 2) { const uint32_t w = blockIdx.x * 2 + threadIdx.x; if (w * 32 < N) bits_out[w] = bits[w] ^ s_acc[threadIdx.x]; }
}
// (B) component-major: thread per particle, component k at th[k*N + i]
__global__ __launch_bounds__(256) void k_soa(const double* __restrict__ th, uint32_t N, double* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const uint32_t a = hash32(i * 2 + 1) % N, b = hash32(i * 2 + 2) % N;
  double acc = 0;
#pragma unroll 8
  for (int k = 0; k < D; ++k) acc += th[(size_t)k * N + i] + (th[(size_t)k * N + a] - th[(size_t)k * N + b]);
  if (acc == 1.2345e300) out[i] = acc;
}
// (C) rows staged in LDS: a 256-thread block handles 64 particles; 16 lanes copy one 256-B row
__global__ __launch_bounds__(256) void k_rows_lds(const double* __restrict__ th, uint32_t N, double* __restrict__ out) {
  __shared__ double s[3][64][D + 2];
  const uint32_t p0 = blockIdx.x * 64;
  for (int r = threadIdx.x >> 4; r < 64 * 3; r += 16) {      // row r of the 192 rows this block needs
    const uint32_t p = p0 + (r % 64); const int which = r / 64;
    if (p < N) {
      const uint32_t src = which == 0 ? p : hash32(p * 2 + which) % N;
      const int l = threadIdx.x & 15;
      const double2 v = *(const double2*)(th + (size_t)src * D + 2 * l);
      s[which][r % 64][2 * l] = v.x; s[which][r % 64][2 * l + 1] = v.y;
    }
  }
  __syncthreads();
  const uint32_t i = p0 + (threadIdx.x >> 2); const int j = threadIdx.x & 3;
  if (i >= N) return;
  double acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc += s[0][threadIdx.x >> 2][j * 8 + k] + (s[1][threadIdx.x >> 2][j * 8 + k] - s[2][threadIdx.x >> 2][j * 8 + k]);
  if (acc == 1.2345e300) out[i] = acc;
}

// --packed <prefix> <accepted %> [positions]: only variant P at that operating point, one JSON line (bench.py runs this on the
// GPU it has just timed: the ceiling of the sweep's access pattern on THAT part, in THAT thermal state)
static int packed_only(uint32_t N, uint32_t M, int wf, int occ = 0) {
  // occ > 0: at most `occ` blocks (= waves per SIMD) resident per CU, enforced with dynamic LDS the kernel never touches
  const size_t dyn_lds = occ > 0 ? (size_t)(160 * 1024 / occ - 1024) / 256 * 256 : 0;
  if (dyn_lds > 64 * 1024) CHECK(hipFuncSetAttribute((const void*)k_packed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
  const size_t bytes = (size_t)N * D * 8;
  double *q0, *q1, *nlp; uint32_t *bits, *bo;
  CHECK(hipMalloc(&q0, bytes)); CHECK(hipMalloc(&q1, bytes)); CHECK(hipMemset(q0, 0, bytes)); CHECK(hipMemset(q1, 0, bytes));
  CHECK(hipMalloc(&nlp, (size_t)N * 16)); CHECK(hipMemset(nlp, 0, (size_t)N * 16));
  CHECK(hipMalloc(&bits, N / 8)); CHECK(hipMalloc(&bo, N / 8));
  { std::vector<uint32_t> h(N / 32); for (uint32_t k = 0; k < N / 32; ++k) h[k] = hash32(k * 977 + 5); CHECK(hipMemcpy(bits, h.data(), N / 8, hipMemcpyHostToDevice)); }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)(((uint64_t)M * 4 + 255) / 256);
  float best = 1e30f, sum = 0.f;
  const int reps = 5, inner = 20;
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k_packed, dim3(grid), dim3(256), dyn_lds, 0, q0, q1, bits, bo, M, q0, q1, nlp, wf);
  CHECK(hipDeviceSynchronize());
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0));
    for (int rr = 0; rr < inner; ++rr) hipLaunchKernelGGL(k_packed, dim3(grid), dim3(256), dyn_lds, 0, q0, q1, bits, bo, M, q0, q1, nlp, wf);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= inner;
    best = ms < best ? ms : best; sum += ms;
  }
  printf("{\"variant\": \"P\", \"positions\": %u, \"prefix\": %u, \"accepted_percent\": %d, \"waves_per_simd_cap\": %d, \"ms_mean\": %.4f, \"ms_best\": %.4f, "
         "\"particles_per_s\": %.4e, \"particles_per_s_best\": %.4e}\n", N, M, wf, occ, sum / reps, best, M / (sum / reps * 1e-3), M / (best * 1e-3));
  return 0;
}

// The same measurement as an in-process entry point (tools/liblayout_bench.so, loaded by bench.py with ctypes): no child
// process, nothing on stdout, every allocation released, errors as a return code.
extern "C" __attribute__((visibility("default"))) int layout_bench_packed(uint32_t N, uint32_t M, int wf, int occ, int reps, int inner,
                                                                        double* ms_mean, double* ms_best) {
#define LB(x) do { if ((x) != hipSuccess) { rc = -1; goto done; } } while (0)
  int rc = 0;
  double *q0 = nullptr, *q1 = nullptr, *nlp = nullptr; uint32_t *bits = nullptr, *bo = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (M < 64 || M > N || (N & 31u) || wf < 0 || wf > 100 || reps < 1 || inner < 1) return -2;
  {
    const size_t dyn_lds = occ > 0 ? (size_t)(160 * 1024 / occ - 1024) / 256 * 256 : 0;
    const size_t bytes = (size_t)N * D * 8;
    const unsigned grid = (unsigned)(((uint64_t)M * 4 + 255) / 256);
    float best = 1e30f, sum = 0.f;
    if (dyn_lds > 64 * 1024) LB(hipFuncSetAttribute((const void*)k_packed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
    LB(hipMalloc(&q0, bytes)); LB(hipMalloc(&q1, bytes)); LB(hipMemset(q0, 0, bytes)); LB(hipMemset(q1, 0, bytes));
    LB(hipMalloc(&nlp, (size_t)N * 16)); LB(hipMemset(nlp, 0, (size_t)N * 16));
    LB(hipMalloc(&bits, N / 8)); LB(hipMalloc(&bo, N / 8));
    { std::vector<uint32_t> h(N / 32); for (uint32_t k = 0; k < N / 32; ++k) h[k] = hash32(k * 977 + 5); LB(hipMemcpy(bits, h.data(), N / 8, hipMemcpyHostToDevice)); }
    LB(hipEventCreate(&e0)); LB(hipEventCreate(&e1));
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k_packed, dim3(grid), dim3(256), dyn_lds, 0, q0, q1, bits, bo, M, q0, q1, nlp, wf);
    LB(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) {
      LB(hipEventRecord(e0, 0));
      for (int rr = 0; rr < inner; ++rr) hipLaunchKernelGGL(k_packed, dim3(grid), dim3(256), dyn_lds, 0, q0, q1, bits, bo, M, q0, q1, nlp, wf);
      LB(hipEventRecord(e1, 0)); LB(hipEventSynchronize(e1));
      float ms; LB(hipEventElapsedTime(&ms, e0, e1)); ms /= inner;
      best = ms < best ? ms : best; sum += ms;
    }
    *ms_mean = sum / reps; *ms_best = best;
  }
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(q0); (void)hipFree(q1); (void)hipFree(nlp); (void)hipFree(bits); (void)hipFree(bo);
  return rc;
#undef LB
}

int main(int argc, char** argv) {
  if (argc >= 4 && std::string(argv[1]) == "--packed") {
    const uint32_t M = (uint32_t)strtoul(argv[2], nullptr, 10);
    const int wf = atoi(argv[3]);
    const uint32_t NN = argc >= 5 ? (uint32_t)strtoul(argv[4], nullptr, 10) : (1u << 22);
    if (M < 64 || M > NN || (NN & 31u) || wf < 0 || wf > 100) { printf("bad arguments\n"); return 2; }
    return packed_only(NN, M, wf, argc >= 6 ? atoi(argv[5]) : 0);
  }
  const uint32_t N = 1u << 22;
  const size_t bytes = (size_t)N * D * 8;
  double *th, *out;
  CHECK(hipMalloc(&th, bytes)); CHECK(hipMalloc(&out, (size_t)N * 4 * 8));
  CHECK(hipMemset(th, 0, bytes));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double useful = (double)N * 3 * D * 8;     // 768 B per particle
  auto run = [&](const char* name, auto launch) {
    for (int w = 0; w < 3; ++w) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("{\"variant\": \"%s\", \"ms\": %.4f, \"useful_GBps\": %.1f, \"particles_per_s\": %.4e}\n", name, ms, useful / ms / 1e6, N / (ms * 1e-3));
  };
  run("A rows f64[N][32], 4 lanes x 8 comps, direct to registers", [&] { hipLaunchKernelGGL(k_rows, dim3(N * 4 / 256), dim3(256), 0, 0, th, N, out); });
  uint32_t* idx; CHECK(hipMalloc(&idx, (size_t)N * 4));
  { std::vector<uint32_t> h(N); for (uint32_t k = 0; k < N; ++k) h[k] = k; CHECK(hipMemcpy(idx, h.data(), (size_t)N * 4, hipMemcpyHostToDevice)); }
  double *nth, *nlp; CHECK(hipMalloc(&nth, bytes)); CHECK(hipMalloc(&nlp, (size_t)N * 16));
  run("A2 rows + two random 4-B index look-ups (16 MB table)", [&] { hipLaunchKernelGGL(k_rows_idx, dim3(N * 4 / 256), dim3(256), 0, 0, th, idx, N, out); });
  run("A3 rows, reads + all row writes + 16 B state, no index", [&] { hipLaunchKernelGGL(k_rows_rw, dim3(N * 4 / 256), dim3(256), 0, 0, th, idx, N, nth, nlp, 0, 100); });
  run("A4 rows, reads + 45% row writes + 16 B state, no index", [&] { hipLaunchKernelGGL(k_rows_rw, dim3(N * 4 / 256), dim3(256), 0, 0, th, idx, N, nth, nlp, 0, 45); });
  run("A5 rows, reads + 45% row writes + 16 B state + index look-ups", [&] { hipLaunchKernelGGL(k_rows_rw, dim3(N * 4 / 256), dim3(256), 0, 0, th, idx, N, nth, nlp, 1, 45); });
  uint32_t* aout; CHECK(hipMalloc(&aout, (size_t)N * 4));
  CHECK(hipMemset(nlp, 0, (size_t)N * 16));
  run("A6 row store: reads + index look-ups + 25% rows written to the other slot, state in place on accept, 4 B alive list", [&] { hipLaunchKernelGGL(k_rows_store, dim3(N * 4 / 256), dim3(256), 0, 0, th, idx, N, nth, nlp, aout, 25, 1); });
  run("A7 row store, 20% accepted", [&] { hipLaunchKernelGGL(k_rows_store, dim3(N * 4 / 256), dim3(256), 0, 0, th, idx, N, nth, nlp, aout, 20, 1); });
  run("A8 row store, 0% accepted (reads + look-ups + 4 B alive list only)", [&] { hipLaunchKernelGGL(k_rows_store, dim3(N * 4 / 256), dim3(256), 0, 0, th, idx, N, nth, nlp, aout, 0, 1); });
  run("A9 row store, 25% accepted, NO index look-ups (physically compacted population)", [&] { hipLaunchKernelGGL(k_rows_store, dim3(N * 4 / 256), dim3(256), 0, 0, th, idx, N, nth, nlp, aout, 25, 0); });
  run("A10 row store, 0% accepted, NO index look-ups", [&] { hipLaunchKernelGGL(k_rows_store, dim3(N * 4 / 256), dim3(256), 0, 0, th, idx, N, nth, nlp, aout, 0, 0); });
  run("A11 as A10 without the 4-B alive-list write", [&] { hipLaunchKernelGGL(k_rows_store, dim3(N * 4 / 256), dim3(256), 0, 0, th, idx, N, nth, nlp, aout, 0, 0, 1); });
  run("A12 as A10 without the 16-B state read", [&] { hipLaunchKernelGGL(k_rows_store, dim3(N * 4 / 256), dim3(256), 0, 0, th, idx, N, nth, nlp, aout, 0, 0, 2); });
  run("A13 as A6 (25% accepted, look-ups) without the 4-B alive-list write", [&] { hipLaunchKernelGGL(k_rows_store, dim3(N * 4 / 256), dim3(256), 0, 0, th, idx, N, nth, nlp, aout, 25, 1, 1); });
  {
    double* st2; CHECK(hipMalloc(&st2, 2 * bytes)); CHECK(hipMemset(st2, 0, 2 * bytes));
    run("A14 as A6 (25% accepted) with the two slots of a particle adjacent in memory (store[N][2][32])", [&] { hipLaunchKernelGGL(k_rows_store_il, dim3(N * 4 / 256), dim3(256), 0, 0, st2, idx, N, st2, nlp, aout, 25); });
    run("A15 as A14 with 0% accepted", [&] { hipLaunchKernelGGL(k_rows_store_il, dim3(N * 4 / 256), dim3(256), 0, 0, st2, idx, N, st2, nlp, aout, 0); });
    CHECK(hipFree(st2));
  }
  {
    double *q0, *q1; uint32_t *bits, *bo;
    CHECK(hipMalloc(&q0, bytes)); CHECK(hipMalloc(&q1, bytes)); CHECK(hipMemset(q0, 0, bytes)); CHECK(hipMemset(q1, 0, bytes));
    CHECK(hipMalloc(&bits, N / 8)); CHECK(hipMalloc(&bo, N / 8));
    { std::vector<uint32_t> h(N / 32); for (uint32_t k = 0; k < N / 32; ++k) h[k] = hash32(k * 977 + 5); CHECK(hipMemcpy(bits, h.data(), N / 8, hipMemcpyHostToDevice)); }
    for (uint32_t M : {N, 3u * (N / 4), N / 2}) {          // the alive prefix shrinks from N to N/2 between two resamplings
      for (int wf : {15, 25, 0}) {
        char name[160];
        snprintf(name, sizeof name, "P packed population: prefix %u of %u positions, %d%% accepted (bitmap look-ups, rows to the other slot)", M, N, wf);
        const uint32_t saveN = N; (void)saveN;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_packed, dim3(M * 4 / 256), dim3(256), 0, 0, q0, q1, bits, bo, M, q0, q1, nlp, wf);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int rr = 0; rr < 20; ++rr) hipLaunchKernelGGL(k_packed, dim3(M * 4 / 256), dim3(256), 0, 0, q0, q1, bits, bo, M, q0, q1, nlp, wf);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("{\"variant\": \"%s\", \"ms\": %.4f, \"useful_GBps\": %.1f, \"particles_per_s\": %.4e}\n", name, ms, (double)M * 768 / ms / 1e6, M / (ms * 1e-3));
      }
    }
    {
      const uint32_t M = 3u * (N / 4);
      auto runq = [&](const char* nm, auto kern) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(M * 4 / 256), dim3(256), 0, 0, q0, q1, bits, bo, M, q0, q1, nlp, 15);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int rr = 0; rr < 20; ++rr) hipLaunchKernelGGL(kern, dim3(M * 4 / 256), dim3(256), 0, 0, q0, q1, bits, bo, M, q0, q1, nlp, 15);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("{\"variant\": \"Q packed, prefix 3/4 N, 15%% accepted, %s\", \"ms\": %.4f, \"particles_per_s\": %.4e}\n", nm, ms, M / (ms * 1e-3));
      };
      runq("plain loads and stores", k_packed_nt<0>);
      runq("non-temporal row stores", k_packed_nt<1>);
      runq("non-temporal own-row loads", k_packed_nt<2>);
      runq("non-temporal own-row loads + row stores", k_packed_nt<3>);
      runq("non-temporal donor loads", k_packed_nt<4>);
      runq("all non-temporal", k_packed_nt<7>);
    }
    CHECK(hipFree(q0)); CHECK(hipFree(q1)); CHECK(hipFree(bits)); CHECK(hipFree(bo));
  }
  run("B component-major f64[32][N], thread per particle", [&] { hipLaunchKernelGGL(k_soa, dim3(N / 256), dim3(256), 0, 0, th, N, out); });
  run("C rows f64[N][32] staged through LDS per workgroup", [&] { hipLaunchKernelGGL(k_rows_lds, dim3(N / 64), dim3(256), 0, 0, th, N, out); });
  return 0;
}
