// Issue cost of the vector instructions the sweep kernels are made of (fp64 add / mul / fma, 32x32->64 integer multiply,
// xor3, fp64 reciprocal), measured as time per wave-instruction per SIMD with every SIMD saturated (8 waves, 8 independent
// chains per lane).   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096, CH = 8;
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double seed, uint32_t useed) {
  double x[CH]; uint64_t u[CH]; uint32_t w[CH];
  for (int c = 0; c < CH; ++c) { x[c] = seed + c * 1e-3 + threadIdx.x * 1e-6; u[c] = useed + c * 977u + threadIdx.x; w[c] = useed * 31u + c; }
  const double a = seed * 0.999, b = seed * 1e-9;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (OP == 0) x[c] = x[c] + a;
      if (OP == 1) x[c] = x[c] * a;
      if (OP == 2) x[c] = __builtin_fma(x[c], a, b);
      if (OP == 3) u[c] = (uint64_t)(uint32_t)u[c] * 0xD2511F53u + (u[c] >> 32);      // v_mad_u64_u32
      if (OP == 4) w[c] = __builtin_amdgcn_bitop3_b32(w[c], (uint32_t)it, useed, 0x96);
      if (OP == 5) x[c] = __builtin_amdgcn_rcp(x[c]);
      if (OP == 6) w[c] = __umulhi(w[c], 0xCD9E8D57u) + 1u;                              // v_mul_hi_u32 (+ add)
      if (OP == 7) x[c] = __builtin_amdgcn_ldexp(x[c], 1);
    }
  }
  double s = 0; for (int c = 0; c < CH; ++c) s += x[c] + (double)u[c] + (double)w[c];
  if (s == 1.2345e301) out[threadIdx.x] = s;
}
int main() {
  double* out; CK(hipMalloc(&out, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const char* names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_mad_u64_u32", "v_bitop3_b32", "v_rcp_f64", "v_mul_hi_u32 + v_add_u32", "v_ldexp_f64"};
  for (int op = 0; op < 8; ++op) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      const dim3 g(cus * 8), b(256);          // 8 blocks of 4 waves per CU = 8 waves per SIMD
      switch (op) {
        case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, 1.0000001, 12345u); break;
        case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, 1.0000001, 12345u); break;
        case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, 1.0000001, 12345u); break;
        case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, 1.0000001, 12345u); break;
        case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, out, 1.0000001, 12345u); break;
        case 5: hipLaunchKernelGGL(k<5>, g, b, 0, 0, out, 1.0000001, 12345u); break;
        case 6: hipLaunchKernelGGL(k<6>, g, b, 0, 0, out, 1.0000001, 12345u); break;
        case 7: hipLaunchKernelGGL(k<7>, g, b, 0, 0, out, 1.0000001, 12345u); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    const double per_simd = 8.0 * ITER * CH;                 // wave-instructions of this type per SIMD
    printf("{\"instruction\": \"%s\", \"ns_per_wave_instruction_per_simd\": %.3f, \"cycles_at_2.4GHz\": %.2f}\n", names[op],
           best * 1e6 / per_simd, best * 1e6 / per_simd * 2.4);
  }
  return 0;
}
