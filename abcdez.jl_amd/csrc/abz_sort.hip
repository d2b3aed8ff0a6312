/*
 * abz_sort.hip -- (distance, index)-sorted order for abcdemc's "better particle" draw
 * (src/abcdez_mc.jl:23).  A plain library sort: rocPRIM's stable LSD radix sort on the
 * order-preserving bit pattern of the distance with the particle index as payload,
 * so ties keep index order.
 */
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "abz_ctx.h"
#include "abz_device.h"

__global__ __launch_bounds__(ABZ_BLOCK) void sort_keys_kernel(const double* __restrict__ delta, uint32_t n,
                                                              unsigned long long* __restrict__ keys,
                                                              uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  if (i >= n) return;
  const unsigned long long u = abz_d2u(delta[i]);
  keys[i] = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
  vals[i] = i;
}
__global__ __launch_bounds__(ABZ_BLOCK) void sort_unkey_kernel(const unsigned long long* __restrict__ keys, uint32_t n,
                                                               double* __restrict__ sorted) {
  const uint32_t i = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  sorted[i] = (k >> 63) ? abz_u2d(k & 0x7FFFFFFFFFFFFFFFull) : abz_u2d(~k);
}

int abz_rank_prepare_impl(abcdez_ctx* ctx, const double* delta, int64_t N, uint32_t* order, double* sorted_delta) {
  const uint32_t n = (uint32_t)N;
  size_t temp_bytes = 0;
  ABZ_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, temp_bytes, (unsigned long long*)nullptr,
                                          (unsigned long long*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, n, 0,
                                          64, ctx->stream));
  const size_t kb = abz_align((size_t)n * 8), vb = abz_align((size_t)n * 4);
  int rc = abz_ws_reserve(ctx, 2 * kb + vb + abz_align(temp_bytes));
  if (rc) return rc;
  char* p = (char*)ctx->ws;
  unsigned long long* keys_in = (unsigned long long*)p; p += kb;
  unsigned long long* keys_out = (unsigned long long*)p; p += kb;
  uint32_t* vals_in = (uint32_t*)p; p += vb;
  void* temp = p;
  const unsigned grid = (n + ABZ_BLOCK - 1) / ABZ_BLOCK;
  hipLaunchKernelGGL(sort_keys_kernel, dim3(grid), dim3(ABZ_BLOCK), 0, ctx->stream, delta, n, keys_in, vals_in);
  ABZ_HIP_CHECK(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, order, n, 0, 64, ctx->stream));
  hipLaunchKernelGGL(sort_unkey_kernel, dim3(grid), dim3(ABZ_BLOCK), 0, ctx->stream, keys_out, n, sorted_delta);
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}
