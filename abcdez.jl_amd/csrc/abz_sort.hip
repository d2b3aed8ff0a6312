/*
 * abz_sort.hip -- the enumeration abcdemc's "better particle" draw indexes into
 * (s = rand(rng, (1:N)[Ds .<= Ds[i]]), src/abcdez_mc.jl:23).
 *
 * order[0 .. nA)  : the particles with Ds <= eps_pop, in index order (they belong to the candidate set of EVERY
 *                   particle that draws -- a draw happens only for Ds[i] > eps_pop, mc:19-20 -- so their relative
 *                   order is immaterial and a stable partition is enough);
 * order[nA .. N)  : the others sorted by (Ds, index);
 * sorted_delta[p] = max(Ds[order[p]], eps_pop): non-decreasing, so the candidate set of particle i is
 *                   order[0 .. upper_bound(sorted_delta, Ds[i])) exactly as in the reference's mask;
 * cnt[i]          = that upper bound, #{j : Ds[j] <= Ds[i]}, for every particle outside the first block (the sweep
 *                   reads it coalesced instead of searching sorted_delta: 20 dependent loads per particle).
 *
 * Hand-written for this path instead of a library sort of 64-bit keys:
 *   1. every particle gets a 24-bit BUCKET id: 0 for Ds <= eps_pop, else 1 + ((key - key(eps_pop) - 1) >> shift),
 *      clamped -- a monotone binning of the order-preserving bit pattern over the window (eps_pop, max Ds] the
 *      driver already knows (mc:146); at N = 2^20 a bucket holds 0.06 particles on average;
 *   2. a stable LSD radix sort of (bucket, index) pairs, three passes of 8 bits; one wavefront owns a tile and ranks
 *      its elements with ballots (no atomics anywhere, so the result is deterministic);
 *   3. a fix-up pass puts the few buckets that hold more than one distinct distance into (Ds, index) order.
 * Any binning is correct (the fix-up sorts whatever shares a bucket); a fitting one is fast.
 */
#include <hip/hip_runtime.h>
#include <string.h>

#include "abz_ctx.h"
#include "abz_device.h"

#define MCR_BITS 24
#define MCR_ROUND 64                    /* one wave-round */
#define MCR_WAVES (ABZ_BLOCK / 64)
#define MCR_BATCH 8                     /* rounds whose loads are issued together */

__device__ inline uint32_t mcr_bucket(double d, unsigned long long klo, int shift) {
  const unsigned long long key = f64_order_key(d);
  if (key <= klo) return 0u;
  const unsigned long long t = (key - klo - 1ull) >> shift;
  return t < (1ull << MCR_BITS) - 2ull ? (uint32_t)t + 1u : (1u << MCR_BITS) - 1u;
}

/* tile histogram of one 8-bit digit.  PASS 0 also makes the (bucket, index) pairs from the distances.
 * table[digit * ntiles + tile]; one wave per tile of `rounds` x 64 consecutive elements.                 */
template <int PASS>
__global__ __launch_bounds__(ABZ_BLOCK) void mcr_hist_kernel(const double* __restrict__ delta, uint32_t n,
                                                             unsigned long long klo, int shift,
                                                             uint32_t* __restrict__ key, uint32_t* __restrict__ val,
                                                             uint32_t* __restrict__ table, uint32_t ntiles,
                                                             uint32_t rounds, const unsigned long long* __restrict__ win) {
  __shared__ uint32_t s_h[MCR_WAVES][256];
  if (PASS == 0 && win) { klo = win[ABZ_S_MCW_KLO - ABZ_S_MCW_EPS]; shift = (int)win[ABZ_S_MCW_SHIFT - ABZ_S_MCW_EPS]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x * MCR_WAVES + wave;
  for (int b = lane; b < 256; b += 64) s_h[wave][b] = 0u;
  __builtin_amdgcn_wave_barrier();
  if (tile < ntiles) {
    const uint64_t base = (uint64_t)tile * rounds * MCR_ROUND;
    for (uint32_t r0 = 0; r0 < rounds; r0 += MCR_BATCH) {        /* rounds is a multiple of MCR_BATCH */
      uint32_t k[MCR_BATCH];
      bool in[MCR_BATCH];
#pragma unroll
      for (int u = 0; u < MCR_BATCH; ++u) {                      /* MCR_BATCH independent loads in flight */
        const uint64_t i = base + (uint64_t)(r0 + u) * MCR_ROUND + lane;
        in[u] = i < n;
        if constexpr (PASS == 0) {
          k[u] = in[u] ? mcr_bucket(delta[i], klo, shift) : 0u;
        } else {
          k[u] = in[u] ? key[i] : 0u;
        }
      }
#pragma unroll
      for (int u = 0; u < MCR_BATCH; ++u) {
        if (in[u]) {
          if constexpr (PASS == 0) {
            const uint64_t i = base + (uint64_t)(r0 + u) * MCR_ROUND + lane;
            key[i] = k[u];
            val[i] = (uint32_t)i;
          }
        }
        /* bucket 0 (Ds <= eps_pop: most of the population late in a run) has digit 0 in every pass: counted with one
         * ballot instead of up to 64 colliding LDS atomics */
        const unsigned long long z = __ballot(in[u] && k[u] == 0u);
        if (lane == 0 && z) atomicAdd(&s_h[wave][0], (uint32_t)__popcll(z));
        if (in[u] && k[u] != 0u) atomicAdd(&s_h[wave][(k[u] >> (8 * PASS)) & 255u], 1u);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (tile < ntiles)
    for (int b = lane; b < 256; b += 64) table[(size_t)b * ntiles + tile] = s_h[wave][b];
}

/* offsets, step 1: block d turns the tile counts of digit d (contiguous in the digit-major table: coalesced) into
 * their exclusive prefix over the tiles and leaves the digit's total in totals[d].  Step 2 -- the exclusive scan of
 * the 256 totals -- is done by every scattering wave for itself (mcr_scatter_kernel).  256 small blocks instead of
 * one block walking the whole table with a 512-byte stride (that cost 0.2 ms per pass).                        */
__global__ __launch_bounds__(ABZ_BLOCK) void mcr_scan_kernel(uint32_t* __restrict__ table, uint32_t ntiles,
                                                             uint32_t* __restrict__ totals) {
  __shared__ uint32_t s_part[ABZ_BLOCK];
  uint32_t* v = table + (size_t)blockIdx.x * ntiles;
  const uint32_t t = threadIdx.x;
  const uint32_t per = (ntiles + ABZ_BLOCK - 1) / ABZ_BLOCK;
  const uint32_t lo = t * per < ntiles ? t * per : ntiles, hi = lo + per < ntiles ? lo + per : ntiles;
  uint32_t s = 0;
  for (uint32_t k = lo; k < hi; ++k) s += v[k];
  s_part[t] = s;
  __syncthreads();
  for (uint32_t off = 1; off < ABZ_BLOCK; off <<= 1) {
    const uint32_t add = t >= off ? s_part[t - off] : 0;
    __syncthreads();
    s_part[t] += add;
    __syncthreads();
  }
  uint32_t run = t ? s_part[t - 1] : 0;
  for (uint32_t k = lo; k < hi; ++k) { const uint32_t c = v[k]; v[k] = run; run += c; }
  if (t == ABZ_BLOCK - 1) totals[blockIdx.x] = s_part[ABZ_BLOCK - 1];
}

/* stable scatter of one digit.  The wave walks its tile in index order, 64 elements per round; lanes holding the
 * same digit find each other with 8 ballots, take consecutive slots behind the digit's running offset (LDS, private
 * to the wave) and the last of them advances it.  LAST: the pairs end as order[] / bucket[] and the clamped
 * distance of every position is gathered.                                                               */
template <int PASS, bool LAST>
__global__ __launch_bounds__(ABZ_BLOCK) void mcr_scatter_kernel(const uint32_t* __restrict__ key_in,
                                                                const uint32_t* __restrict__ val_in, uint32_t n,
                                                                const uint32_t* __restrict__ table,
                                                                const uint32_t* __restrict__ totals, uint32_t ntiles,
                                                                uint32_t rounds, uint32_t* __restrict__ key_out,
                                                                uint32_t* __restrict__ val_out,
                                                                const double* __restrict__ delta, double eps_pop,
                                                                double* __restrict__ sorted_delta,
                                                                uint32_t* __restrict__ cnt_of,
                                                                const unsigned long long* __restrict__ win) {
  __shared__ uint32_t s_run[MCR_WAVES][256];
  if (LAST && win) eps_pop = abz_u2d(win[0]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x * MCR_WAVES + wave;
  if (tile >= ntiles) return;                         /* whole waves leave; no block-level barrier below */
  volatile uint32_t* run = s_run[wave];
  {   /* start of digit d = exclusive scan of the 256 digit totals (lane l owns digits 4l .. 4l+3) + this tile's prefix */
    const uint32_t t0 = totals[4 * lane], t1 = totals[4 * lane + 1], t2 = totals[4 * lane + 2], t3 = totals[4 * lane + 3];
    uint32_t inc = t0 + t1 + t2 + t3;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    const uint32_t s0 = inc - (t0 + t1 + t2 + t3);
    run[4 * lane] = s0 + table[(size_t)(4 * lane) * ntiles + tile];
    run[4 * lane + 1] = s0 + t0 + table[(size_t)(4 * lane + 1) * ntiles + tile];
    run[4 * lane + 2] = s0 + t0 + t1 + table[(size_t)(4 * lane + 2) * ntiles + tile];
    run[4 * lane + 3] = s0 + t0 + t1 + t2 + table[(size_t)(4 * lane + 3) * ntiles + tile];
  }
  __builtin_amdgcn_wave_barrier();
  const uint64_t base = (uint64_t)tile * rounds * MCR_ROUND;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (uint32_t r0 = 0; r0 < rounds; r0 += MCR_BATCH) {          /* rounds is a multiple of MCR_BATCH */
    uint32_t kk[MCR_BATCH], vv[MCR_BATCH];
#pragma unroll
    for (int u = 0; u < MCR_BATCH; ++u) {                        /* the batch's loads go out together ... */
      const uint64_t i = base + (uint64_t)(r0 + u) * MCR_ROUND + lane;
      kk[u] = i < n ? key_in[i] : 0u;
      vv[u] = i < n ? val_in[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < MCR_BATCH; ++u) {                        /* ... the rounds are ranked one after the other */
      const uint64_t i = base + (uint64_t)(r0 + u) * MCR_ROUND + lane;
      const bool valid = i < n;
      const uint32_t k = kk[u], v = vv[u];
      const uint32_t d = (k >> (8 * PASS)) & 255u;
      unsigned long long same = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const bool one = (d >> bit) & 1u;
        const unsigned long long bal = __ballot(one);
        same &= one ? bal : ~bal;
      }
      const uint32_t rank = (uint32_t)__popcll(same & below), cnt = (uint32_t)__popcll(same);
      const uint32_t pos = run[d] + rank;             /* every lane of the group reads before its last lane writes */
      __builtin_amdgcn_wave_barrier();
      if (valid) {
        key_out[pos] = k;
        val_out[pos] = v;
        if constexpr (LAST) {
          const double x = delta[v];
          sorted_delta[pos] = k == 0u ? eps_pop : x;
          cnt_of[v] = pos + 1u;                       /* right for a bucket of one; shared buckets: mcr_fixup_kernel */
        }
        if (rank + 1u == cnt) run[d] = pos + 1u;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

/* buckets holding several particles: put them into (Ds, index) order.  One thread per run start; runs of one
 * element (nearly all) and runs of equal distances (atoms of a discrete distance: the stable sort already left them
 * in index order) cost one pass over the run; bucket 0 (Ds <= eps_pop) keeps its index order by definition.   */
__global__ __launch_bounds__(ABZ_BLOCK) void mcr_fixup_kernel(const uint32_t* __restrict__ bucket, uint32_t n,
                                                              uint32_t* __restrict__ order,
                                                              double* __restrict__ sorted_delta,
                                                              uint32_t* __restrict__ cnt) {
  const uint32_t p = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  if (p >= n) return;
  const uint32_t b = bucket[p];
  if (b == 0u || (p > 0u && bucket[p - 1u] == b)) return;
  uint32_t e = p + 1u;
  while (e < n && bucket[e] == b) ++e;
  if (e == p + 1u) return;                            /* a bucket of one: nothing to do */
  for (uint32_t q = p + 1u; q < e; ++q) {             /* insertion sort: linear on sorted input */
    const double x = sorted_delta[q];
    const uint32_t ix = order[q];
    uint32_t at = q;
    while (at > p) {
      const double y = sorted_delta[at - 1u];
      const uint32_t iy = order[at - 1u];
      if (y < x || (y == x && iy < ix)) break;
      sorted_delta[at] = y;
      order[at] = iy;
      --at;
    }
    if (at != q) { sorted_delta[at] = x; order[at] = ix; }
  }
  /* cnt of a particle = end of its run of EQUAL distances (upper bound), walking the sorted bucket from the back */
  uint32_t end = e;
  for (uint32_t q = e; q-- > p;) {
    if (q + 1u < e && sorted_delta[q] != sorted_delta[q + 1u]) end = q + 1u;
    cnt[order[q]] = end;
  }
}

static inline unsigned long long host_order_key(double x) {
  unsigned long long u;
  memcpy(&u, &x, 8);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

/* The generation's eps_pop (mc:147) and the binning window of the rank pass, made ON THE DEVICE from the extrema the
 * previous sweep left in its min / max bank (bank >= 0) or from host values (bank < 0): abcdez_mc_generation_async.
 * eps_pop = max(eps_target, lo + alpha (hi - lo)) with the host driver's operations (two roundings, `b > a ? b : a`). */
__device__ inline void mc_window_write(unsigned long long* __restrict__ scal, double lo, double hi, double alpha, double eps_target) {
  const double v = lo + alpha * (hi - lo);
  const double eps_pop = v > eps_target ? v : eps_target;
  const unsigned long long klo = f64_order_key(eps_pop);
  unsigned long long khi = f64_order_key(hi);
  if (!(hi > eps_pop)) khi = klo + 1ull;
  const unsigned long long range = khi - klo;
  int bits = 0;
  while (bits < 64 && (range >> bits) != 0ull) ++bits;
  unsigned long long* w = scal + ABZ_S_MCW_EPS;
  w[0] = abz_d2u(eps_pop);
  w[ABZ_S_MCW_KLO - ABZ_S_MCW_EPS] = klo;
  w[ABZ_S_MCW_SHIFT - ABZ_S_MCW_EPS] = (unsigned long long)(bits > MCR_BITS ? bits - MCR_BITS : 0);
  w[ABZ_S_MCW_LO - ABZ_S_MCW_EPS] = abz_d2u(lo);
  w[ABZ_S_MCW_HI - ABZ_S_MCW_EPS] = abz_d2u(hi);
}
__global__ __launch_bounds__(64) void mc_window_kernel(unsigned long long* __restrict__ scal, int bank, double lo_h, double hi_h,
                                                       double alpha, double eps_target) {
  double lo = lo_h, hi = hi_h;
  if (bank >= 0) {
    const unsigned long long* m = scal + ABZ_S_MM0 + (size_t)bank * 2 * ABZ_MMSLOTS;
    unsigned long long mn = m[2 * threadIdx.x], mx = m[2 * threadIdx.x + 1];      /* ABZ_MMSLOTS == 64 */
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned long long a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
      mn = a < mn ? a : mn;
      mx = b > mx ? b : mx;
    }
    lo = f64_from_order_key_dev(mn);
    hi = f64_from_order_key_dev(mx);
  }
  if (threadIdx.x != 0) return;
  mc_window_write(scal, lo, hi, alpha, eps_target);
}
int abz_launch_mc_window(abcdez_ctx* ctx, int bank, double lo, double hi, double alpha, double eps_target) {
  static_assert(ABZ_MMSLOTS == 64, "mc_window_kernel folds one slot per lane");
  hipLaunchKernelGGL(mc_window_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_scal, bank, lo, hi, alpha, eps_target);
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

/* What the driver reads of a generation (mc:156, mc:146, nsims), folded on the device and written STRAIGHT into pinned
 * host memory: totals of the two cumulative counter classes, extrema of the bank the sweep reduced into, eps_pop; the
 * ticket word last, behind a system-scope fence -- the host polls it (no copy engine, no event, no stream synchronisation) */
/* It also prepares the NEXT generation: the extrema it has just folded are that generation's mc:146, so eps_pop and the
 * binning window (mc_window_write) are made here and the next generation starts without a window launch.            */
__global__ __launch_bounds__(ABZ_CSLOTS) void mc_snapshot_kernel(unsigned long long* __restrict__ scal, int bank,
                                                                 unsigned long long* __restrict__ out, unsigned long long seq,
                                                                 double alpha, double eps_target) {
  __shared__ unsigned long long s_g[ABZ_CSLOTS / 64], s_s[ABZ_CSLOTS / 64];
  const unsigned long long* cs = scal + ABZ_S_CSLOT0 + (size_t)threadIdx.x * ABZ_CSTRIDE;
  unsigned long long vg = cs[ABZ_C_MCGT], vs = cs[ABZ_C_MCSIM];
  unsigned long long mn = ~0ull, mx = 0ull;
  if (threadIdx.x < ABZ_MMSLOTS) {
    const unsigned long long* m = scal + ABZ_S_MM0 + (size_t)bank * 2 * ABZ_MMSLOTS;
    mn = m[2 * threadIdx.x]; mx = m[2 * threadIdx.x + 1];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    vg += __shfl_xor(vg, o); vs += __shfl_xor(vs, o);
    const unsigned long long a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0) { s_g[threadIdx.x >> 6] = vg; s_s[threadIdx.x >> 6] = vs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tg = 0, ts = 0;
    for (int w = 0; w < ABZ_CSLOTS / 64; ++w) { tg += s_g[w]; ts += s_s[w]; }
    out[0] = tg; out[1] = ts; out[2] = mn; out[3] = mx;        /* wave 0 holds the bank's extrema */
    out[4] = scal[ABZ_S_MCW_EPS];
    __threadfence_system();
    __hip_atomic_store(out + ABZ_RING_WORDS - 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    mc_window_write(scal, f64_from_order_key_dev(mn), f64_from_order_key_dev(mx), alpha, eps_target);
  }
}
int abz_launch_mc_snapshot(abcdez_ctx* ctx, int bank, unsigned long long* d_slot, unsigned long long seq, double alpha,
                           double eps_target) {
  static_assert(ABZ_MMSLOTS <= 64, "mc_snapshot_kernel reduces the bank in wave 0");
  hipLaunchKernelGGL(mc_snapshot_kernel, dim3(1), dim3(ABZ_CSLOTS), 0, ctx->stream, ctx->d_scal, bank, d_slot, seq, alpha, eps_target);
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

/* win: NULL = (eps_pop, dmax_hint) are host values; else the device window mc_window_kernel wrote (the two host values are ignored) */
int abz_rank_prepare_impl(abcdez_ctx* ctx, const double* delta, int64_t N, double eps_pop, double dmax_hint,
                          uint32_t* order, double* sorted_delta, uint32_t* cnt, const unsigned long long* win) {
  const uint32_t n = (uint32_t)N;
  /* window of the binning: (eps_pop, dmax_hint] in key space -> 2^24 - 2 buckets */
  const unsigned long long klo = host_order_key(eps_pop);
  unsigned long long khi = host_order_key(dmax_hint);
  if (!(dmax_hint > eps_pop)) khi = klo + 1ull;          /* also catches a NaN hint */
  const unsigned long long range = khi - klo;
  int bits = 0;
  while (bits < 64 && (range >> bits) != 0ull) ++bits;   /* range < 2^bits */
  const int shift = bits > MCR_BITS ? bits - MCR_BITS : 0;
  /* one wave per tile of rounds x 64 elements; short tiles = many waves in flight (each walks its rounds one after the
   * other), at most 4096 tiles so that the count table stays small */
  uint32_t rounds = MCR_BATCH;
  while ((uint64_t)rounds * MCR_ROUND * 4096ull < (uint64_t)n) rounds *= 2;
  const uint32_t ntiles = (uint32_t)(((uint64_t)n + (uint64_t)rounds * MCR_ROUND - 1) / ((uint64_t)rounds * MCR_ROUND));
  const size_t pb = abz_align((size_t)n * 4), tb = abz_align((size_t)256 * ntiles * 4 + 256 * 4);
  int rc = abz_ws_reserve(ctx, 3 * pb + tb);
  if (rc) return rc;
  char* w = (char*)ctx->ws;
  uint32_t* keyA = (uint32_t*)w; w += pb;
  uint32_t* valA = (uint32_t*)w; w += pb;
  uint32_t* keyB = (uint32_t*)w; w += pb;
  uint32_t* table = (uint32_t*)w;
  uint32_t* totals = table + (size_t)256 * ntiles;
  uint32_t* valB = order;                                /* pass 0 -> (keyB, order), pass 1 -> (keyA, valA), pass 2 -> (keyB, order) */
  const unsigned grid = (ntiles + MCR_WAVES - 1) / MCR_WAVES;
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL((mcr_hist_kernel<0>), dim3(grid), dim3(ABZ_BLOCK), 0, st, delta, n, klo, shift, keyA, valA, table, ntiles, rounds, win);
  hipLaunchKernelGGL(mcr_scan_kernel, dim3(256), dim3(ABZ_BLOCK), 0, st, table, ntiles, totals);
  hipLaunchKernelGGL((mcr_scatter_kernel<0, false>), dim3(grid), dim3(ABZ_BLOCK), 0, st, keyA, valA, n, table, totals, ntiles, rounds,
                     keyB, valB, delta, eps_pop, sorted_delta, cnt, win);
  hipLaunchKernelGGL((mcr_hist_kernel<1>), dim3(grid), dim3(ABZ_BLOCK), 0, st, delta, n, klo, shift, keyB, valB, table, ntiles, rounds, win);
  hipLaunchKernelGGL(mcr_scan_kernel, dim3(256), dim3(ABZ_BLOCK), 0, st, table, ntiles, totals);
  hipLaunchKernelGGL((mcr_scatter_kernel<1, false>), dim3(grid), dim3(ABZ_BLOCK), 0, st, keyB, valB, n, table, totals, ntiles, rounds,
                     keyA, valA, delta, eps_pop, sorted_delta, cnt, win);
  hipLaunchKernelGGL((mcr_hist_kernel<2>), dim3(grid), dim3(ABZ_BLOCK), 0, st, delta, n, klo, shift, keyA, valA, table, ntiles, rounds, win);
  hipLaunchKernelGGL(mcr_scan_kernel, dim3(256), dim3(ABZ_BLOCK), 0, st, table, ntiles, totals);
  hipLaunchKernelGGL((mcr_scatter_kernel<2, true>), dim3(grid), dim3(ABZ_BLOCK), 0, st, keyA, valA, n, table, totals, ntiles, rounds,
                     keyB, order, delta, eps_pop, sorted_delta, cnt, win);
  hipLaunchKernelGGL(mcr_fixup_kernel, dim3((n + ABZ_BLOCK - 1) / ABZ_BLOCK), dim3(ABZ_BLOCK), 0, st, keyB, n, order,
                     sorted_delta, cnt);
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}
