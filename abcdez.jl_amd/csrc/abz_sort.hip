/*
 * abz_sort.hip -- the enumeration abcdemc's "better particle" draw indexes into
 * (s = rand(rng, (1:N)[Ds .<= Ds[i]]), src/abcdez_mc.jl:23).
 *
 * order[0 .. nA)  : the particles with Ds <= eps_pop, in index order (they belong to the candidate set of EVERY
 *                   particle that draws -- a draw happens only for Ds[i] > eps_pop, mc:19-20 -- so their relative
 *                   order is immaterial and a stable partition is enough);
 * order[nA .. N)  : the others -- the TAIL: the particles that draw -- sorted by (Ds, index);
 * sorted_delta[p] = max(Ds[order[p]], eps_pop): non-decreasing, so the candidate set of particle i is
 *                   order[0 .. upper_bound(sorted_delta, Ds[i])) exactly as in the reference's mask;
 * cnt[i]          = that upper bound, #{j : Ds[j] <= Ds[i]}, for every particle of the tail (the sweep reads it
 *                   coalesced instead of searching sorted_delta: 20 dependent loads per particle).
 *
 * RANK ONLY WHO DRAWS.  abcdemc! passes alpha = 0 to mc:147, so eps_pop = max(eps_target, min Ds): the head is the converged
 * particles, the tail everybody else -- 95 % of the population in the first generation of a run, a few per cent after forty, a
 * handful at the end (17 % on average over the 100 generations of BASELINE configs[1]).  The head needs a compaction, not a sort:
 *   1. mcr_count / mcr_split: tail flags by wave ballots and per-tile counts, then -- every workgroup adds up the counts in front of
 *      its tiles itself -- every head particle goes to order[i - #tail before i] and every tail particle's (order key, index)
 *      pair to a compact list: two launches, the population's distances read twice (no atomics: deterministic);
 *   2. a tail of up to MCR_SMALL pairs is sorted by ONE workgroup in LDS (bitonic network on (key, index)), which also
 *      finds every particle's upper bound among equal distances -- one launch;
 *   3. a longer tail (most generations of a run) goes through the
 *      stable LSD radix sort of (24-bit bucket, index) pairs + fix-up of round 2, now over the compact list only; its
 *      launches are sized from the tail the host last saw (the kernels stride over their tiles, so any size is
 *      correct) and return at once when step 2 has done the work.
 */
#include <hip/hip_runtime.h>
#include <string.h>

#include "abz_ctx.h"
#include "abz_device.h"

#define MCR_BITS 24
#define MCR_ROUND 64                    /* one wave-round */
#define MCR_WAVES (ABZ_BLOCK / 64)
#define MCR_BATCH 8                     /* rounds whose loads are issued together */
#define MCR_SMALL 4096                  /* tail pairs one workgroup sorts in LDS */
#define MCR_SMALL_THREADS 1024
#define MCR_TILE (MCR_BATCH * MCR_ROUND)   /* elements per wave-tile of the LSD passes over the tail */

/* state of one rank pass, in the workspace: [0] n_tail, [1] n_head */
#define MCR_ST_NTAIL 0
#define MCR_ST_NHEAD 1

__device__ inline uint32_t mcr_bucket_of_key(unsigned long long key, unsigned long long klo, int shift) {   /* key > klo */
  const unsigned long long t = (key - klo - 1ull) >> shift;
  return t < (1ull << MCR_BITS) - 2ull ? (uint32_t)t + 1u : (1u << MCR_BITS) - 1u;
}

/* ---- step 1a: tail particles per tile (one wave per tile of rounds x 64 consecutive particles) */
__global__ __launch_bounds__(ABZ_BLOCK) void mcr_count_kernel(const double* __restrict__ delta, uint32_t n, unsigned long long klo,
                                                              uint32_t* __restrict__ tile_cnt, uint32_t ntiles, uint32_t rounds,
                                                              const unsigned long long* __restrict__ win) {
  if (win) klo = win[ABZ_S_MCW_KLO - ABZ_S_MCW_EPS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x * MCR_WAVES + wave;
  if (tile >= ntiles) return;
  /* launched for a generation that draws by rejection (the host had not seen the switch yet, abz_ctx.h mc_reject_known):
   * nobody will read the enumeration -- an empty tail, and every later kernel of the pass returns at once */
  if (win && abz_mc_draws_by_rejection(win[ABZ_S_MC_NABOVE - ABZ_S_MCW_EPS], (unsigned long long)n)) {
    if (lane == 0) tile_cnt[tile] = 0u;
    return;
  }
  const uint64_t base = (uint64_t)tile * rounds * MCR_ROUND;
  uint32_t c = 0;
  for (uint32_t r0 = 0; r0 < rounds; r0 += MCR_BATCH) {          /* rounds is a multiple of MCR_BATCH */
    double x[MCR_BATCH];
    bool in[MCR_BATCH];
#pragma unroll
    for (int u = 0; u < MCR_BATCH; ++u) {
      const uint64_t i = base + (uint64_t)(r0 + u) * MCR_ROUND + lane;
      in[u] = i < n;
      x[u] = in[u] ? delta[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < MCR_BATCH; ++u) c += (uint32_t)__popcll(__ballot(in[u] && f64_order_key(x[u]) > klo));
  }
  if (lane == 0) tile_cnt[tile] = c;
}

/* ---- step 1b: the head in index order, the tail as a compact list of (order key, index) pairs.  A workgroup adds up the counts of
 * the tiles in front of its own (at most 4096 counts, 16 KB from L2 -- cheaper than the one-workgroup scan launch that used to sit
 * between the two passes); the workgroup that holds the last tile leaves the totals in the state. */
__global__ __launch_bounds__(ABZ_BLOCK) void mcr_split_kernel(const double* __restrict__ delta, uint32_t n, unsigned long long klo,
                                                              double eps_pop, const uint32_t* __restrict__ tile_cnt, uint32_t ntiles,
                                                              uint32_t rounds, uint32_t* __restrict__ order,
                                                              double* __restrict__ sorted_delta, unsigned long long* __restrict__ tk,
                                                              uint32_t* __restrict__ tv, const unsigned long long* __restrict__ win,
                                                              uint32_t* __restrict__ state) {
  __shared__ uint32_t s_pre[MCR_WAVES];
  if (win) { klo = win[ABZ_S_MCW_KLO - ABZ_S_MCW_EPS]; eps_pop = abz_u2d(win[0]); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tile0 = blockIdx.x * MCR_WAVES;
  uint32_t before = 0;
  for (uint32_t k = threadIdx.x; k < tile0; k += ABZ_BLOCK) before += tile_cnt[k];
  for (int off = 32; off; off >>= 1) before += __shfl_xor(before, off, 64);
  if (lane == 0) s_pre[wave] = before;
  __syncthreads();
  before = 0;
  for (int w = 0; w < MCR_WAVES; ++w) before += s_pre[w];                       /* tail particles in front of the workgroup's tiles */
  uint32_t own[MCR_WAVES];
  for (int w = 0; w < MCR_WAVES; ++w) own[w] = tile0 + (uint32_t)w < ntiles ? tile_cnt[tile0 + (uint32_t)w] : 0u;
  if (threadIdx.x == 0 && tile0 + MCR_WAVES >= ntiles) {                        /* the workgroup with the last tile */
    uint32_t tot = before;
    for (int w = 0; w < MCR_WAVES; ++w) tot += own[w];
    state[MCR_ST_NTAIL] = tot; state[MCR_ST_NHEAD] = n - tot;
  }
  if (win && abz_mc_draws_by_rejection(win[ABZ_S_MC_NABOVE - ABZ_S_MCW_EPS], (unsigned long long)n)) return;   /* mcr_count_kernel */
  const uint32_t tile = tile0 + (uint32_t)wave;
  if (tile >= ntiles) return;
  const uint64_t base = (uint64_t)tile * rounds * MCR_ROUND;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t running = before;                         /* tail particles in front of this round (wave-uniform) */
  for (int w = 0; w < wave; ++w) running += own[w];
  for (uint32_t r0 = 0; r0 < rounds; r0 += MCR_BATCH) {
    double x[MCR_BATCH];
    bool in[MCR_BATCH];
#pragma unroll
    for (int u = 0; u < MCR_BATCH; ++u) {
      const uint64_t i = base + (uint64_t)(r0 + u) * MCR_ROUND + lane;
      in[u] = i < n;
      x[u] = in[u] ? delta[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < MCR_BATCH; ++u) {
      const uint32_t i = (uint32_t)(base + (uint64_t)(r0 + u) * MCR_ROUND + lane);
      const unsigned long long key = f64_order_key(x[u]);
      const bool tail = in[u] && key > klo;
      const unsigned long long mt = __ballot(tail);
      const uint32_t tb = running + (uint32_t)__popcll(mt & below);   /* tail particles with a smaller index */
      if (tail) { tk[tb] = key; tv[tb] = i; }
      else if (in[u]) { order[i - tb] = i; sorted_delta[i - tb] = eps_pop; }
      running += (uint32_t)__popcll(mt);
    }
  }
}

/* ---- step 2: a short tail, sorted by (key, index) in LDS by one workgroup; cnt = upper bound among equal distances */
__global__ __launch_bounds__(MCR_SMALL_THREADS) void mcr_tail_small_kernel(const uint32_t* __restrict__ state,
                                                                           const unsigned long long* __restrict__ tk,
                                                                           const uint32_t* __restrict__ tv,
                                                                           uint32_t* __restrict__ order,
                                                                           double* __restrict__ sorted_delta,
                                                                           uint32_t* __restrict__ cnt) {
  __shared__ unsigned long long s_k[MCR_SMALL];
  __shared__ uint32_t s_v[MCR_SMALL];
  __shared__ uint32_t s_e[MCR_SMALL];
  const uint32_t nt = state[MCR_ST_NTAIL], nh = state[MCR_ST_NHEAD];
  if (nt == 0u || nt > MCR_SMALL) return;            /* nothing to sort / the LSD passes do it */
  uint32_t P = 64u;
  while (P < nt) P <<= 1;
  const uint32_t t = threadIdx.x;
  for (uint32_t p = t; p < P; p += MCR_SMALL_THREADS) {
    s_k[p] = p < nt ? tk[p] : ~0ull;                 /* padding sorts behind every real key */
    s_v[p] = p < nt ? tv[p] : 0xFFFFFFFFu;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= P; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t l = t; l < P / 2; l += MCR_SMALL_THREADS) {
        const uint32_t i = ((l & ~(j - 1u)) << 1) | (l & (j - 1u)), q = i | j;
        const unsigned long long ka = s_k[i], kb = s_k[q];
        const uint32_t va = s_v[i], vb = s_v[q];
        const bool gt = ka > kb || (ka == kb && va > vb);
        if (gt == ((i & k) == 0u)) { s_k[i] = kb; s_k[q] = ka; s_v[i] = vb; s_v[q] = va; }
      }
      __syncthreads();
    }
  }
  /* end of every run of equal keys, brought to each of its members by a suffix-minimum scan (in place: a step's partner
   * values wait in registers across the barrier) */
  for (uint32_t p = t; p < P; p += MCR_SMALL_THREADS)
    s_e[p] = (p + 1u >= nt || s_k[p] != s_k[p + 1u]) ? p + 1u : 0xFFFFFFFFu;
  __syncthreads();
  for (uint32_t off = 1; off < P; off <<= 1) {
    uint32_t b[MCR_SMALL / MCR_SMALL_THREADS];
#pragma unroll
    for (uint32_t q = 0; q < MCR_SMALL / MCR_SMALL_THREADS; ++q) {
      const uint32_t p = t + q * MCR_SMALL_THREADS;
      b[q] = (p < P && p + off < P) ? s_e[p + off] : 0xFFFFFFFFu;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < MCR_SMALL / MCR_SMALL_THREADS; ++q) {
      const uint32_t p = t + q * MCR_SMALL_THREADS;
      if (p < P && b[q] < s_e[p]) s_e[p] = b[q];
    }
    __syncthreads();
  }
  for (uint32_t p = t; p < nt; p += MCR_SMALL_THREADS) {
    const uint32_t v = s_v[p];
    order[nh + p] = v;
    sorted_delta[nh + p] = f64_from_order_key_dev(s_k[p]);
    cnt[v] = nh + s_e[p];
  }
}

/* ---- step 3: a long tail -- stable LSD radix sort of (bucket, index) pairs over the compact list.  The kernels read the
 * list's length from the state, stride over their tiles (any grid is correct) and leave when step 2 did the work. */
/* tile histogram of one 8-bit digit.  PASS 0 also makes the buckets from the keys.  table[digit * ntiles + tile]. */
template <int PASS>
__global__ __launch_bounds__(ABZ_BLOCK) void mcr_hist_kernel(const uint32_t* __restrict__ state, const unsigned long long* __restrict__ tk,
                                                             unsigned long long klo, int shift, uint32_t* __restrict__ key,
                                                             uint32_t* __restrict__ table, const unsigned long long* __restrict__ win,
                                                             uint32_t small_limit) {
  __shared__ uint32_t s_h[MCR_WAVES][256];
  const uint32_t n = state[MCR_ST_NTAIL];
  if (n <= small_limit) return;
  if (PASS == 0 && win) { klo = win[ABZ_S_MCW_KLO - ABZ_S_MCW_EPS]; shift = (int)win[ABZ_S_MCW_SHIFT - ABZ_S_MCW_EPS]; }
  const uint32_t ntiles = (n + MCR_TILE - 1) / MCR_TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t tile = blockIdx.x * MCR_WAVES + wave; tile < ntiles; tile += gridDim.x * MCR_WAVES) {
    for (int b = lane; b < 256; b += 64) s_h[wave][b] = 0u;
    __builtin_amdgcn_wave_barrier();
    const uint32_t base = tile * MCR_TILE;
    uint32_t k[MCR_BATCH];
    bool in[MCR_BATCH];
#pragma unroll
    for (int u = 0; u < MCR_BATCH; ++u) {
      const uint32_t i = base + (uint32_t)u * MCR_ROUND + lane;
      in[u] = i < n;
      if constexpr (PASS == 0) k[u] = in[u] ? mcr_bucket_of_key(tk[i], klo, shift) : 0u;
      else k[u] = in[u] ? key[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < MCR_BATCH; ++u) {
      if (in[u]) {
        if constexpr (PASS == 0) key[base + (uint32_t)u * MCR_ROUND + lane] = k[u];
        atomicAdd(&s_h[wave][(k[u] >> (8 * PASS)) & 255u], 1u);
      }
    }
    __builtin_amdgcn_wave_barrier();
    for (int b = lane; b < 256; b += 64) table[(size_t)b * ntiles + tile] = s_h[wave][b];
    __builtin_amdgcn_wave_barrier();
  }
}

/* offsets, step 1: block d turns the tile counts of digit d into their exclusive prefix over the tiles and leaves the digit's
 * total in totals[d].  Step 2 -- the exclusive scan of the 256 totals -- is done by every scattering wave for itself. */
__global__ __launch_bounds__(ABZ_BLOCK) void mcr_scan_kernel(const uint32_t* __restrict__ state, uint32_t* __restrict__ table,
                                                             uint32_t* __restrict__ totals, uint32_t small_limit) {
  __shared__ uint32_t s_part[ABZ_BLOCK];
  const uint32_t n = state[MCR_ST_NTAIL];
  if (n <= small_limit) return;
  const uint32_t ntiles = (n + MCR_TILE - 1) / MCR_TILE;
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

/* stable scatter of one digit.  The wave walks its tile in list order, 64 elements per round; lanes holding the same digit
 * find each other with 8 ballots, take consecutive slots behind the digit's running offset (LDS, private to the wave) and
 * the last of them advances it.  LAST: the pairs end as order[nA + .] / bucket[] with the distance of every position. */
template <int PASS, bool LAST>
__global__ __launch_bounds__(ABZ_BLOCK) void mcr_scatter_kernel(const uint32_t* __restrict__ state, const uint32_t* __restrict__ key_in,
                                                                const uint32_t* __restrict__ val_in,
                                                                const uint32_t* __restrict__ table,
                                                                const uint32_t* __restrict__ totals, uint32_t* __restrict__ key_out,
                                                                uint32_t* __restrict__ val_out,
                                                                const double* __restrict__ delta, uint32_t* __restrict__ order,
                                                                double* __restrict__ sorted_delta, uint32_t* __restrict__ cnt_of,
                                                                uint32_t small_limit) {
  __shared__ uint32_t s_run[MCR_WAVES][256];
  const uint32_t n = state[MCR_ST_NTAIL], nh = state[MCR_ST_NHEAD];
  if (n <= small_limit) return;
  const uint32_t ntiles = (n + MCR_TILE - 1) / MCR_TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  volatile uint32_t* run = s_run[wave];
  const unsigned long long below = (1ull << lane) - 1ull;
  /* start of digit d = exclusive scan of the 256 digit totals (lane l owns digits 4l .. 4l+3) */
  const uint32_t t0 = totals[4 * lane], t1 = totals[4 * lane + 1], t2 = totals[4 * lane + 2], t3 = totals[4 * lane + 3];
  uint32_t inc = t0 + t1 + t2 + t3;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  const uint32_t s0 = inc - (t0 + t1 + t2 + t3);
  for (uint32_t tile = blockIdx.x * MCR_WAVES + wave; tile < ntiles; tile += gridDim.x * MCR_WAVES) {
    run[4 * lane] = s0 + table[(size_t)(4 * lane) * ntiles + tile];          /* + this tile's prefix inside the digit */
    run[4 * lane + 1] = s0 + t0 + table[(size_t)(4 * lane + 1) * ntiles + tile];
    run[4 * lane + 2] = s0 + t0 + t1 + table[(size_t)(4 * lane + 2) * ntiles + tile];
    run[4 * lane + 3] = s0 + t0 + t1 + t2 + table[(size_t)(4 * lane + 3) * ntiles + tile];
    __builtin_amdgcn_wave_barrier();
    const uint32_t base = tile * MCR_TILE;
    uint32_t kk[MCR_BATCH], vv[MCR_BATCH];
#pragma unroll
    for (int u = 0; u < MCR_BATCH; ++u) {                        /* the tile's loads go out together ... */
      const uint32_t i = base + (uint32_t)u * MCR_ROUND + lane;
      kk[u] = i < n ? key_in[i] : 0u;
      vv[u] = i < n ? val_in[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < MCR_BATCH; ++u) {                        /* ... the rounds are ranked one after the other */
      const uint32_t i = base + (uint32_t)u * MCR_ROUND + lane;
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
        if constexpr (LAST) {
          order[nh + pos] = v;
          sorted_delta[nh + pos] = delta[v];
          cnt_of[v] = nh + pos + 1u;                  /* right for a bucket of one; shared buckets: mcr_fixup_kernel */
        } else {
          val_out[pos] = v;
        }
        if (rank + 1u == cnt) run[d] = pos + 1u;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

/* buckets holding several particles: put them into (Ds, index) order.  One thread per run start; runs of one
 * element (nearly all) and runs of equal distances (atoms of a discrete distance: the stable sort already left them
 * in index order) cost one pass over the run.   */
__global__ __launch_bounds__(ABZ_BLOCK) void mcr_fixup_kernel(const uint32_t* __restrict__ state, const uint32_t* __restrict__ bucket,
                                                              uint32_t* __restrict__ order_all,
                                                              double* __restrict__ sorted_delta_all,
                                                              uint32_t* __restrict__ cnt, uint32_t small_limit) {
  const uint32_t n = state[MCR_ST_NTAIL], nh = state[MCR_ST_NHEAD];
  if (n <= small_limit) return;
  uint32_t* order = order_all + nh;
  double* sorted_delta = sorted_delta_all + nh;
  for (uint32_t p = blockIdx.x * ABZ_BLOCK + threadIdx.x; p < n; p += gridDim.x * ABZ_BLOCK) {
    const uint32_t b = bucket[p];
    if (p > 0u && bucket[p - 1u] == b) continue;
    uint32_t e = p + 1u;
    while (e < n && bucket[e] == b) ++e;
    if (e == p + 1u) continue;                          /* a bucket of one: nothing to do */
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
      cnt[order[q]] = nh + end;
    }
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
                                                                 unsigned long long* __restrict__ ring,
                                                                 double alpha, double eps_target, const uint32_t* __restrict__ rank_state,
                                                                 uint32_t rank_limit, uint32_t N, int sharded) {
  /* which generation this is: counted on the device (the host's mc_issued when it enqueued this launch -- or when it replays
   * the graph this launch was captured into): ring slot and ticket follow from it */
  const unsigned long long gen = scal[ABZ_S_MCSEQ];
  unsigned long long* out = ring + (size_t)(gen % ABZ_MC_RING) * ABZ_RING_WORDS;
  const unsigned long long seq = gen + 1ull;
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
    out[7] = 0ull;
    if (sharded) {
      /* the sweep covered this rank's particles only: mc_partial_kernel folded its counts and extrema and the ranks have exchanged them
       * (abz_comm_mc_exchange) -- this generation's counts over the WHOLE population, the global extrema, anybody's fail word */
      const unsigned long long* part = scal + ABZ_S_MC_PART;
      out[7] = 1ull; out[8] = part[5]; out[9] = part[6];
      tg = part[0]; ts = part[1]; mn = part[2]; mx = ~part[3];
      if (~part[4] != 0ull) scal[ABZ_S_MC_REJFAIL] = ~part[4];
    }
    out[0] = tg; out[1] = ts; out[2] = mn; out[3] = mx;        /* wave 0 holds the bank's extrema */
    out[4] = scal[ABZ_S_MCW_EPS];
    out[5] = rank_state ? (unsigned long long)rank_state[MCR_ST_NTAIL] : ~0ull;   /* how many particles drew (sizes the next rank pass) */
    /* fail safe: the rank pass launched ONLY the LDS sort on the strength of a tail bound, and the tail turned out longer -- the
     * enumeration the sweep drew from was not built.  The host turns this word into an error when it redeems the ticket.    */
    out[6] = (rank_state && rank_state[MCR_ST_NTAIL] > rank_limit) ? 1ull : 0ull;
    /* how the sweep just run drew its better particles was decided from #(Ds > eps_target) of its input (ABZ_S_MC_NABOVE);
     * by rank without a rank pass launched (a host that said do_rank = 0 of an unconverged population): same failure */
    const unsigned long long n_in = scal[ABZ_S_MC_NABOVE];
    if (!rank_state && n_in != 0ull && !abz_mc_draws_by_rejection(n_in, (unsigned long long)N)) out[6] = 1ull;
    /* the sweep's own fail word: 1 = drawn by rejection, trials exhausted; 2 = drawn by rank from an enumeration that was not
     * this generation's (no rank pass launched, or one that gave up: the sweep kept s = i for those particles) */
    const unsigned long long rf = scal[ABZ_S_MC_REJFAIL];
    if (rf == 1ull) out[6] = 2ull;
    else if (rf != 0ull && out[6] == 0ull) out[6] = 1ull;
    if (rf != 0ull) scal[ABZ_S_MC_REJFAIL] = 0ull;
    /* ... and the next generation's: what this sweep added to the cumulative ABZ_C_MCGT slots (sharded: all ranks' sweeps, mc_partial_kernel
     * keeps the local baselines) */
    if (sharded) scal[ABZ_S_MC_NABOVE] = tg;
    else { scal[ABZ_S_MC_NABOVE] = tg - scal[ABZ_S_MC_TGPREV]; scal[ABZ_S_MC_TGPREV] = tg; }
    __threadfence_system();
    __hip_atomic_store(out + ABZ_RING_WORDS - 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    mc_window_write(scal, f64_from_order_key_dev(mn), f64_from_order_key_dev(mx), alpha, eps_target);
    scal[ABZ_S_MCSEQ] = seq;
  }
}
/* sharded generations: this rank's part of the reductions, made for the exchange -- counts of THIS sweep (growth of the local slot
 * totals since the last snapshot), local extrema, local fail word (layout: abz_ctx.h, ABZ_S_MC_PART) */
__global__ __launch_bounds__(ABZ_CSLOTS) void mc_partial_kernel(unsigned long long* __restrict__ scal, int bank) {
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
    unsigned long long* part = scal + ABZ_S_MC_PART;
    part[0] = tg - scal[ABZ_S_MC_TGPREV]; part[1] = ts - scal[ABZ_S_MC_TSPREV];
    part[2] = mn; part[3] = ~mx; part[4] = ~scal[ABZ_S_MC_REJFAIL];
    part[5] = tg; part[6] = ts;
    scal[ABZ_S_MC_TGPREV] = tg; scal[ABZ_S_MC_TSPREV] = ts;
    scal[ABZ_S_MC_REJFAIL] = 0ull;          /* travels in part[4]; the snapshot kernel restores the reduced word */
  }
}
int abz_launch_mc_partial(abcdez_ctx* ctx, int bank) {
  hipLaunchKernelGGL(mc_partial_kernel, dim3(1), dim3(ABZ_CSLOTS), 0, ctx->stream, ctx->d_scal, bank);
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}
int abz_launch_mc_snapshot(abcdez_ctx* ctx, int bank, unsigned long long* d_ring, double alpha,
                           double eps_target, const uint32_t* rank_state, uint32_t N, int sharded) {
  static_assert(ABZ_MMSLOTS <= 64, "mc_snapshot_kernel reduces the bank in wave 0");
  hipLaunchKernelGGL(mc_snapshot_kernel, dim3(1), dim3(ABZ_CSLOTS), 0, ctx->stream, ctx->d_scal, bank, d_ring, alpha, eps_target, rank_state,
                     ctx->mc_rank_limit, N, sharded);
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

/* Start of a chain of asynchronous generations: #(Ds > eps_target) of the population it starts from -> ABZ_S_MC_NABOVE (the
 * first sweep decides from it how to draw its better particles), and the total of the cumulative ABZ_C_MCGT slots as it
 * stands -> ABZ_S_MC_TGPREV (the snapshot kernels take differences against it).  Once per chain.                    */
__global__ __launch_bounds__(ABZ_BLOCK) void mc_chain_count_kernel(const double* __restrict__ delta, uint32_t n, double eps_target,
                                                                   unsigned long long* __restrict__ scal) {
  unsigned int c = 0u;
  for (uint32_t k = blockIdx.x * ABZ_BLOCK + threadIdx.x; k < n; k += gridDim.x * ABZ_BLOCK) c += delta[k] > eps_target ? 1u : 0u;
  for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
  __shared__ unsigned int s_c[ABZ_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < ABZ_BLOCK / 64; ++w) t += s_c[w];
    if (t) atomicAdd(scal + ABZ_S_MC_NABOVE, t);
  }
}
__global__ __launch_bounds__(ABZ_CSLOTS) void mc_chain_total_kernel(unsigned long long* __restrict__ scal) {
  __shared__ unsigned long long s_g[ABZ_CSLOTS / 64], s_s[ABZ_CSLOTS / 64];
  unsigned long long vg = scal[ABZ_S_CSLOT0 + (size_t)threadIdx.x * ABZ_CSTRIDE + ABZ_C_MCGT];
  unsigned long long vs = scal[ABZ_S_CSLOT0 + (size_t)threadIdx.x * ABZ_CSTRIDE + ABZ_C_MCSIM];
  for (int o = 32; o >= 1; o >>= 1) { vg += __shfl_xor(vg, o); vs += __shfl_xor(vs, o); }
  if ((threadIdx.x & 63) == 0) { s_g[threadIdx.x >> 6] = vg; s_s[threadIdx.x >> 6] = vs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tg = 0, ts = 0;
    for (int w = 0; w < ABZ_CSLOTS / 64; ++w) { tg += s_g[w]; ts += s_s[w]; }
    scal[ABZ_S_MC_TGPREV] = tg;
    scal[ABZ_S_MC_TSPREV] = ts;          /* (sharded chains: mc_partial_kernel sends differences against both) */
  }
}
int abz_launch_mc_chain_start(abcdez_ctx* ctx, const double* delta, int64_t N, double eps_target) {
  ABZ_HIP_CHECK(hipMemsetAsync(ctx->d_scal + ABZ_S_MC_NABOVE, 0, 8, ctx->stream));
  unsigned grid = (unsigned)((N + 4 * ABZ_BLOCK - 1) / (4 * ABZ_BLOCK));
  if (grid > 256u) grid = 256u;              /* every block ends with one same-address atomic */
  hipLaunchKernelGGL(mc_chain_count_kernel, dim3(grid), dim3(ABZ_BLOCK), 0, ctx->stream, delta, (uint32_t)N, eps_target, ctx->d_scal);
  hipLaunchKernelGGL(mc_chain_total_kernel, dim3(1), dim3(ABZ_CSLOTS), 0, ctx->stream, ctx->d_scal);
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

abz_rank_plan abz_rank_plan_for(int64_t N, int64_t tail_hint, int64_t tail_bound) {
  const uint32_t n = (uint32_t)N;
  abz_rank_plan p;
  /* both paths are launched (each looks at the tail's length on the device and the wrong one returns at once) unless the host
   * holds a PROVED upper bound of that length (abz_ctx.h, mc_tail_bound): then only the path the bound calls for */
  p.small_path = tail_bound < 0 || tail_bound <= (int64_t)MCR_SMALL;
  p.long_path = tail_bound < 0 || tail_bound > (int64_t)MCR_SMALL;
  /* step 3 is sized for the tail the host expects (4 x the last one seen, at least 16 K pairs; everything when it knows
   * nothing; a proved bound is exact); its kernels stride, so a longer tail is sorted correctly, only slower */
  uint64_t expect = tail_bound >= 0 ? (uint64_t)tail_bound : tail_hint < 0 ? (uint64_t)n : (uint64_t)tail_hint * 4u + 16384u;
  if (expect > n) expect = n;
  const uint32_t ltiles_max = (n + MCR_TILE - 1) / MCR_TILE;                 /* tiles of a tail that is the whole population */
  /* two grid sizes only (every distinct launch shape of a generation is a graph to capture, ~0.1 ms each): a short tail
   * (<= 32 K pairs: 16 workgroups) or one wave-tile per 512 pairs of the whole population, capped at 1024 workgroups -- the
   * kernels stride over their tiles and leave at once when there is nothing for them */
  const uint32_t lt = (uint32_t)((expect + MCR_TILE - 1) / MCR_TILE);
  const uint32_t big = ltiles_max < 1024u * MCR_WAVES ? ltiles_max : 1024u * MCR_WAVES;
  p.ltiles = lt <= 64u ? (64u < big ? 64u : big) : big;
  if (p.ltiles < 1u) p.ltiles = 1u;
  uint32_t rounds = MCR_BATCH;
  while ((uint64_t)rounds * MCR_ROUND * 4096ull < (uint64_t)n) rounds *= 2;
  const uint32_t ntiles = (uint32_t)(((uint64_t)n + (uint64_t)rounds * MCR_ROUND - 1) / ((uint64_t)rounds * MCR_ROUND));
  const size_t pb = abz_align((size_t)n * 4), kb = abz_align((size_t)n * 8);
  const size_t tb = abz_align((size_t)256 * ltiles_max * 4 + 256 * 4), cb = abz_align((size_t)ntiles * 4), sb = abz_align(64);
  p.ws_bytes = kb + 4 * pb + tb + cb + sb;
  return p;
}

/* win: NULL = (eps_pop, dmax_hint) are host values; else the device window mc_window_kernel wrote (the two host values are
 * ignored).  tail_hint: the length of the tail the host last saw (< 0: unknown) -- sizes the launches of the long-tail path. */
int abz_rank_prepare_impl(abcdez_ctx* ctx, const double* delta, int64_t N, double eps_pop, double dmax_hint,
                          uint32_t* order, double* sorted_delta, uint32_t* cnt, const unsigned long long* win, int64_t tail_hint,
                          int64_t tail_bound) {
  const uint32_t n = (uint32_t)N;
  /* window of the binning: (eps_pop, dmax_hint] in key space -> 2^24 - 2 buckets */
  const unsigned long long klo = host_order_key(eps_pop);
  unsigned long long khi = host_order_key(dmax_hint);
  if (!(dmax_hint > eps_pop)) khi = klo + 1ull;          /* also catches a NaN hint */
  const unsigned long long range = khi - klo;
  int bits = 0;
  while (bits < 64 && (range >> bits) != 0ull) ++bits;   /* range < 2^bits */
  const int shift = bits > MCR_BITS ? bits - MCR_BITS : 0;
  /* step 1: one wave per tile of rounds x 64 particles, at most 4096 tiles (one workgroup scans their counts) */
  uint32_t rounds = MCR_BATCH;
  while ((uint64_t)rounds * MCR_ROUND * 4096ull < (uint64_t)n) rounds *= 2;
  const uint32_t ntiles = (uint32_t)(((uint64_t)n + (uint64_t)rounds * MCR_ROUND - 1) / ((uint64_t)rounds * MCR_ROUND));
  const abz_rank_plan plan = abz_rank_plan_for(N, tail_hint, tail_bound);
  const uint32_t ltiles_max = (n + MCR_TILE - 1) / MCR_TILE;                 /* tiles of a tail that is the whole population */
  const uint32_t ltiles = plan.ltiles;
  const size_t pb = abz_align((size_t)n * 4), kb = abz_align((size_t)n * 8);
  const size_t tb = abz_align((size_t)256 * ltiles_max * 4 + 256 * 4), cb = abz_align((size_t)ntiles * 4), sb = abz_align(64);
  int rc = abz_ws_reserve(ctx, kb + 4 * pb + tb + cb + sb);
  if (rc) return rc;
  char* w = (char*)ctx->ws;
  unsigned long long* tk = (unsigned long long*)w; w += kb;
  uint32_t* tv = (uint32_t*)w; w += pb;
  uint32_t* keyA = (uint32_t*)w; w += pb;
  uint32_t* keyB = (uint32_t*)w; w += pb;
  uint32_t* valB = (uint32_t*)w; w += pb;
  uint32_t* table = (uint32_t*)w;
  uint32_t* totals = table + (size_t)256 * ltiles_max; w += tb;
  uint32_t* tile_cnt = (uint32_t*)w; w += cb;
  uint32_t* state = (uint32_t*)w;
  const unsigned grid = (ntiles + MCR_WAVES - 1) / MCR_WAVES;
  const unsigned lgrid = (ltiles + MCR_WAVES - 1) / MCR_WAVES;
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(mcr_count_kernel, dim3(grid), dim3(ABZ_BLOCK), 0, st, delta, n, klo, tile_cnt, ntiles, rounds, win);
  hipLaunchKernelGGL(mcr_split_kernel, dim3(grid), dim3(ABZ_BLOCK), 0, st, delta, n, klo, eps_pop, tile_cnt, ntiles, rounds, order,
                     sorted_delta, tk, tv, win, state);
  /* which sort: abz_rank_plan_for */
  const bool small_path = plan.small_path, long_path = plan.long_path;
  ctx->n_rank_paths[small_path && long_path ? 0 : small_path ? 1 : 2] += 1;
  const uint32_t lim = small_path ? (uint32_t)MCR_SMALL : 0u;      /* tails up to here are the LDS sort's; 0: the radix sort takes any length */
  if (small_path)
    hipLaunchKernelGGL(mcr_tail_small_kernel, dim3(1), dim3(MCR_SMALL_THREADS), 0, st, state, tk, tv, order, sorted_delta, cnt);
  if (long_path) {
    /* pass 0 (tk -> keyA; tv) -> (keyB, valB); pass 1 -> (keyA, tv); pass 2 -> (keyB = buckets, order) */
    hipLaunchKernelGGL((mcr_hist_kernel<0>), dim3(lgrid), dim3(ABZ_BLOCK), 0, st, state, tk, klo, shift, keyA, table, win, lim);
    hipLaunchKernelGGL(mcr_scan_kernel, dim3(256), dim3(ABZ_BLOCK), 0, st, state, table, totals, lim);
    hipLaunchKernelGGL((mcr_scatter_kernel<0, false>), dim3(lgrid), dim3(ABZ_BLOCK), 0, st, state, keyA, tv, table, totals, keyB, valB, delta,
                       order, sorted_delta, cnt, lim);
    hipLaunchKernelGGL((mcr_hist_kernel<1>), dim3(lgrid), dim3(ABZ_BLOCK), 0, st, state, tk, klo, shift, keyB, table, win, lim);
    hipLaunchKernelGGL(mcr_scan_kernel, dim3(256), dim3(ABZ_BLOCK), 0, st, state, table, totals, lim);
    hipLaunchKernelGGL((mcr_scatter_kernel<1, false>), dim3(lgrid), dim3(ABZ_BLOCK), 0, st, state, keyB, valB, table, totals, keyA, tv, delta,
                       order, sorted_delta, cnt, lim);
    hipLaunchKernelGGL((mcr_hist_kernel<2>), dim3(lgrid), dim3(ABZ_BLOCK), 0, st, state, tk, klo, shift, keyA, table, win, lim);
    hipLaunchKernelGGL(mcr_scan_kernel, dim3(256), dim3(ABZ_BLOCK), 0, st, state, table, totals, lim);
    hipLaunchKernelGGL((mcr_scatter_kernel<2, true>), dim3(lgrid), dim3(ABZ_BLOCK), 0, st, state, keyA, tv, table, totals, keyB, valB, delta,
                       order, sorted_delta, cnt, lim);
    hipLaunchKernelGGL(mcr_fixup_kernel, dim3((unsigned)(((uint64_t)ltiles * MCR_TILE + ABZ_BLOCK - 1) / ABZ_BLOCK)), dim3(ABZ_BLOCK), 0, st,
                       state, keyB, order, sorted_delta, cnt, lim);
  }
  ctx->mc_rank_limit = long_path ? 0xFFFFFFFFu : (uint32_t)MCR_SMALL;   /* longest tail the launched path(s) can sort */
  ctx->mc_rank_state = state;       /* the snapshot kernel of an asynchronous generation reports the tail's length from here */
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}
