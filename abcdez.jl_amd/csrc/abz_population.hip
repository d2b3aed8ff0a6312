/*
 * abz_population.hip -- population-wide streaming passes around the sweeps:
 *   tile-tree sums + ABC-kernel reweighting  (src/abcdez_smc.jl:59-83, :308-311, :8)
 *   partition of the packed population (wave ballot / popcount prefix)   (replaces the scans of smc:121,125)
 *   integer cumulative weights + stratified search     (src/abcdez_smc.jl:15-56)
 *   radix select for the eps-quantile                  (src/abcdez_smc.jl:301)
 *   extrema / counts                                   (smc:286,364; mc:133,146,156,163)
 * All HBM-streaming, a few bytes per particle per generation.
 */
#define ABZ_PRIOR_WRAP 1        /* this translation unit's kernels evaluate the wrapper prior families too (include/abcdez_spec.h) */
#include <chrono>
#include <sched.h>
#include <string.h>

#include "abz_ctx.h"
#include "abz_device.h"
#include "abz_dispatch.h"

/* One global atomic per BLOCK (same-address atomics serialise at ~10 ns each; the first build
 * issued one per wave and spent 0.3 ms per pass on them).                                    */
__device__ inline unsigned long long block_sum_u64(unsigned long long v) {
  __shared__ unsigned long long s_bs[ABZ_BLOCK / 64];
  for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) s_bs[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long t = 0;
  if (threadIdx.x == 0) for (int w = 0; w < ABZ_BLOCK / 64; ++w) t += s_bs[w];
  return t;   /* valid on thread 0 */
}
__device__ inline void block_minmax_u64(unsigned long long& lo, unsigned long long& hi) {
  __shared__ unsigned long long s_lo[ABZ_BLOCK / 64], s_hi[ABZ_BLOCK / 64];
  for (int off = 32; off; off >>= 1) {
    const unsigned long long a = __shfl_xor(lo, off, 64), b = __shfl_xor(hi, off, 64);
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0)
    for (int w = 0; w < ABZ_BLOCK / 64; ++w) { lo = s_lo[w] < lo ? s_lo[w] : lo; hi = s_hi[w] > hi ? s_hi[w] : hi; }
}
#define ABZ_REDUCE_GRID 512u

/* ================================================================ tile-tree sum
 * The fixed summation tree of abcdez_spec.h.  One block = one tile of 2048.     */
enum { LOAD_PLAIN = 0, LOAD_SQUARE = 1, LOAD_WPROD = 2, LOAD_NORMALISE = 3 };

struct TileArgs {
  const double* x;        /* PLAIN/SQUARE: input; NORMALISE: wprod                      */
  double* partials;       /* one per tile                                              */
  int64_t n;
  /* reweight */
  const double* delta;
  double* wns;
  uint8_t* alive;
  double* wprod;          /* WPROD: output temp                                        */
  const double* wnorm;    /* NORMALISE: device scalar                                  */
  double* tile_alive;     /* NORMALISE: per-tile alive count (exact in f64)           */
  double eps_old, eps_new;
  const double* eps_new_dev;   /* WPROD: non-null = read eps_new from the device (fused prologue) */
  int abck;
};

template <int MODE>
__device__ inline double tile_elem(const TileArgs& a, int64_t k, int& n_pos) {
  if (k >= a.n) return 0.0;
  if constexpr (MODE == LOAD_PLAIN) {
    return a.x[k];
  } else if constexpr (MODE == LOAD_SQUARE) {
    const double v = a.x[k];
    return v * v;
  } else if constexpr (MODE == LOAD_WPROD) {
    /* ws[i] = exp(logpdf(K_new, D_i) - logpdf(K_old, D_i)) for alive i (smc:77-82); wprod = Wns .* ws (smc:308) */
    double wp = 0.0;
    if (a.alive[k]) {
      const double d = a.delta[k];
      const double en = a.eps_new_dev ? *a.eps_new_dev : a.eps_new;
      const double w = abz_exp(abz_kernel_logpdf(a.abck, en, d) - abz_kernel_logpdf(a.abck, a.eps_old, d));
      wp = a.wns[k] * w;
    }
    a.wprod[k] = wp;
    return wp;
  } else {
    /* Wns = wprod ./ wnorm; alive = Wns .> 0 (smc:310-311); element = Wns^2 for get_ess (smc:8) */
    const double W = a.x[k] / *a.wnorm;
    a.wns[k] = W;
    a.alive[k] = (uint8_t)(W > 0.0);
    n_pos += W > 0.0;
    return W * W;
  }
}

/* the prologue's streaming reads of the distances (as non-temporal loads they change nothing: the first sweep of a generation is not
 * slower for rows these passes displaced, profiles/HISTORY.md round 5) */
__device__ inline double abz_ld_stream(const double* p) { return *p; }

template <int MODE>
__global__ __launch_bounds__(ABZ_BLOCK) void tile_sum_kernel(const TileArgs a) {
  __shared__ double s_w[4];
  const int t = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * ABZ_TILE;
  double e[8];
  int c = 0;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    e[2 * m] = tile_elem<MODE>(a, base + m * 512 + 2 * t, c);
    e[2 * m + 1] = tile_elem<MODE>(a, base + m * 512 + 2 * t + 1, c);
  }
  if constexpr (MODE == LOAD_NORMALISE) {
    /* alive count rides along (sum(alive), smc:352,357) */
    const unsigned long long bc = block_sum_u64((unsigned long long)c);
    if (t == 0) a.tile_alive[blockIdx.x] = (double)bc;
  }
  double s = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) s = s + shfl_xor_f64(s, off);
  if ((t & 63) == 0) s_w[t >> 6] = s;
  __syncthreads();
  if (t == 0) a.partials[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

/* The upper levels of the tile tree in ONE launch (the recursion used to be one launch per level and per sum): block b
 * finishes tree b -- level-1 tiles of its n_b partials one after the other (results in LDS), then the one level-2 tile --
 * with the operations and the order of tile_sum_kernel<LOAD_PLAIN>, so the same bits.  n_b <= 2048 * 2048 partials. */
struct FinishArgs {
  const double* part[2];
  double* out[2];
  int64_t n[2];
  /* pub_host != null: the block that finishes last also publishes the scalar area to the host (what publish_kernel does):
   * one launch and its dependency gap less per prologue */
  unsigned int* ticket;
  const unsigned long long* pub_scal;
  unsigned long long* pub_host;
  int pub_words;
  unsigned long long pub_seq;
};
__device__ inline double finish_tile(const double* __restrict__ x, int64_t n, int64_t base, double* s_w) {
  const int t = threadIdx.x;
  double e[8];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int64_t k = base + m * 512 + 2 * t;
    e[2 * m] = k < n ? x[k] : 0.0;
    e[2 * m + 1] = k + 1 < n ? x[k + 1] : 0.0;
  }
  double s = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) s = s + shfl_xor_f64(s, off);
  __syncthreads();                                     /* s_w is reused from tile to tile */
  if ((t & 63) == 0) s_w[t >> 6] = s;
  __syncthreads();
  return (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);        /* every thread gets the tile's sum */
}
__global__ __launch_bounds__(ABZ_BLOCK) void tree_finish_kernel(const FinishArgs a) {
  __shared__ double s_w[4];
  __shared__ double s_l1[ABZ_TILE];
  __shared__ int s_last;
  const double* x = a.part[blockIdx.x];
  const int64_t n = a.n[blockIdx.x];
  const int64_t nt = (n + ABZ_TILE - 1) / ABZ_TILE;
  double r;
  if (nt == 1) {
    r = finish_tile(x, n, 0, s_w);
  } else {
    for (int64_t tile = 0; tile < nt; ++tile) {
      const double v = finish_tile(x, n, tile * ABZ_TILE, s_w);
      if (threadIdx.x == 0) s_l1[tile] = v;
    }
    __syncthreads();
    r = finish_tile(s_l1, nt, 0, s_w);
  }
  if (threadIdx.x == 0) *a.out[blockIdx.x] = r;
  if (!a.pub_host) return;
  if (threadIdx.x == 0) {
    __threadfence();                                   /* this block's result before its ticket */
    s_last = atomicAdd(a.ticket, 1u) == gridDim.x - 1u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (threadIdx.x == 0) *a.ticket = 0u;                /* for the next launch (same stream: nobody else is looking) */
  for (int k = threadIdx.x; k < a.pub_words; k += ABZ_BLOCK)
    a.pub_host[k] = __hip_atomic_load(a.pub_scal + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   /* past the vector L1 */
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(a.pub_host + ABZ_S_N, a.pub_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* sums `n` elements under MODE: launches the element pass; the result is in *d_out if that was one tile, else `*pending`
 * partials wait in part0 for tree_finish.  Uses ws. */
template <int MODE>
static int tree_sum_begin(abcdez_ctx* ctx, TileArgs a, double* d_out, double* part0, int64_t* pending) {
  const int64_t nt = (a.n + ABZ_TILE - 1) / ABZ_TILE;
  a.partials = nt == 1 ? d_out : part0;
  hipLaunchKernelGGL((tile_sum_kernel<MODE>), dim3((unsigned)nt), dim3(ABZ_BLOCK), 0, ctx->stream, a);
  ABZ_HIP_CHECK(hipGetLastError());
  *pending = nt == 1 ? 0 : nt;
  return 0;
}
/* finishes up to two trees in one launch (n = 0: nothing to do for that tree) */
static int tree_finish(abcdez_ctx* ctx, const double* pa, int64_t na, double* outa, const double* pb = nullptr, int64_t nb = 0,
                       double* outb = nullptr, int pub_words = 0, unsigned long long* pub_seq = nullptr) {
  FinishArgs f{};
  int k = 0;
  if (na > 0) { f.part[k] = pa; f.n[k] = na; f.out[k] = outa; ++k; }
  if (nb > 0) { f.part[k] = pb; f.n[k] = nb; f.out[k] = outb; ++k; }
  if (k == 0) return pub_seq ? abz_publish_launch(ctx, pub_words, pub_seq) : 0;
  if (pub_seq) {
    *pub_seq = ++ctx->pub_seq;
    f.ticket = ctx->d_sync + ABZ_SYNC_TILE; f.pub_scal = ctx->d_scal; f.pub_host = ctx->h_scal_dev; f.pub_words = pub_words;
    f.pub_seq = *pub_seq;
  }
  for (int q = 0; q < k; ++q)
    if (f.n[q] > (int64_t)ABZ_TILE * ABZ_TILE) { abz_set_error("tile tree: too many partials for one finishing block"); return -3; }
  hipLaunchKernelGGL(tree_finish_kernel, dim3((unsigned)k), dim3(ABZ_BLOCK), 0, ctx->stream, f);
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}
template <int MODE>
static int tree_sum_device(abcdez_ctx* ctx, TileArgs a, double* d_out, double* part0, double* /*part1*/) {
  int64_t pending = 0;
  int rc = tree_sum_begin<MODE>(ctx, a, d_out, part0, &pending);
  if (rc) return rc;
  return tree_finish(ctx, part0, pending, d_out);
}

__global__ __launch_bounds__(ABZ_BLOCK) void publish_kernel(const unsigned long long* __restrict__ scal,
                                                           unsigned long long* __restrict__ host, int nwords,
                                                           unsigned long long seq) {
  for (int k = threadIdx.x; k < nwords; k += ABZ_BLOCK) host[k] = scal[k];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host + ABZ_S_N, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int abz_publish_launch(abcdez_ctx* ctx, int nwords, unsigned long long* seq_out) {
  const unsigned long long seq = ++ctx->pub_seq;
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(ABZ_BLOCK), 0, ctx->stream, ctx->d_scal, ctx->h_scal_dev, nwords, seq);
  ABZ_HIP_CHECK(hipGetLastError());
  *seq_out = seq;
  return 0;
}
int abz_poll_word(abcdez_ctx* ctx, const unsigned long long* word, unsigned long long expected) {
  /* The waits of the hot loop are 0.1 - 1.5 ms (a prologue, a group of sweeps) and every microsecond of wake-up latency is a
   * microsecond of idle GPU: those are spun through with `pause` (kind to the core's other hardware thread), exactly as long as
   * the bare spin of round 2 took.  Only a wait beyond 4 ms (a heavy user simulator, a huge population) starts giving the core
   * away, asks the stream now and then, and after 2 s blocks in hipStreamSynchronize like the copy-engine path would. */
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == expected) return 0;
    __builtin_ia32_pause();
    if ((++spins & 1023u) != 0u) continue;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(4)) break;
  }
  for (spins = 0;; ++spins) {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == expected) return 0;
    sched_yield();
    if ((spins & 1023u) != 1023u) continue;
    const hipError_t q = hipStreamQuery(ctx->stream);
    const bool late = std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2);
    if (q == hipSuccess || late) {
      if (late) ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));   /* block, however long it takes */
      return __atomic_load_n(word, __ATOMIC_ACQUIRE) == expected ? 0 : 1;
    }
    if (q != hipErrorNotReady) ABZ_HIP_CHECK(q);
  }
}
/* waits for THAT publish kernel only: work enqueued behind it may still be running when this returns */
int abz_publish_wait(abcdez_ctx* ctx, int nwords, unsigned long long seq) {
  const int rc = abz_poll_word(ctx, ctx->h_scal + ABZ_S_N, seq);
  if (rc < 0) return rc;
  if (rc == 1) {
    /* the stream drained without the word (a failed launch): fall back to the copy engine so that the error surfaces */
    ABZ_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, (size_t)nwords * 8, hipMemcpyDeviceToHost, ctx->stream));
    ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  return 0;
}
int abz_publish(abcdez_ctx* ctx, int nwords) {
  unsigned long long seq = 0;
  if (int rc = abz_publish_launch(ctx, nwords, &seq)) return rc;
  return abz_publish_wait(ctx, nwords, seq);
}
static int read_scalars(abcdez_ctx* ctx) {        /* the plain scalars only; the counter slots are read by abz_api.hip */
  return abz_publish(ctx, ABZ_S_SCALARS);
}
static double scal_f64(abcdez_ctx* ctx, int slot) { double v; memcpy(&v, &ctx->h_scal[slot], 8); return v; }

static int partial_buffers(abcdez_ctx* ctx, int64_t n, size_t extra, double** p0, double** p1, char** rest) {
  const size_t nt = (size_t)((n + ABZ_TILE - 1) / ABZ_TILE);
  const size_t pb = abz_align(nt * 8 + 8);
  int rc = abz_ws_reserve(ctx, 2 * pb + extra);
  if (rc) return rc;
  *p0 = (double*)ctx->ws;
  *p1 = (double*)((char*)ctx->ws + pb);
  if (rest) *rest = (char*)ctx->ws + 2 * pb;
  return 0;
}

int abz_tree_sum_impl(abcdez_ctx* ctx, const double* x, int64_t n, int square, double* out) {
  double *p0, *p1;
  int rc = partial_buffers(ctx, n, 0, &p0, &p1, nullptr);
  if (rc) return rc;
  TileArgs a{};
  a.x = x; a.n = n;
  double* d_out = (double*)(ctx->d_scal + ABZ_S_SUM);
  rc = square ? tree_sum_device<LOAD_SQUARE>(ctx, a, d_out, p0, p1) : tree_sum_device<LOAD_PLAIN>(ctx, a, d_out, p0, p1);
  if (rc) return rc;
  rc = read_scalars(ctx);
  if (rc) return rc;
  *out = scal_f64(ctx, ABZ_S_SUM);
  return 0;
}

#define ABZ_CHUNK 1024      /* positions per chunk of the partition's count / list passes */

/* ---- the reweight of an INDICATOR kernel on UNIFORM weights (abcdez_ctx_set_uniform_weights; the oracle's
 * orc_smc_reweight_uniform): ws[i] is 1 or 0 (types.jl:26-50) and Wns is 1 / n_alive on the alive particles, so
 *     wnorm = n_new / n_old,   Wns = 1 / n_new on the survivors,   1 / sum(Wns.^2) = n_new
 * (smc:308-311, :8) -- IEEE divisions of exactly represented integers instead of floating sums of n_new equal terms.  ONE pass
 * writes the new flags and counts the survivors per tile (10 B per position instead of 26 + 17 in two passes); the finishing
 * launch leaves n_new, wnorm and 1 / n_new in the scalar area and publishes; the weights themselves are a fill that rides in
 * the partition's list pass (part_list_kernel).  Two launches instead of four. */
__global__ __launch_bounds__(ABZ_BLOCK) void ind_reweight_kernel(const double* __restrict__ delta, uint8_t* __restrict__ alive, int64_t n,
                                                                 int abck, const double* __restrict__ eps_dev, double eps_host,
                                                                 uint32_t* __restrict__ acnt) {
  const double eps = eps_dev ? *eps_dev : eps_host;
  const int t = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * ABZ_TILE;
  unsigned long long c[2] = {0ull, 0ull};                   /* survivors of the tile's two 1024-chunks (the partition's chunks) */
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int64_t k = base + m * 512 + 2 * t;               /* two consecutive positions per thread: 16-byte distance loads */
    double d0 = 0.0, d1 = 0.0;
    uint8_t a0 = 0, a1 = 0;
    if (k + 1 < n) {
      const double2 dd = *reinterpret_cast<const double2*>(delta + k);
      const uchar2 aa = *reinterpret_cast<const uchar2*>(alive + k);
      d0 = dd.x; d1 = dd.y; a0 = aa.x; a1 = aa.y;
    } else if (k < n) { d0 = delta[k]; a0 = alive[k]; }
    const bool s0 = a0 && abz_kernel_insupport(abck, eps, d0), s1 = a1 && abz_kernel_insupport(abck, eps, d1);
    if (k + 1 < n) *reinterpret_cast<uchar2*>(alive + k) = make_uchar2((unsigned char)s0, (unsigned char)s1);
    else if (k < n) alive[k] = (uint8_t)s0;
    c[m >> 1] += (unsigned long long)s0 + (unsigned long long)s1;
  }
  static_assert(ABZ_TILE == 2048 && ABZ_BLOCK == 256, "a tile is two chunks of 1024 positions");
  const unsigned long long b0 = block_sum_u64(c[0]);
  __syncthreads();                                          /* block_sum_u64 reuses its LDS words */
  const unsigned long long b1 = block_sum_u64(c[1]);
  if (t == 0) {
    const int64_t nchunk = (n + 1023) / 1024;
    acnt[2 * blockIdx.x] = (uint32_t)b0;
    if (2 * (int64_t)blockIdx.x + 1 < nchunk) acnt[2 * blockIdx.x + 1] = (uint32_t)b1;
  }
}

/* finish + scan of the indicator fast path in ONE launch of two fat blocks (what tree_finish + part_count + part_scan do in three):
 * every block adds up the chunk counts -> n_new; block 0 leaves n_new, wnorm = n_new / n_old, 1 / n_new in the scalar area and
 * publishes the scalars to the host; block 0 then makes the exclusive offsets of the HOLES per chunk (dead positions below
 * n_new), block 1 those of the FILLERS (alive positions at or above n_new) -- from the chunk counts alone, except for the one
 * chunk n_new falls into, whose flags are read.  Nothing is listed when a resampling is ahead (ESS < ess_min, smc:323). */
__global__ __launch_bounds__(1024) void ind_finish_scan_kernel(const uint32_t* __restrict__ acnt, uint32_t nchunk, uint32_t n_prev,
                                                               double ess_min, const uint8_t* __restrict__ alive,
                                                               uint32_t* __restrict__ cnt, unsigned long long* __restrict__ scal,
                                                               unsigned long long* __restrict__ pub_host, int pub_words,
                                                               unsigned long long pub_seq) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_below;
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const uint32_t per = (nchunk + 1023u) / 1024u;
  const uint32_t lo = t * per < nchunk ? t * per : nchunk, hi = lo + per < nchunk ? lo + per : nchunk;
  uint32_t s = 0;
  for (uint32_t k = lo; k < hi; ++k) s += acnt[k];
  for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off, 64);          /* wave sums by shuffles, the 16 of them through LDS: */
  if (lane == 0) s_wave[wave] = s;                                          /* two barriers where a tree over 1024 words takes ten */
  if (t == 0) s_below = 0u;
  __syncthreads();
  uint32_t n_new = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) n_new += s_wave[w];
  __syncthreads();
  const double nn = (double)n_new, sumsq = 1.0 / nn, ess = 1.0 / sumsq;
  const bool go = !(nn > 0.0 && ess < ess_min);
  if (blockIdx.x == 0 && t == 0) {
    scal[ABZ_S_NALIVE] = abz_d2u(nn);
    scal[ABZ_S_WNORM] = abz_d2u(nn / (double)n_prev);
    scal[ABZ_S_SUMSQ] = abz_d2u(sumsq);
    scal[ABZ_S_PART_ERR] = 0ull;
  }
  if (blockIdx.x == 0 && pub_host) {                   /* the host needs nothing of what follows */
    __threadfence();
    __syncthreads();
    for (int k = (int)t; k < pub_words; k += 1024)
      pub_host[k] = __hip_atomic_load(scal + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(pub_host + ABZ_S_N, pub_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  /* the chunk n_new falls into: alive positions in [ks 1024, n_new) */
  const uint32_t ks = n_new / ABZ_CHUNK;
  if (go && ks < nchunk && (n_new % ABZ_CHUNK) != 0u) {
    const uint32_t k = ks * ABZ_CHUNK + t;
    const bool a = k < n_new && alive[k];
    const unsigned long long b = __ballot(a);
    if ((t & 63u) == 0u && b) atomicAdd(&s_below, (uint32_t)__popcll(b));
  }
  __syncthreads();
  const uint32_t below = s_below;
  /* this block's per-chunk value: holes (block 0) or fillers (block 1) */
  auto value = [&](uint32_t k) -> uint32_t {
    if (!go) return 0u;
    const uint32_t a = acnt[k];
    const uint32_t len = (k + 1u) * ABZ_CHUNK <= n_prev ? (uint32_t)ABZ_CHUNK : n_prev - k * ABZ_CHUNK;
    if (blockIdx.x == 0) return k < ks ? len - a : k == ks ? (n_new - ks * ABZ_CHUNK) - below : 0u;
    return k < ks ? 0u : k == ks ? a - below : a;
  };
  uint32_t v = 0;
  for (uint32_t k = lo; k < hi; ++k) v += value(k);
  uint32_t inc = v;                                                         /* inclusive scan over the block: in the wave by shuffles ... */
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(inc, off, 64);
    if (lane >= (uint32_t)off) inc += o;
  }
  if (lane == 63u) s_wave[wave] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;                                           /* ... and over the 16 wave totals by everybody */
#pragma unroll
  for (int w = 0; w < 16; ++w) { const uint32_t x = s_wave[w]; before += (uint32_t)w < wave ? x : 0u; total += x; }
  uint32_t run = before + inc - v;
  uint32_t* out = cnt + (size_t)blockIdx.x * nchunk;
  for (uint32_t k = lo; k < hi; ++k) { const uint32_t c = value(k); out[k] = run; run += c; }
  if (t == 1023) scal[ABZ_S_PART_H + blockIdx.x] = (unsigned long long)total;
}

/* pub_seq != null: the launch that finishes the sums also publishes the first ABZ_S_SCALARS scalars to the host */
static int reweight_enqueue(abcdez_ctx* ctx, const double* delta, double* wns, uint8_t* alive, int64_t N, double eps_old,
                            double eps_new, const double* eps_new_dev, unsigned long long* pub_seq = nullptr) {
  double *p0, *p1;
  char* rest;
  const size_t ntile = (size_t)((N + ABZ_TILE - 1) / ABZ_TILE);
  int rc = partial_buffers(ctx, N, abz_align((size_t)N * 8) + abz_align(ntile * 8), &p0, &p1, &rest);
  if (rc) return rc;
  double* wprod = (double*)rest;
  double* tile_alive = (double*)(rest + abz_align((size_t)N * 8));
  TileArgs a{};
  a.n = N; a.delta = delta; a.wns = wns; a.alive = alive; a.wprod = wprod;
  a.eps_old = eps_old; a.eps_new = eps_new; a.eps_new_dev = eps_new_dev; a.abck = ctx->h_model.abck;
  rc = tree_sum_device<LOAD_WPROD>(ctx, a, (double*)(ctx->d_scal + ABZ_S_WNORM), p0, p1);
  if (rc) return rc;
  TileArgs b{};
  b.n = N; b.x = wprod; b.wns = wns; b.alive = alive;
  b.wnorm = (const double*)(ctx->d_scal + ABZ_S_WNORM);
  b.tile_alive = tile_alive;
  int64_t pending = 0;
  rc = tree_sum_begin<LOAD_NORMALISE>(ctx, b, (double*)(ctx->d_scal + ABZ_S_SUMSQ), p0, &pending);
  if (rc) return rc;
  /* sum(Wns^2) and sum(alive) (tile counts: integers < 2^53, the f64 tree sum is exact) finish in one launch */
  return tree_finish(ctx, p0, pending, (double*)(ctx->d_scal + ABZ_S_SUMSQ), tile_alive, (int64_t)ntile,
                     (double*)(ctx->d_scal + ABZ_S_NALIVE), ABZ_S_SCALARS, pub_seq);
}
int abz_reweight_impl(abcdez_ctx* ctx, const double* delta, double* wns, uint8_t* alive, int64_t N, double eps_old,
                      double eps_new, double* wnorm, double* ess, int64_t* n_alive) {
  int rc = reweight_enqueue(ctx, delta, wns, alive, N, eps_old, eps_new, nullptr);
  if (rc) return rc;
  rc = read_scalars(ctx);
  if (rc) return rc;
  *wnorm = scal_f64(ctx, ABZ_S_WNORM);
  *ess = 1.0 / scal_f64(ctx, ABZ_S_SUMSQ);
  *n_alive = (int64_t)scal_f64(ctx, ABZ_S_NALIVE);
  return 0;
}


/* ================================================================ partition of the packed population
 * After a reweight the alive flags of the prefix [0, n_prev) have holes; n_new = sum(alive) is known to the host.
 * The k-th dead position below n_new ("hole") swaps its state with the k-th alive position at or above n_new
 * ("filler"): afterwards the alive particles are the positions [0, n_new).  About 5 % of the prefix moves per
 * generation at alpha = 0.95 (each hole costs two row reads and two row writes).
 *   part_count   per 1024-chunk: #holes, #fillers                     (wave ballots)
 *   part_scan    exclusive scans of both count arrays (two blocks), totals -> scalar area
 *   part_list    the two position lists in ascending order (ballot prefix inside the chunk) + bit-array sync
 *   part_swap    lane group k swaps rows / log-prior / distance / weight / flag / stamp of (hole_k, filler_k)     */
/* dyn (fused prologue): n_new and the resampling decision are still on the device -- n_new = scal[NALIVE] (an f64), and
 * when the ESS 1 / scal[SUMSQ] is below ess_min the driver will resample (smc:323-326), so nothing is partitioned */
__device__ inline uint32_t part_n_new(const unsigned long long* __restrict__ dyn, uint32_t n_new, uint32_t n_prev, double ess_min,
                                      bool* go) {
  *go = true;
  if (!dyn) return n_new;
  const double na = abz_u2d(dyn[ABZ_S_NALIVE]);
  const double ess = 1.0 / abz_u2d(dyn[ABZ_S_SUMSQ]);
  if (na > 0.0 && ess < ess_min) *go = false;          /* resampling ahead: neither holes nor fillers are listed */
  return (uint32_t)na;
}
__global__ __launch_bounds__(ABZ_BLOCK) void part_count_kernel(const uint8_t* __restrict__ alive, uint32_t n_prev,
                                                               uint32_t n_new_h, uint32_t* __restrict__ cnt, uint32_t nchunk,
                                                               const unsigned long long* __restrict__ dyn, double ess_min,
                                                               unsigned long long* __restrict__ err) {
  __shared__ uint32_t s_c[2];
  bool go;
  const uint32_t n_new = part_n_new(dyn, n_new_h, n_prev, ess_min, &go);
  if (blockIdx.x == 0 && threadIdx.x == 0) *err = 0ull;
  if (threadIdx.x < 2) s_c[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * ABZ_CHUNK;
  uint32_t h = 0, f = 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const uint32_t k = base + it * ABZ_BLOCK + threadIdx.x;
    const bool in = go && k < n_prev;
    const bool al = in && alive[k];
    h += (uint32_t)__popcll(__ballot(in && !al && k < n_new));
    f += (uint32_t)__popcll(__ballot(al && k >= n_new));
  }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&s_c[0], h); atomicAdd(&s_c[1], f); }
  __syncthreads();
  if (threadIdx.x < 2) cnt[threadIdx.x * nchunk + blockIdx.x] = s_c[threadIdx.x];
}

__global__ __launch_bounds__(1024) void part_scan_kernel(uint32_t* __restrict__ cnt, uint32_t nchunk,
                                                         unsigned long long* __restrict__ totals) {
  __shared__ uint32_t s_part[1024];
  uint32_t* v = cnt + (size_t)blockIdx.x * nchunk;
  const uint32_t t = threadIdx.x;
  const uint32_t per = (nchunk + 1023) / 1024;
  const uint32_t lo = t * per < nchunk ? t * per : nchunk, hi = lo + per < nchunk ? lo + per : nchunk;
  uint32_t s = 0;
  for (uint32_t k = lo; k < hi; ++k) s += v[k];
  s_part[t] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t add = t >= off ? s_part[t - off] : 0;
    __syncthreads();
    s_part[t] += add;
    __syncthreads();
  }
  uint32_t run = t ? s_part[t - 1] : 0;
  for (uint32_t k = lo; k < hi; ++k) { const uint32_t c = v[k]; v[k] = run; run += c; }
  if (t == 1023) totals[blockIdx.x] = s_part[1023];
}

__global__ __launch_bounds__(ABZ_BLOCK) void part_list_kernel(const uint8_t* __restrict__ alive, uint32_t n_prev,
                                                              uint32_t n_new_h, const uint32_t* __restrict__ off,
                                                              uint32_t nchunk, uint32_t* __restrict__ holes,
                                                              uint32_t* __restrict__ fillers,
                                                              const uint32_t* __restrict__ bits,
                                                              uint32_t* __restrict__ bits_other, uint32_t nwords,
                                                              const unsigned long long* __restrict__ dyn, double ess_min,
                                                              double* __restrict__ wfill) {
  __shared__ uint32_t s_wave[2][4];
  bool go;
  const uint32_t n_new = part_n_new(dyn, n_new_h, n_prev, ess_min, &go);
  /* wfill != null (indicator reweight on uniform weights, ind_reweight_kernel): the new weights are a fill -- 1 / n_new on the
   * survivors, 0 on the others -- written here, where the flags of the whole prefix are read anyway (also when a resampling is
   * ahead and nothing is listed: the resampling reads them) */
  const double winv = wfill ? abz_u2d(dyn[ABZ_S_SUMSQ]) : 0.0;
  /* both bit arrays must agree wherever no sweep writes: positions that just left the prefix keep the bit of the
   * CURRENT array (their last sweep may have flipped it) */
  for (uint32_t w = blockIdx.x * ABZ_BLOCK + threadIdx.x; w < nwords; w += gridDim.x * ABZ_BLOCK) bits_other[w] = bits[w];
  const uint32_t base = blockIdx.x * ABZ_CHUNK;
  uint32_t run_h = off[blockIdx.x], run_f = off[nchunk + blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int it = 0; it < 4; ++it) {
    const uint32_t k = base + it * ABZ_BLOCK + threadIdx.x;
    const bool inp = k < n_prev;
    const bool alv = inp && alive[k];
    if (wfill && inp) wfill[k] = alv ? winv : 0.0;
    const bool in = go && inp;
    const bool al = in && alv;
    const bool is_h = in && !al && k < n_new, is_f = al && k >= n_new;
    const unsigned long long bh = __ballot(is_h), bf = __ballot(is_f);
    if (lane == 0) { s_wave[0][wave] = (uint32_t)__popcll(bh); s_wave[1][wave] = (uint32_t)__popcll(bf); }
    __syncthreads();
    uint32_t wh = 0, wf = 0;
    for (int w = 0; w < wave; ++w) { wh += s_wave[0][w]; wf += s_wave[1][w]; }
    if (is_h) holes[run_h + wh + (uint32_t)__popcll(bh & below)] = k;
    if (is_f) fillers[run_f + wf + (uint32_t)__popcll(bf & below)] = k;
    run_h += s_wave[0][0] + s_wave[0][1] + s_wave[0][2] + s_wave[0][3];
    run_f += s_wave[1][0] + s_wave[1][1] + s_wave[1][2] + s_wave[1][3];
    __syncthreads();
  }
}

template <int L, int C>
__global__ __launch_bounds__(ABZ_BLOCK) void part_swap_kernel(const uint32_t* __restrict__ holes,
                                                              const uint32_t* __restrict__ fillers,
                                                              const unsigned long long* __restrict__ totals,
                                                              const uint32_t* __restrict__ bits, double* __restrict__ slot0,
                                                              double* __restrict__ slot1, double* __restrict__ logpi,
                                                              double* __restrict__ delta, double* __restrict__ wns,
                                                              uint8_t* __restrict__ alive, uint64_t* __restrict__ stamp,
                                                              unsigned long long* __restrict__ err) {
  constexpr int LD = L * C;
  const uint32_t gid = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  const int j = (int)(gid % L);
  const uint32_t m = (uint32_t)totals[0];
  if (gid == 0 && totals[0] != totals[1]) *err = 1ull;          /* the flags do not describe a prefix of length n_prev */
  if (totals[0] != totals[1]) return;
  for (uint32_t k = gid / L; k < m; k += gridDim.x * (ABZ_BLOCK / L)) {     /* the grid does not depend on m */
    const uint32_t h = holes[k], f = fillers[k];
    double* rh = ((bits[h >> 5] >> (h & 31u)) & 1u ? slot1 : slot0) + (size_t)h * LD;
    double* rf = ((bits[f >> 5] >> (f & 31u)) & 1u ? slot1 : slot0) + (size_t)f * LD;
    double a[C], b[C];
    load_row<L, C>(rh, j, a);
    load_row<L, C>(rf, j, b);
    store_row<L, C>(rh, j, b);          /* (non-temporal stores here: no difference, profiles/HISTORY.md round 5) */
    store_row<L, C>(rf, j, a);
    if (j == 0) {
      double t;
      t = logpi[h]; logpi[h] = logpi[f]; logpi[f] = t;
      t = delta[h]; delta[h] = delta[f]; delta[f] = t;
      t = wns[h]; wns[h] = wns[f]; wns[f] = t;
      alive[h] = 1; alive[f] = 0;
      if (stamp) { const uint64_t u = stamp[h]; stamp[h] = stamp[f]; stamp[f] = u; }
    }
  }
}

/* dyn != null: n_new and the "no resampling ahead" predicate are read from the device (fused prologue; n_new is ignored) */
int abz_partition_impl(abcdez_ctx* ctx, uint8_t* alive, int64_t N, int64_t n_prev, int64_t n_new, const uint32_t* bits,
                       uint32_t* bits_other, double* slot0, double* slot1, double* logpi, double* delta, double* wns,
                       const unsigned long long* dyn, double ess_min, bool wfill) {
  const uint32_t np = (uint32_t)n_prev, nn = (uint32_t)n_new;
  const uint32_t nchunk = (np + ABZ_CHUNK - 1) / ABZ_CHUNK;
  const uint32_t bound = dyn ? np / 2 : (nn < np - nn ? nn : np - nn);             /* #swaps <= min(#dead, #alive) */
  const size_t cb = abz_align((size_t)2 * nchunk * 4), lb = abz_align((size_t)(bound + 1) * 4);
  /* (every stage of the fused prologue is done with the workspace before the next stage's kernels start: one stream) */
  int rc = abz_ws_reserve(ctx, cb + 2 * lb);
  if (rc) return rc;
  char* base = (char*)ctx->ws;
  uint32_t* cnt = (uint32_t*)base;
  uint32_t* holes = (uint32_t*)(base + cb);
  uint32_t* fillers = (uint32_t*)(base + cb + lb);
  unsigned long long* totals = ctx->d_scal + ABZ_S_PART_H;
  const uint32_t nwords = (uint32_t)((N + 31) / 32);
  if (nchunk == 0) return 0;
  hipLaunchKernelGGL(part_count_kernel, dim3(nchunk), dim3(ABZ_BLOCK), 0, ctx->stream, alive, np, nn, cnt, nchunk, dyn, ess_min,
                     ctx->d_scal + ABZ_S_PART_ERR);
  hipLaunchKernelGGL(part_scan_kernel, dim3(2), dim3(1024), 0, ctx->stream, cnt, nchunk, totals);
  hipLaunchKernelGGL(part_list_kernel, dim3(nchunk), dim3(ABZ_BLOCK), 0, ctx->stream, alive, np, nn, cnt, nchunk, holes, fillers,
                     bits, bits_other, nwords, dyn, ess_min, wfill ? wns : nullptr);
  if (bound > 0) {
    bool ok = abz_dispatch_lc(ctx->L, ctx->C, [&](auto LL, auto CC) {
      uint64_t blocks = ((uint64_t)bound * LL() + ABZ_BLOCK - 1) / ABZ_BLOCK;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL((part_swap_kernel<LL(), CC()>), dim3((unsigned)blocks), dim3(ABZ_BLOCK), 0, ctx->stream, holes, fillers,
                         totals, bits, slot0, slot1, logpi, delta, wns, alive, ctx->stamp_cur, ctx->d_scal + ABZ_S_PART_ERR);
    });
    if (!ok) { abz_set_error("smc_partition: unsupported layout"); return -3; }
  }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

/* The reweight + partition of the prologue for an indicator kernel on uniform weights: four launches (count pass, finish + scan,
 * list + weight fill, swap) where the general path has eight. */
int abz_ind_reweight_partition(abcdez_ctx* ctx, const double* delta_all, uint8_t* alive, int64_t N, int64_t n_prev, double ess_min,
                               const uint32_t* bits, uint32_t* bits_other, double* slot0, double* slot1, double* logpi,
                               double* delta_rw, double* wns, unsigned long long* pub_seq) {
  const uint32_t np = (uint32_t)n_prev;
  const uint32_t nchunk = (np + ABZ_CHUNK - 1) / ABZ_CHUNK;
  const uint32_t ntile = (np + ABZ_TILE - 1) / ABZ_TILE;
  const uint32_t bound = np / 2;                                                   /* #swaps <= min(#dead, #alive) */
  const size_t cb = abz_align((size_t)2 * nchunk * 4), lb = abz_align((size_t)(bound + 1) * 4), ab = abz_align((size_t)(nchunk + 1) * 4);
  int rc = abz_ws_reserve(ctx, cb + 2 * lb + ab);
  if (rc) return rc;
  char* base = (char*)ctx->ws;
  uint32_t* cnt = (uint32_t*)base;
  uint32_t* holes = (uint32_t*)(base + cb);
  uint32_t* fillers = (uint32_t*)(base + cb + lb);
  uint32_t* acnt = (uint32_t*)(base + cb + 2 * lb);
  unsigned long long* totals = ctx->d_scal + ABZ_S_PART_H;
  const uint32_t nwords = (uint32_t)((N + 31) / 32);
  hipLaunchKernelGGL(ind_reweight_kernel, dim3(ntile), dim3(ABZ_BLOCK), 0, ctx->stream, delta_all, alive, n_prev, ctx->h_model.abck,
                     (const double*)(ctx->d_scal + ABZ_S_EPS), 0.0, acnt);
  *pub_seq = ++ctx->pub_seq;
  hipLaunchKernelGGL(ind_finish_scan_kernel, dim3(2), dim3(1024), 0, ctx->stream, (const uint32_t*)acnt, nchunk, np, ess_min,
                     (const uint8_t*)alive, cnt, ctx->d_scal, ctx->h_scal_dev, (int)ABZ_S_SCALARS, *pub_seq);
  hipLaunchKernelGGL(part_list_kernel, dim3(nchunk), dim3(ABZ_BLOCK), 0, ctx->stream, alive, np, 0u, cnt, nchunk, holes, fillers,
                     bits, bits_other, nwords, (const unsigned long long*)ctx->d_scal, ess_min, wns);
  if (bound > 0) {
    bool ok = abz_dispatch_lc(ctx->L, ctx->C, [&](auto LL, auto CC) {
      uint64_t blocks = ((uint64_t)bound * LL() + ABZ_BLOCK - 1) / ABZ_BLOCK;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL((part_swap_kernel<LL(), CC()>), dim3((unsigned)blocks), dim3(ABZ_BLOCK), 0, ctx->stream, holes, fillers,
                         totals, bits, slot0, slot1, logpi, delta_rw, wns, alive, ctx->stamp_cur, ctx->d_scal + ABZ_S_PART_ERR);
    });
    if (!ok) { abz_set_error("smc_partition: unsupported layout"); return -3; }
  }
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

/* ================================================================ stratified resampling (smc:15-56)
 * integer weights (abz_weight_fix) -> inclusive scan in u64 -> per-stratum search. */
#define ABZ_SCAN_TILE 2048 /* 256 threads x 8 consecutive elements */

__device__ inline void wfix_load8(const double* __restrict__ wns, uint32_t base, uint32_t N, double (&w)[8]) {
  if (base + 8u <= N) {                  /* base is a multiple of 8: 64-byte aligned */
    const double4 a = *reinterpret_cast<const double4*>(wns + base), b = *reinterpret_cast<const double4*>(wns + base + 4);
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = base + (uint32_t)q < N ? wns[base + q] : 0.0;
  }
}
__global__ __launch_bounds__(ABZ_BLOCK) void wfix_tile_sum_kernel(const double* __restrict__ wns, uint32_t N,
                                                                  unsigned long long* __restrict__ tile_sum,
                                                                  uint32_t* __restrict__ tile_lp) {
  __shared__ unsigned long long s_w[4];
  const uint32_t base = blockIdx.x * ABZ_SCAN_TILE + threadIdx.x * 8;
  unsigned long long s = 0;
  uint32_t lp1 = 0;                      /* 1 + last index with positive weight, 0 = none */
  double w8[8];
  wfix_load8(wns, base, N, w8);          /* a thread's 8 consecutive weights as two 32-byte loads (eight 8-byte loads at a 64-byte lane stride cost the address unit eight requests per line) */
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const uint32_t k = base + q;
    if (k < N) {
      const unsigned long long f = abz_weight_fix(w8[q], N);
      s += f;
      if (f) lp1 = k + 1;
    }
  }
  for (int off = 32; off; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    const uint32_t o = __shfl_xor(lp1, off, 64);
    lp1 = o > lp1 ? o : lp1;
  }
  __shared__ uint32_t s_lp[4];
  if ((threadIdx.x & 63) == 0) {
    s_w[threadIdx.x >> 6] = s;
    s_lp[threadIdx.x >> 6] = lp1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    tile_sum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    uint32_t m = s_lp[0];
    for (int w = 1; w < 4; ++w) m = s_lp[w] > m ? s_lp[w] : m;
    tile_lp[blockIdx.x] = m;             /* folded by scan_u64_kernel (2048 atomicMax on one word cost more than this whole pass) */
  }
}

/* exclusive scan of the tile sums (one block); tile_lp != NULL: also 1 + the last position with a positive weight = the
 * maximum of the tiles' values -> *last_pos */
__global__ __launch_bounds__(1024) void scan_u64_kernel(unsigned long long* __restrict__ v, uint32_t n,
                                                        const uint32_t* __restrict__ tile_lp, unsigned long long* __restrict__ last_pos) {
  __shared__ unsigned long long s_part[1024];
  __shared__ uint32_t s_mx[16];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t lo = t * per, hi = lo + per < n ? lo + per : n;
  unsigned long long s = 0;
  uint32_t mx = 0;
  for (uint32_t k = lo; k < hi; ++k) { s += v[k]; if (tile_lp) { const uint32_t m = tile_lp[k]; mx = m > mx ? m : mx; } }
  if (tile_lp) {
    for (int off = 32; off; off >>= 1) { const uint32_t o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; }
    if ((t & 63u) == 0u) s_mx[t >> 6] = mx;
  }
  s_part[t] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    unsigned long long add = t >= off ? s_part[t - off] : 0;
    __syncthreads();
    s_part[t] += add;
    __syncthreads();
  }
  unsigned long long run = t ? s_part[t - 1] : 0;
  for (uint32_t k = lo; k < hi; ++k) { const unsigned long long c = v[k]; v[k] = run; run += c; }
  if (tile_lp && t == 0) {               /* (s_mx was written before the scan's barriers) */
    uint32_t m = 0;
    for (int w = 0; w < 16; ++w) m = s_mx[w] > m ? s_mx[w] : m;
    *last_pos = (unsigned long long)m;
  }
}

__global__ __launch_bounds__(ABZ_BLOCK) void wfix_scan_kernel(const double* __restrict__ wns, uint32_t N,
                                                              const unsigned long long* __restrict__ tile_off,
                                                              unsigned long long* __restrict__ cum) {
  __shared__ unsigned long long s_w[4];
  const uint32_t base = blockIdx.x * ABZ_SCAN_TILE + threadIdx.x * 8;
  unsigned long long f[8];
  unsigned long long s = 0;
  double w8[8];
  wfix_load8(wns, base, N, w8);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const uint32_t k = base + q;
    f[q] = k < N ? abz_weight_fix(w8[q], N) : 0ull;
    s += f[q];
  }
  /* inclusive scan of s over the wave */
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long inc = s;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  unsigned long long run = tile_off[blockIdx.x] + (inc - s);
  for (int w = 0; w < wave; ++w) run += s_w[w];
  unsigned long long c8[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { run += f[q]; c8[q] = run; }
  if (base + 8u <= N) {                  /* two 32-byte stores */
    *reinterpret_cast<ulonglong4*>(cum + base) = make_ulonglong4(c8[0], c8[1], c8[2], c8[3]);
    *reinterpret_cast<ulonglong4*>(cum + base + 4) = make_ulonglong4(c8[4], c8[5], c8[6], c8[7]);
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) if (base + (uint32_t)q < N) cum[base + q] = c8[q];
  }
}

__global__ __launch_bounds__(ABZ_BLOCK) void stratified_search_kernel(const unsigned long long* __restrict__ cum,
                                                                      uint32_t N, uint64_t seed, uint32_t draw,
                                                                      const unsigned long long* __restrict__ last_pos,
                                                                      uint32_t* __restrict__ inds) {
  const uint32_t s = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  if (s >= N) return;
  const uint64_t R = abz_stratum_point(seed, N, s, draw);
  uint32_t lo = 0, hi = N;               /* smallest i with cum[i] > R */
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (cum[mid] > R) hi = mid; else lo = mid + 1;
  }
  const uint32_t lp1 = (uint32_t)*last_pos;
  const uint32_t lp = lp1 ? lp1 - 1 : 0u;
  inds[s] = (lo >= N || lo > lp) ? lp : lo;
}

int abz_stratified_impl(abcdez_ctx* ctx, const double* wns, int64_t N, uint32_t draw, uint32_t* inds) {
  const uint32_t n = (uint32_t)N;
  const uint32_t nt = (n + ABZ_SCAN_TILE - 1) / ABZ_SCAN_TILE;
  const size_t tb = abz_align((size_t)nt * 8);
  const size_t lpb = abz_align((size_t)nt * 4);
  int rc = abz_ws_reserve(ctx, tb + abz_align((size_t)n * 8) + lpb);
  if (rc) return rc;
  unsigned long long* tile = (unsigned long long*)ctx->ws;
  unsigned long long* cum = (unsigned long long*)((char*)ctx->ws + tb);
  uint32_t* tile_lp = (uint32_t*)((char*)ctx->ws + tb + abz_align((size_t)n * 8));
  hipLaunchKernelGGL(wfix_tile_sum_kernel, dim3(nt), dim3(ABZ_BLOCK), 0, ctx->stream, wns, n, tile, tile_lp);
  hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, ctx->stream, tile, nt, (const uint32_t*)tile_lp,
                     ctx->d_scal + ABZ_S_LASTPOS);
  hipLaunchKernelGGL(wfix_scan_kernel, dim3(nt), dim3(ABZ_BLOCK), 0, ctx->stream, wns, n, tile, cum);
  hipLaunchKernelGGL(stratified_search_kernel, dim3((n + ABZ_BLOCK - 1) / ABZ_BLOCK), dim3(ABZ_BLOCK), 0, ctx->stream,
                     cum, n, ctx->h_model.seed, draw, ctx->d_scal + ABZ_S_LASTPOS, inds);
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}

/* ================================================================ order statistics of alive distances (smc:301)
 * Exact selection of the rank-k alive distance and of the next larger one, on the order-preserving
 * u64 image of the IEEE bit patterns (f64_order_key: negative values and -0.0 sort below +0.0; NaN never
 * enters the population; +Inf is just the largest key).
 *
 *   pass 1  qs_hist_kernel     2048-bin histogram of  (key - klo) >> shift  (clamped: any
 *                              monotone binning is correct, a good one is fast) + min/max key
 *   pass 2  qs_compact_kernel  every block scans the histogram for the bin holding rank k,
 *                              keys of that bin -> buffer (LDS-staged, one atomic per flush),
 *                              smallest key of any higher bin -> `above`
 *   finish  qs_final_kernel    ONE block narrows the buffer by 11 key bits per round until
 *                              <= 1024 candidates remain, ranks those in LDS; leaves the
 *                              device state ready for the next call (histogram zeroed,
 *                              accumulators reset) and the next call's binning window.
 *
 * The window [klo, khi] lives on the device: the finish kernel sets it to [smallest alive key
 * seen, key of rank k+1] -- in the driver's loop the next call's alive distances all lie below
 * this call's quantile (smc:301-311), so the 2048 bins resolve exactly the range that is left.
 * A stale window only costs time (clamped bins grow, the finish kernel works longer).  When the
 * arrays change (first call, after a resampling's buffer swap) a min/max pass seeds the window.
 * Two population passes, 3 launches, no memset / upload; replaces a 6-digit MSB radix select
 * (3 passes, 14 launches).                                                                      */
#define ABZ_QS_BINS 2048
#define ABZ_QS_FAT 1024             /* passes 1 / 2: one fat block per CU -> few histogram flushes, few append atomics */
#define ABZ_QS_GRID 256
#define ABZ_QS_CAP 12288            /* keys a block of pass 2 stages in LDS (96 KB; one fat block per CU) */
#define ABZ_QS_LDSKEYS 4096
#define ABZ_QS_FINAL_THREADS 1024    /* threads of the finishing block of the select */

__device__ inline int qs_shift(unsigned long long klo, unsigned long long khi) {
  if (khi <= klo) return 0;
  const int bl = 64 - __clzll((long long)(khi - klo));      /* khi - klo < 2^bl */
  return bl > 11 ? bl - 11 : 0;                             /* (khi - klo) >> shift < 2048 */
}
__device__ inline uint32_t qs_bin(unsigned long long key, unsigned long long klo, int shift) {
  if (key <= klo) return 0u;
  const unsigned long long b = (key - klo) >> shift;
  return b < ABZ_QS_BINS ? (uint32_t)b : ABZ_QS_BINS - 1u;
}

/* 2048 sub-bins of the selected bin `sel` (again clamped and monotone): 11 more key bits */
__device__ inline uint32_t qs_sub(unsigned long long key, unsigned long long base, int s2) {
  if (key <= base) return 0u;
  const unsigned long long b = (key - base) >> s2;
  return b < ABZ_QS_BINS ? (uint32_t)b : ABZ_QS_BINS - 1u;
}

#define QS(slot) st[(slot) - ABZ_S_SEL_PREFIX]

/* window seed: min / max alive key -> st[HLO], st[HHI] (preset to ~0 / 0 by the host) */
__global__ __launch_bounds__(ABZ_BLOCK) void qs_minmax_kernel(const double* __restrict__ delta,
                                                              const uint8_t* __restrict__ alive, int64_t N,
                                                              unsigned long long* __restrict__ st) {
  unsigned long long lo = ~0ull, hi = 0ull;
  const int64_t stride = (int64_t)gridDim.x * ABZ_BLOCK;
  for (int64_t k = (int64_t)blockIdx.x * ABZ_BLOCK + threadIdx.x; k < N; k += stride) {
    const unsigned long long key = f64_order_key(abz_ld_stream(delta + k));
    if (!alive || alive[k]) { lo = key < lo ? key : lo; hi = key > hi ? key : hi; }
  }
  block_minmax_u64(lo, hi);
  if (threadIdx.x == 0 && lo <= hi) { atomicMin(&QS(ABZ_S_SEL_HLO), lo); atomicMax(&QS(ABZ_S_SEL_HHI), hi); }
}

/* min over a fat block; valid on thread 0 */
__device__ inline unsigned long long qs_fat_min(unsigned long long v, unsigned long long* s_w) {
  for (int off = 32; off; off >>= 1) { const unsigned long long a = __shfl_xor(v, off, 64); v = a < v ? a : v; }
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) for (int w = 1; w < ABZ_QS_FAT / 64; ++w) v = s_w[w] < v ? s_w[w] : v;
  return v;
}

/* the contiguous range of n positions a block of passes 1 / 2 streams: n / grid rounded up to whole rounds of the block */
__device__ inline int64_t qs_block_share(int64_t n, unsigned grid) {
  const int64_t per = (n + (int64_t)grid - 1) / (int64_t)grid;
  return (per + ABZ_QS_FAT - 1) / ABZ_QS_FAT * ABZ_QS_FAT;
}
/* pass 1: histogram of the alive keys in the window's bins; smallest alive key per block -> bmin[block] */
/* n_all > 0 (fused prologue): extrema(Ds) over ALL n_all positions, alive or not (smc:364), ride along -- the positions past the
 * alive prefix are read for this only; per-block results -> ball[block], ball[grid + block], folded by qs_final_kernel */
__global__ __launch_bounds__(ABZ_QS_FAT) void qs_hist_kernel(const double* __restrict__ delta,
                                                             const uint8_t* __restrict__ alive, int64_t N,
                                                             const unsigned long long* __restrict__ st,
                                                             uint32_t* __restrict__ hist,
                                                             unsigned long long* __restrict__ bmin, int64_t n_all,
                                                             unsigned long long* __restrict__ ball) {
  __shared__ uint32_t s_h[ABZ_QS_BINS];
  __shared__ unsigned long long s_w[ABZ_QS_FAT / 64];
  unsigned long long alo = ~0ull, ahi = 0ull;
  for (int b = threadIdx.x; b < ABZ_QS_BINS; b += ABZ_QS_FAT) s_h[b] = 0;
  const unsigned long long klo = QS(ABZ_S_SEL_HLO);
  const int shift = qs_shift(klo, QS(ABZ_S_SEL_HHI));
  __syncthreads();
  unsigned long long lo = ~0ull;
  /* One fat block per CU: nothing else runs on the CU while this block waits for memory, so every trip puts all it can in
   * flight at once -- 12 (distance, flag) pairs of the alive prefix AND 8 distances of the dead tail (read for the extrema
   * only) per lane; at 2^22 particles that is the whole kernel in one round trip. */
  constexpr int UM = 12, UT = 8;
  /* block b streams ONE contiguous range of the prefix and one of the tail (consecutive 8 KB pieces: blocks striding through
   * the whole array touched a different 2 MB page with every load -- 3 us of 16, tools/qs_ablate.hip); alive == NULL: every
   * position of the prefix is alive (the packed population's invariant: the flag bytes are not read, 1.3 us) */
  const int64_t per = qs_block_share(N, gridDim.x), tper = qs_block_share(n_all > N ? n_all - N : 0, gridDim.x);
  int64_t k0 = (int64_t)blockIdx.x * per + threadIdx.x;                        /* cursor in the prefix */
  int64_t t0 = N + (int64_t)blockIdx.x * tper + threadIdx.x;                   /* cursor in the dead tail */
  const int64_t kend = ((int64_t)blockIdx.x + 1) * per < N ? ((int64_t)blockIdx.x + 1) * per : N;
  const int64_t tend = N + ((int64_t)blockIdx.x + 1) * tper < n_all ? N + ((int64_t)blockIdx.x + 1) * tper : n_all;
  constexpr int64_t stride = ABZ_QS_FAT;
  while (k0 < kend || t0 < tend) {                                             /* block-uniform */
    unsigned long long key[UM], tkey[UT];
    uint8_t al[UM];
#pragma unroll
    for (int u = 0; u < UM; ++u) {
      const int64_t k = k0 + u * stride;
      const bool in = k < kend;
      key[u] = in ? f64_order_key(abz_ld_stream(delta + k)) : 0ull;
      al[u] = in ? (alive ? alive[k] : (uint8_t)1) : (uint8_t)0;
    }
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int64_t k = t0 + u * stride;
      tkey[u] = k < tend ? f64_order_key(abz_ld_stream(delta + k)) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < UM; ++u) {
      if (n_all > 0 && k0 + u * stride < kend) { alo = key[u] < alo ? key[u] : alo; ahi = key[u] > ahi ? key[u] : ahi; }
      if (al[u]) {
        atomicAdd(&s_h[qs_bin(key[u], klo, shift)], 1u);
        lo = key[u] < lo ? key[u] : lo;
      }
    }
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      if (t0 + u * stride < tend) { alo = tkey[u] < alo ? tkey[u] : alo; ahi = tkey[u] > ahi ? tkey[u] : ahi; }
    }
    k0 += UM * stride;
    t0 += UT * stride;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < ABZ_QS_BINS; b += ABZ_QS_FAT)
    if (s_h[b]) atomicAdd(&hist[b], s_h[b]);
  lo = qs_fat_min(lo, s_w);
  if (threadIdx.x == 0) bmin[blockIdx.x] = lo;
  if (n_all > 0) {
    __syncthreads();
    alo = qs_fat_min(alo, s_w);
    __syncthreads();
    ahi = ~qs_fat_min(~ahi, s_w);
    if (threadIdx.x == 0) { ball[blockIdx.x] = alo; ball[gridDim.x + blockIdx.x] = ahi; }
  }
}

/* pass 2.  Every block finds the bin of rank k0 itself (2048 counters from L2: cheaper than a launch); keys of that
 * bin are staged in LDS and appended with one returning atomic per flush (same-address returning atomics cost ~30 ns
 * each, serialised: hence one fat block per CU); smallest key of a higher bin -> babove[block].                     */
__global__ __launch_bounds__(ABZ_QS_FAT) void qs_compact_kernel(const double* __restrict__ delta,
                                                                const uint8_t* __restrict__ alive, int64_t N,
                                                                unsigned long long k0, const uint32_t* __restrict__ hist,
                                                                unsigned long long* __restrict__ st,
                                                                unsigned long long* __restrict__ buf,
                                                                unsigned long long* __restrict__ babove,
                                                                uint32_t* __restrict__ hist2) {
  /* babove[0 .. grid): smallest key above the bin; [grid .. 2 grid): smallest, [2 grid .. 3 grid): ~largest key IN the
   * bin (stored complemented, so that one min-reduction serves all three) */
  __shared__ unsigned long long s_buf[ABZ_QS_CAP];
  __shared__ uint32_t s_h2[ABZ_QS_BINS];
  __shared__ unsigned long long s_w[ABZ_QS_FAT / 64];
  __shared__ uint32_t s_n, s_bin;
  __shared__ unsigned long long s_base;
  const int t = threadIdx.x;
  /* one fat block per CU: the first trip's loads go out before anything else -- they need neither the histogram nor the bin --
   * and every later trip's loads go out before the trip in hand is processed (nothing else runs on this CU while it waits) */
  constexpr int64_t stride = ABZ_QS_FAT;                      /* a contiguous range per block, as in pass 1 */
  const int64_t per = qs_block_share(N, gridDim.x);
  const int64_t kend = ((int64_t)blockIdx.x + 1) * per < N ? ((int64_t)blockIdx.x + 1) * per : N;
  unsigned long long key[8];
  uint8_t al[8];
  int64_t base = (int64_t)blockIdx.x * per;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int64_t k = base + t + u * stride;
    const bool in = k < kend;
    key[u] = in ? f64_order_key(abz_ld_stream(delta + k)) : 0ull;
    al[u] = in ? (alive ? alive[k] : (uint8_t)1) : (uint8_t)0;
  }
  const uint32_t h0 = hist[2 * t], h1 = hist[2 * t + 1];      /* 2 consecutive bins per thread */
  const unsigned long long klo = QS(ABZ_S_SEL_HLO), khi_w = QS(ABZ_S_SEL_HHI);   /* requested now, used after the bin is known */
  const unsigned long long mine = (unsigned long long)h0 + h1;
  unsigned long long incl = mine;                      /* inclusive scan over the block */
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long v = __shfl_up(incl, off, 64);
    if ((t & 63) >= off) incl += v;
  }
  if ((t & 63) == 63) s_w[t >> 6] = incl;
  if (t == 0) { s_n = 0; s_bin = 0xFFFFFFFFu; }
  s_h2[2 * t] = 0; s_h2[2 * t + 1] = 0;
  __syncthreads();
  unsigned long long before = incl - mine;
  for (int w = 0; w < (t >> 6); ++w) before += s_w[w];
  if (k0 >= before && k0 < before + mine) {            /* exactly one thread, if k0 < total */
    const bool first = k0 < before + h0;
    s_bin = (uint32_t)(2 * t + (first ? 0 : 1));
    if (blockIdx.x == 0) {
      QS(ABZ_S_SEL_BIN) = (unsigned long long)(2 * t + (first ? 0 : 1));
      QS(ABZ_S_SEL_K) = k0 - (first ? before : before + h0);
      QS(ABZ_S_SEL_LESS) = first ? before : before + h0;
      QS(ABZ_S_SEL_PAD) = 0;
    }
  }
  if (blockIdx.x == 0 && t == ABZ_QS_FAT - 1 && k0 >= before + mine) QS(ABZ_S_SEL_PAD) = 1;   /* rank beyond the population */
  __syncthreads();
  const uint32_t sel = s_bin;
  if (sel == 0xFFFFFFFFu) {
    if (t == 0) { babove[blockIdx.x] = ~0ull; babove[gridDim.x + blockIdx.x] = ~0ull; babove[2 * gridDim.x + blockIdx.x] = ~0ull; }
    return;
  }
  const int shift = qs_shift(klo, khi_w);
  const unsigned long long base2 = klo + ((unsigned long long)sel << shift);
  const int s2 = shift > 11 ? shift - 11 : 0;
  unsigned long long above = ~0ull, inmin = ~0ull, inmaxc = ~0ull;
  auto flush = [&]() {
    if (t == 0) s_base = atomicAdd(&QS(ABZ_S_SEL_NBUF), (unsigned long long)s_n);
    __syncthreads();
    const uint32_t n = s_n;
    for (uint32_t q = t; q < n; q += ABZ_QS_FAT) buf[s_base + q] = s_buf[q];
    __syncthreads();
    if (t == 0) s_n = 0;
    __syncthreads();
  };
  while (base < kend) {                                                                      /* block-uniform trips */
    const int64_t nbase = base + 8 * stride;
    unsigned long long nkey[8];
    uint8_t nal[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {                                                           /* the next trip's loads */
      const int64_t k = nbase + t + u * stride;
      const bool in = k < kend;
      nkey[u] = in ? f64_order_key(abz_ld_stream(delta + k)) : 0ull;
      nal[u] = in ? (alive ? alive[k] : (uint8_t)1) : (uint8_t)0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (!al[u]) continue;
      const uint32_t b = qs_bin(key[u], klo, shift);
      if (b > sel && key[u] < above) above = key[u];
      if (b == sel) {
        s_buf[atomicAdd(&s_n, 1u)] = key[u];                                /* room for 8 x 1024 guaranteed */
        atomicAdd(&s_h2[qs_sub(key[u], base2, s2)], 1u);
        inmin = key[u] < inmin ? key[u] : inmin;
        inmaxc = ~key[u] < inmaxc ? ~key[u] : inmaxc;
      }
    }
    __syncthreads();
    if (s_n > ABZ_QS_CAP - 8 * ABZ_QS_FAT) flush();                         /* s_n is block-uniform here */
#pragma unroll
    for (int u = 0; u < 8; ++u) { key[u] = nkey[u]; al[u] = nal[u]; }
    base = nbase;
  }
  if (s_n) flush();
  if (s_h2[2 * t]) atomicAdd(&hist2[2 * t], s_h2[2 * t]);
  if (s_h2[2 * t + 1]) atomicAdd(&hist2[2 * t + 1], s_h2[2 * t + 1]);
  above = qs_fat_min(above, s_w);
  if (t == 0) babove[blockIdx.x] = above;
  __syncthreads();
  inmin = qs_fat_min(inmin, s_w);
  if (t == 0) babove[gridDim.x + blockIdx.x] = inmin;
  __syncthreads();
  inmaxc = qs_fat_min(inmaxc, s_w);
  if (t == 0) babove[2 * gridDim.x + blockIdx.x] = inmaxc;
}

/* one block of NT threads: sum / min / max over the block, result broadcast to every thread */
template <int NT>
__device__ inline void qs_block_reduce(unsigned long long& cnt, unsigned long long& mn, unsigned long long& mx,
                                       unsigned long long* s_red) {
  for (int off = 32; off; off >>= 1) {
    cnt += __shfl_xor(cnt, off, 64);
    const unsigned long long a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();                                    /* s_red may still be read from the previous round */
  if ((threadIdx.x & 63) == 0) { s_red[3 * w] = cnt; s_red[3 * w + 1] = mn; s_red[3 * w + 2] = mx; }
  __syncthreads();
  cnt = 0; mn = ~0ull; mx = 0ull;
  for (int q = 0; q < NT / 64; ++q) {
    cnt += s_red[3 * q];
    mn = s_red[3 * q + 1] < mn ? s_red[3 * q + 1] : mn;
    mx = s_red[3 * q + 2] > mx ? s_red[3 * q + 2] : mx;
  }
}

/* finish: rank k of the buffered bin, starting from the sub-histogram pass 2 took of it.  Rounds over the candidates (in LDS when they fit): count / min / max; all
 * equal -> done; <= 64 left -> rank them against each other; else 2048 sub-bins of [min, max], keep the one holding
 * the rank.  Every round strips 11 bits off the key range.                                                          */
/* fused prologue: what used to be two more launches rides at the end of the finish kernel */
struct QsTail {
  int eps_on;                 /* smc:301 on the device: q = x_(j) + g (x_(j+1) - x_(j)); eps = max(min(q, eps_prev), eps_target) */
  int single;
  unsigned long long k0;
  double g, eps_prev, eps_target;
  const unsigned long long* ball;   /* per-block extrema of ALL distances from qs_hist_kernel, or NULL */
};
/* NT threads.  What this kernel costs is its ONE pass over the buffered keys of the selected bin -- 5 K to 20 K keys at 2^22
 * particles (the 0.95-quantile sits where the distances are dense), read by a single block: 8 loads in flight per thread, the
 * first batch requested before anything else.  (Timestamps inside the kernel, round 4: that pass was 4-14 us of its 10-20; 256
 * threads instead of 1024 changed nothing -- the barriers are not what it waits for.) */
template <int NT>
__global__ __launch_bounds__(NT) void qs_final_kernel(const unsigned long long* __restrict__ buf,
                                                        unsigned long long* __restrict__ st, uint32_t* __restrict__ hist,
                                                        const unsigned long long* __restrict__ bmin,
                                                        const unsigned long long* __restrict__ babove, int nblk,
                                                        uint32_t* __restrict__ hist2, const QsTail tail) {
  __shared__ uint32_t s_h[ABZ_QS_BINS];
  __shared__ unsigned long long s_keys[ABZ_QS_LDSKEYS];
  __shared__ unsigned long long s_cand[64];
  __shared__ unsigned long long s_red[48];
  __shared__ unsigned long long s_pick[4];            /* bin | key, count before it, (rank branch) hit flag, #equal */
  __shared__ uint32_t s_n;
  const int t = threadIdx.x;
  constexpr int BPT = ABZ_QS_BINS / NT;                                   /* consecutive bins per thread */
  for (int b = t; b < ABZ_QS_BINS; b += NT) hist[b] = 0;                 /* ready for the next call */
  uint32_t gb[BPT];
#pragma unroll
  for (int q = 0; q < BPT; ++q) { gb[q] = hist2[BPT * t + q]; hist2[BPT * t + q] = 0; }
  const bool bad = QS(ABZ_S_SEL_PAD) != 0;
  const int64_t n = bad ? 0 : (int64_t)QS(ABZ_S_SEL_NBUF);
  unsigned long long k = QS(ABZ_S_SEL_K), less = QS(ABZ_S_SEL_LESS);
  /* a single block: every memory round trip is exposed, so the first batch of buffered keys is requested now, before the
   * reductions and the sub-bin scan that say which of them matter */
  constexpr int UB = 8;                                     /* buffered keys in flight per thread */
  unsigned long long x0[UB];
#pragma unroll
  for (int u = 0; u < UB; ++u) x0[u] = buf[t + u * NT];     /* unconditionally: not behind the load of n (the buffer holds at least UB NT words; words past n are never looked at) */
  const unsigned long long w_klo = QS(ABZ_S_SEL_HLO), w_khi = QS(ABZ_S_SEL_HHI), w_bin = QS(ABZ_S_SEL_BIN);   /* the window and the bin, for later */
  unsigned long long ball_lo = ~0ull, ball_hi = 0ull;                     /* per-block extrema of all distances (pass 1), folded at the end */
  if (tail.ball)
    for (int b = t; b < nblk; b += NT) {
      const unsigned long long a = tail.ball[b], c = tail.ball[nblk + b];
      ball_lo = a < ball_lo ? a : ball_lo;
      ball_hi = c > ball_hi ? c : ball_hi;
    }
  unsigned long long above = ~0ull, kmin = ~0ull;
  unsigned long long binmin = ~0ull, binmax = 0ull;       /* smallest / largest key of the selected bin */
  for (int b = t; b < nblk; b += NT) {
    const unsigned long long a = babove[b], m = bmin[b], i0 = babove[nblk + b], i1 = babove[2 * nblk + b];
    above = a < above ? a : above;
    kmin = m < kmin ? m : kmin;
    binmin = i0 < binmin ? i0 : binmin;
    binmax = ~i1 > binmax ? ~i1 : binmax;
  }
  unsigned long long lo = 0ull, hi = ~0ull, key = 0ull, eq = 0ull;
  if (t == 0) s_n = 0;
  bool found = false;
  {
    unsigned long long dcnt = 0;
    qs_block_reduce<NT>(dcnt, binmin, binmax, s_red);
  }
  /* the whole bin is one key value (tied / discrete distances): nothing to search, the buffer is not even read */
  const bool one_value = !bad && n > 0 && binmin == binmax;
  if (one_value) { key = binmin; eq = (unsigned long long)n; found = true; }
  if (!bad && !one_value) {                            /* sub-bin of the rank: 11 key bits without touching a key */
    unsigned long long mine = 0;
#pragma unroll
    for (int q = 0; q < BPT; ++q) mine += gb[q];
    unsigned long long incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long v = __shfl_up(incl, off, 64);
      if ((t & 63) >= off) incl += v;
    }
    if ((t & 63) == 63) s_red[t >> 6] = incl;
    __syncthreads();
    unsigned long long before = incl - mine;
    for (int w = 0; w < (t >> 6); ++w) before += s_red[w];
    if (k >= before && k < before + mine) {                      /* exactly one thread: which of its bins holds rank k */
      unsigned long long run = before;
#pragma unroll
      for (int q = 0; q < BPT; ++q) {
        if (k >= run && k < run + gb[q]) { s_pick[0] = (unsigned long long)(BPT * t + q); s_pick[1] = run; }
        run += gb[q];
      }
    }
    __syncthreads();
    const unsigned long long b2 = s_pick[0];
    less += s_pick[1];
    k -= s_pick[1];
    const unsigned long long klo = w_klo;
    const int shift = qs_shift(klo, w_khi);
    const unsigned long long base2 = klo + (w_bin << shift);
    const int s2 = shift > 11 ? shift - 11 : 0;
    lo = b2 == 0 ? 0ull : base2 + (b2 << s2);
    hi = b2 == ABZ_QS_BINS - 1 ? ~0ull : base2 + ((b2 + 1) << s2) - 1ull;
  }
  __syncthreads();
  /* the one pass over the buffer in global memory: the sub-bin's keys -> LDS (normally a handful), smallest key
   * beyond the sub-bin -> `above`.  Should they not fit, the rounds below read the whole buffer instead.        */
  unsigned long long cmin = ~0ull, cmax = 0ull;
  for (int64_t i0 = t; i0 < (one_value ? 0 : n); i0 += UB * NT) {
    unsigned long long x[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) x[u] = i0 == (int64_t)t ? x0[u] : (i0 + u * NT < n ? buf[i0 + u * NT] : 0ull);
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (i0 + u * NT >= n) continue;
      if (x[u] > hi) above = x[u] < above ? x[u] : above;
      if (x[u] >= lo && x[u] <= hi) {
        const uint32_t q = atomicAdd(&s_n, 1u);
        if (q < ABZ_QS_LDSKEYS) s_keys[q] = x[u];
        cmin = x[u] < cmin ? x[u] : cmin;
        cmax = x[u] > cmax ? x[u] : cmax;
      }
    }
  }
  {
    unsigned long long dcnt = 0;
    qs_block_reduce<NT>(dcnt, cmin, cmax, s_red);      /* (its barriers also publish s_n and s_keys) */
  }
  /* the sub-bin is one key value (an atom of a discrete distance among others): done, however many copies there are */
  const bool one_cand = !bad && !one_value && s_n > 0 && cmin == cmax;
  if (one_cand) { key = cmin; eq = (unsigned long long)s_n; found = true; }
  const bool in_lds = s_n <= ABZ_QS_LDSKEYS;
  const unsigned long long* keys = in_lds ? s_keys : buf;
  const int64_t nk = in_lds ? (int64_t)s_n : n;
  __syncthreads();
  for (int round = 0; round < 16 && !bad && !one_value && !one_cand; ++round) {
    unsigned long long cnt = 0, mn = ~0ull, mx = 0ull;
    for (int64_t i = t; i < nk; i += NT) {
      const unsigned long long x = keys[i];
      if (x >= lo && x <= hi) { ++cnt; mn = x < mn ? x : mn; mx = x > mx ? x : mx; }
    }
    qs_block_reduce<NT>(cnt, mn, mx, s_red);
    if (cnt == 0) break;                               /* cannot happen for k < count; reported as an error below */
    if (mn == mx) { key = mn; eq = cnt; found = true; break; }
    if (cnt <= 64) {
      if (t == 0) { s_n = 0; s_pick[2] = 0; }
      __syncthreads();
      for (int64_t i = t; i < nk; i += NT) {
        const unsigned long long x = keys[i];
        if (x >= lo && x <= hi) s_cand[atomicAdd(&s_n, 1u)] = x;
      }
      __syncthreads();
      if (t < (int)cnt) {
        const unsigned long long me = s_cand[t];
        unsigned long long l = 0, e = 0;
        for (int j = 0; j < (int)cnt; ++j) { const unsigned long long c = s_cand[j]; l += c < me; e += c == me; }
        if (l <= k && k < l + e) { s_pick[0] = me; s_pick[1] = l; s_pick[3] = e; s_pick[2] = 1; }   /* equal keys write equal values */
      }
      __syncthreads();
      if (s_pick[2]) { key = s_pick[0]; less += s_pick[1]; eq = s_pick[3]; found = true; }
      break;
    }
    /* narrow: 2048 sub-bins of [mn, mx] */
    const int bl = 64 - __clzll((long long)(mx - mn));
    const int s = bl > 11 ? bl - 11 : 0;
    for (int b = t; b < ABZ_QS_BINS; b += NT) s_h[b] = 0;
    __syncthreads();
    for (int64_t i = t; i < nk; i += NT) {
      const unsigned long long x = keys[i];
      if (x >= lo && x <= hi) atomicAdd(&s_h[(uint32_t)((x - mn) >> s)], 1u);
    }
    __syncthreads();
    uint32_t hb[BPT];
    unsigned long long mine = 0;
#pragma unroll
    for (int q = 0; q < BPT; ++q) { hb[q] = s_h[BPT * t + q]; mine += hb[q]; }
    unsigned long long incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long v = __shfl_up(incl, off, 64);
      if ((t & 63) >= off) incl += v;
    }
    if ((t & 63) == 63) s_red[t >> 6] = incl;
    __syncthreads();
    unsigned long long before = incl - mine;
    for (int w = 0; w < (t >> 6); ++w) before += s_red[w];
    if (k >= before && k < before + mine) {
      unsigned long long run = before;
#pragma unroll
      for (int q = 0; q < BPT; ++q) {
        if (k >= run && k < run + hb[q]) { s_pick[0] = (unsigned long long)(BPT * t + q); s_pick[1] = run; }
        run += hb[q];
      }
    }
    __syncthreads();
    const unsigned long long b = s_pick[0];
    less += s_pick[1];
    k -= s_pick[1];
    lo = mn + (b << s);
    hi = lo + ((1ull << s) - 1ull);
    __syncthreads();
  }
  /* smallest key strictly greater than the selected one: in the buffer, else the smallest key of a higher bin */
  unsigned long long nxt = above, dc = 0;
  if (found && !one_value && !one_cand)
    for (int64_t i = t; i < nk; i += NT) {
      const unsigned long long x = keys[i];
      if (x > key && x < nxt) nxt = x;
    }
  unsigned long long dm = 0, d0 = 0, d1 = 0;
  qs_block_reduce<NT>(dc, nxt, dm, s_red);
  qs_block_reduce<NT>(d0, kmin, d1, s_red);
  if (t == 0) {
    QS(ABZ_S_SEL_PREFIX) = key;
    QS(ABZ_S_SEL_LESS) = less;
    QS(ABZ_S_SEL_EQ) = eq;
    QS(ABZ_S_SEL_NEXT) = nxt;
    if (!found) QS(ABZ_S_SEL_PAD) = bad ? 1 : 2;
    if (found) {       /* window of the next call: [smallest alive key seen, key of the next rank] */
      QS(ABZ_S_SEL_HLO) = kmin <= key ? kmin : key;
      QS(ABZ_S_SEL_HHI) = nxt != ~0ull ? nxt : key;
    }
    QS(ABZ_S_SEL_NBUF) = 0;
    if (tail.eps_on) {
      unsigned long long* scal = st - ABZ_S_SEL_PREFIX;
      const double a = f64_from_order_key_dev(key);
      const double b = (tail.single || tail.k0 + 1ull < less + eq || nxt == ~0ull) ? a : f64_from_order_key_dev(nxt);
      const double q = a + tail.g * (b - a);
      double e = q < tail.eps_prev ? q : tail.eps_prev;   /* min(q, eps): Julia's min propagates NaN; distances are never NaN */
      e = e > tail.eps_target ? e : tail.eps_target;
      scal[ABZ_S_QVAL] = abz_d2u(q);
      scal[ABZ_S_EPS] = abz_d2u(e);
    }
  }
  if (tail.ball) {
    unsigned long long c0 = 0, lo = ball_lo, hi = ball_hi;
    qs_block_reduce<NT>(c0, lo, hi, s_red);
    if (t == 0) { unsigned long long* scal = st - ABZ_S_SEL_PREFIX; scal[ABZ_S_MIN] = lo; scal[ABZ_S_MAX] = hi; }
  }
}
#undef QS

static inline double f64_from_order_key(unsigned long long k) {
  return (k >> 63) ? abz_u2d(k & 0x7FFFFFFFFFFFFFFFull) : abz_u2d(~k);
}
__device__ inline double dev_from_order_key(unsigned long long k) {
  return (k >> 63) ? abz_u2d(k & 0x7FFFFFFFFFFFFFFFull) : abz_u2d(~k);
}

/* enqueue the three passes of the select for rank k0 (0-based) among the alive distances; results stay on the device */
/* all_alive: every one of the N positions is alive (the prefix of the packed population): the kernels do not read the flags */
static int select_enqueue(abcdez_ctx* ctx, const double* delta, const uint8_t* alive, int64_t N, int64_t k0,
                          const QsTail* tail_in = nullptr, int64_t n_all = 0, bool all_alive = false) {
  const size_t buf_bytes = (size_t)N * 8 > (size_t)8 * ABZ_QS_FINAL_THREADS * 8 ? (size_t)N * 8 : (size_t)8 * ABZ_QS_FINAL_THREADS * 8;
  int rc = abz_ws_reserve(ctx, abz_align(buf_bytes));         /* at least the 8 NT words the finishing block requests up front */
  if (rc) return rc;
  unsigned long long* buf = (unsigned long long*)ctx->ws;
  unsigned long long* st = ctx->d_scal + ABZ_S_SEL_PREFIX;
  if (!ctx->sel_hist) {        /* histogram + sub-histogram (left zeroed by every call) + per-block minima of the two passes */
    ABZ_HIP_CHECK(hipMalloc((void**)&ctx->sel_hist, 2 * ABZ_QS_BINS * 4 + 6 * ABZ_QS_GRID * 8));
    ctx->sel_clean = false;
  }
  uint32_t* hist2 = ctx->sel_hist + ABZ_QS_BINS;
  unsigned long long* bmin = (unsigned long long*)(ctx->sel_hist + 2 * ABZ_QS_BINS);
  unsigned long long* babove = bmin + ABZ_QS_GRID;
  /* a SHORTER prefix of the same arrays keeps the window (packed population: the alive prefix shrinks every generation) */
  const bool reseed = !ctx->sel_clean || ctx->sel_delta != delta || ctx->sel_alive != alive || ctx->sel_N < N;
  if (reseed) {
    /* device state of the select from scratch + window from a min / max pass */
    unsigned long long init[ABZ_S_SEL_END - ABZ_S_SEL_PREFIX] = {0};
    init[ABZ_S_SEL_HLO - ABZ_S_SEL_PREFIX] = ~0ull;
    ABZ_HIP_CHECK(hipMemcpyAsync(st, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    ABZ_HIP_CHECK(hipMemsetAsync(ctx->sel_hist, 0, 2 * ABZ_QS_BINS * 4, ctx->stream));
    unsigned mgrid = (unsigned)((N + ABZ_BLOCK - 1) / ABZ_BLOCK);
    if (mgrid > ABZ_REDUCE_GRID) mgrid = ABZ_REDUCE_GRID;
    hipLaunchKernelGGL(qs_minmax_kernel, dim3(mgrid), dim3(ABZ_BLOCK), 0, ctx->stream, delta, all_alive ? nullptr : alive, N, st);
  }
  ctx->sel_clean = false;
  unsigned grid = (unsigned)((N + 8 * ABZ_QS_FAT - 1) / (8 * ABZ_QS_FAT));
  if (grid > ABZ_QS_GRID) grid = ABZ_QS_GRID;
  unsigned long long* ball = babove + 3 * ABZ_QS_GRID;      /* after the three arrays of qs_compact_kernel */
  QsTail tail{};
  if (tail_in) tail = *tail_in;
  tail.ball = n_all > 0 ? ball : nullptr;
  const uint8_t* flags = all_alive ? nullptr : alive;
  hipLaunchKernelGGL(qs_hist_kernel, dim3(grid), dim3(ABZ_QS_FAT), 0, ctx->stream, delta, flags, N, st, ctx->sel_hist, bmin,
                     n_all, ball);
  hipLaunchKernelGGL(qs_compact_kernel, dim3(grid), dim3(ABZ_QS_FAT), 0, ctx->stream, delta, flags, N,
                     (unsigned long long)k0, ctx->sel_hist, st, buf, babove, hist2);
  hipLaunchKernelGGL((qs_final_kernel<ABZ_QS_FINAL_THREADS>), dim3(1), dim3(ABZ_QS_FINAL_THREADS), 0, ctx->stream, buf, st, ctx->sel_hist, bmin,
                     babove, (int)grid, hist2, tail);
  ABZ_HIP_CHECK(hipGetLastError());
  ctx->sel_delta = delta; ctx->sel_alive = alive; ctx->sel_N = N;
  return 0;
}
/* after a read-back of the scalars: validate and convert */
static int select_finish(abcdez_ctx* ctx, int64_t k0, double* xk, double* xk1, int64_t* n_le) {
  ctx->sel_clean = true;
  const unsigned long long key = ctx->h_scal[ABZ_S_SEL_PREFIX];
  const unsigned long long less = ctx->h_scal[ABZ_S_SEL_LESS], eq = ctx->h_scal[ABZ_S_SEL_EQ];
  const unsigned long long next = ctx->h_scal[ABZ_S_SEL_NEXT];
  if (ctx->h_scal[ABZ_S_SEL_PAD]) {
    ctx->sel_clean = false;
    abz_set_error(ctx->h_scal[ABZ_S_SEL_PAD] == 1
                      ? "quantile_alive: requested rank is beyond the number of alive particles (wrong n_alive_hint?)"
                      : "quantile_alive: internal selection error");
    return -1;
  }
  if (xk) *xk = f64_from_order_key(key);
  if (n_le) *n_le = (int64_t)(less + eq);
  /* rank k0+1 is the same value if it is still inside the run of equal keys */
  if (xk1) *xk1 = ((unsigned long long)k0 + 1 < less + eq || next == ~0ull) ? f64_from_order_key(key) : f64_from_order_key(next);
  return 0;
}
int abz_select_impl(abcdez_ctx* ctx, const double* delta, const uint8_t* alive, int64_t N, int64_t k0,
                    double* xk, double* xk1, int64_t* n_le) {
  int rc = select_enqueue(ctx, delta, alive, N, k0);
  if (rc) return rc;
  rc = read_scalars(ctx);
  if (rc) return rc;
  return select_finish(ctx, k0, xk, xk1, n_le);
}

/* ================================================================ extrema / counts (S10) */

__global__ __launch_bounds__(ABZ_BLOCK) void extrema_kernel(const double* __restrict__ delta, int64_t N,
                                                            unsigned long long* __restrict__ mn,
                                                            unsigned long long* __restrict__ mx) {
  unsigned long long lo = ~0ull, hi = 0ull;
  const int64_t stride = (int64_t)gridDim.x * ABZ_BLOCK;
  for (int64_t k0 = (int64_t)blockIdx.x * ABZ_BLOCK + threadIdx.x; k0 < N; k0 += 8 * stride) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = k0 + u * stride < N ? delta[k0 + u * stride] : delta[k0];   /* 8 loads in flight */
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned long long key = f64_order_key(v[u]);
      lo = key < lo ? key : lo;
      hi = key > hi ? key : hi;
    }
  }
  block_minmax_u64(lo, hi);
  if (threadIdx.x == 0) { atomicMin(mn, lo); atomicMax(mx, hi); }
}

__global__ __launch_bounds__(ABZ_BLOCK) void count_gt_kernel(const double* __restrict__ delta, int64_t N, double thr,
                                                             unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  const int64_t stride = (int64_t)gridDim.x * ABZ_BLOCK;
  for (int64_t k = (int64_t)blockIdx.x * ABZ_BLOCK + threadIdx.x; k < N; k += stride) c += delta[k] > thr;
  c = block_sum_u64(c);
  if (threadIdx.x == 0 && c) atomicAdd(out, c);
}

__global__ __launch_bounds__(ABZ_BLOCK) void count_alive_kernel(const uint8_t* __restrict__ alive, int64_t N,
                                                                unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  const int64_t stride = (int64_t)gridDim.x * ABZ_BLOCK;
  for (int64_t k = (int64_t)blockIdx.x * ABZ_BLOCK + threadIdx.x; k < N; k += stride) c += alive[k] != 0;
  c = block_sum_u64(c);
  if (threadIdx.x == 0 && c) atomicAdd(out, c);
}

int abz_count_alive_impl(abcdez_ctx* ctx, const uint8_t* alive, int64_t N, int64_t* count) {
  ABZ_HIP_CHECK(hipMemsetAsync(ctx->d_scal + ABZ_S_COUNT, 0, 8, ctx->stream));
  unsigned grid = (unsigned)((N + ABZ_BLOCK - 1) / ABZ_BLOCK);
  if (grid > ABZ_REDUCE_GRID) grid = ABZ_REDUCE_GRID;
  hipLaunchKernelGGL(count_alive_kernel, dim3(grid), dim3(ABZ_BLOCK), 0, ctx->stream, alive, N,
                     ctx->d_scal + ABZ_S_COUNT);
  ABZ_HIP_CHECK(hipGetLastError());
  int rc = read_scalars(ctx);
  if (rc) return rc;
  *count = (int64_t)ctx->h_scal[ABZ_S_COUNT];
  return 0;
}

static int extrema_enqueue(abcdez_ctx* ctx, const double* delta, int64_t N) {
  unsigned long long init[2] = {~0ull, 0ull};
  ABZ_HIP_CHECK(hipMemcpyAsync(ctx->d_scal + ABZ_S_MIN, init, 16, hipMemcpyHostToDevice, ctx->stream));
  /* few blocks: every block ends with two same-address atomics, which serialise (2048 blocks spent 45 us on them) */
  unsigned grid = (unsigned)((N + 8 * ABZ_BLOCK - 1) / (8 * ABZ_BLOCK));
  if (grid > 256u) grid = 256u;
  hipLaunchKernelGGL(extrema_kernel, dim3(grid), dim3(ABZ_BLOCK), 0, ctx->stream, delta, N, ctx->d_scal + ABZ_S_MIN,
                     ctx->d_scal + ABZ_S_MAX);
  ABZ_HIP_CHECK(hipGetLastError());
  return 0;
}
int abz_extrema_impl(abcdez_ctx* ctx, const double* delta, int64_t N, double* lo, double* hi) {
  int rc = extrema_enqueue(ctx, delta, N);
  if (rc) return rc;
  rc = read_scalars(ctx);
  if (rc) return rc;
  *lo = f64_from_order_key(ctx->h_scal[ABZ_S_MIN]);
  *hi = f64_from_order_key(ctx->h_scal[ABZ_S_MAX]);
  return 0;
}

int abz_count_gt_impl(abcdez_ctx* ctx, const double* delta, int64_t N, double thr, int64_t* count) {
  ABZ_HIP_CHECK(hipMemsetAsync(ctx->d_scal + ABZ_S_COUNT, 0, 8, ctx->stream));
  unsigned grid = (unsigned)((N + ABZ_BLOCK - 1) / ABZ_BLOCK);
  if (grid > ABZ_REDUCE_GRID) grid = ABZ_REDUCE_GRID;
  hipLaunchKernelGGL(count_gt_kernel, dim3(grid), dim3(ABZ_BLOCK), 0, ctx->stream, delta, N, thr,
                     ctx->d_scal + ABZ_S_COUNT);
  ABZ_HIP_CHECK(hipGetLastError());
  int rc = read_scalars(ctx);
  if (rc) return rc;
  *count = (int64_t)ctx->h_scal[ABZ_S_COUNT];
  return 0;
}

/* ================================================================ sweep counters: cumulative slots -> per-call counts
 * After a read-back of the whole scalar area: totals of the counter classes over the ABZ_CSLOTS slots; what a call
 * reports is the growth since the previous read-back (abz_ctx.h, ABZ_S_CSLOT0).  Also folds the abcdemc sweep's
 * min / max bank.                                                                                             */
void abz_fold_counters(abcdez_ctx* ctx) {
  unsigned long long tot[ABZ_CSTRIDE] = {0};
  for (int k = 0; k < ABZ_CSLOTS; ++k)
    for (int c = 0; c < ABZ_CSTRIDE; ++c) tot[c] += ctx->h_scal[ABZ_S_CSLOT0 + k * ABZ_CSTRIDE + c];
  ctx->h_scal[ABZ_S_NACC] = tot[ABZ_C_NACC] - ctx->cnt_prev[ABZ_C_NACC];
  ctx->h_scal[ABZ_S_NSIM] = tot[ABZ_C_NSIM] - ctx->cnt_prev[ABZ_C_NSIM];
  ctx->h_scal[ABZ_S_RACC] = tot[ABZ_C_RACC] - ctx->cnt_prev[ABZ_C_RACC];
  ctx->h_scal[ABZ_S_RSIM] = tot[ABZ_C_RSIM] - ctx->cnt_prev[ABZ_C_RSIM];
  ctx->h_scal[ABZ_S_MCGT] = tot[ABZ_C_MCGT] - ctx->cnt_prev[ABZ_C_MCGT];
  ctx->h_scal[ABZ_S_COUNT] = tot[ABZ_C_MCSIM] - ctx->cnt_prev[ABZ_C_MCSIM];
  for (int c = 0; c < ABZ_CSTRIDE; ++c) ctx->cnt_prev[c] = tot[c];
}
/* extrema of the distances the LAST abcdemc sweep left, from the bank it reduced into */
void abz_fold_minmax(abcdez_ctx* ctx, int bank, double* lo, double* hi) {
  unsigned long long mn = ~0ull, mx = 0ull;
  const unsigned long long* m = ctx->h_scal + ABZ_S_MM0 + (size_t)bank * 2 * ABZ_MMSLOTS;
  for (int k = 0; k < ABZ_MMSLOTS; ++k) { mn = m[2 * k] < mn ? m[2 * k] : mn; mx = m[2 * k + 1] > mx ? m[2 * k + 1] : mx; }
  *lo = f64_from_order_key(mn);
  *hi = f64_from_order_key(mx);
}

/* ================================================================ one generation's prologue in ONE enqueue + ONE read-back
 * The statements of the driver between two groups of sweeps (smc:301-311,323 and the extrema of :364), in the order the
 * reference runs them, without returning to the host in between:
 *   extrema(Ds) of the generation that just ended            (history, smc:364)
 *   x_(j), x_(j+1) of the alive distances -> q -> eps         (smc:301; eps_prev / eps_target / alpha come from the host)
 *   incremental weights, normalisation, alive flags, ESS      (smc:305-311, :8)
 *   partition of the packed population, unless ESS < ess_min  (then the host resamples: smc:323-326)
 * The host still owns the schedule: it passes the previous eps and the target in and gets eps, wnorm (for logZ), ESS and
 * n_alive back.  Four blocking read-backs become one.                                                            */
/* the first third of the prologue -- extrema(Ds) over all N, the rank select over the alive prefix, eps of smc:301 -- reads the
 * distances and the alive flags and writes only the select's scratch and scalars: it can be enqueued as soon as the last sweep
 * of a generation is (abcdez_smc_select_ahead), before the host has decided that there will be a next generation */
int abz_prologue_select_enqueue(abcdez_ctx* ctx, const double* delta_all, const uint8_t* alive, int64_t N, int64_t n_prev,
                                double alpha, double eps_prev, double eps_target, int64_t* j_out) {
  /* Julia Statistics.quantile, type 7, over the n_prev alive distances: h = (n-1) p + 1, j = clamp(floor(h), 1, n-1), g = h - j */
  const int64_t n = n_prev;
  const double h = (double)(n - 1) * alpha + 1.0;
  int64_t j = (int64_t)__builtin_floor(h);
  if (j < 1) j = 1;
  if (j > n - 1) j = n - 1 > 1 ? n - 1 : 1;
  const double g = h - (double)j;
  QsTail tail{};
  tail.eps_on = 1; tail.single = n == 1 ? 1 : 0; tail.k0 = (unsigned long long)(j - 1); tail.g = g;
  tail.eps_prev = eps_prev; tail.eps_target = eps_target;
  *j_out = j;
  /* + extrema(Ds) over all N, + eps of smc:301; the n_prev positions of the prefix are the alive particles (the packed
   * population's invariant, which the sweeps rely on too): their flags need not be read */
  return select_enqueue(ctx, delta_all, alive, n_prev, j - 1, &tail, N, true);
}
int abz_prologue_packed_impl(abcdez_ctx* ctx, const double* delta_all, int64_t N, int64_t n_prev, double* wns, uint8_t* alive,
                             double alpha, double eps_prev, double eps_target, double eps_k_old, double ess_min,
                             const uint32_t* bits, uint32_t* bits_other, double* slot0, double* slot1, double* logpi,
                             double* delta_rw, double* out /* eps, q, wnorm, ess, lo, hi */, int64_t* n_alive,
                             int32_t* partitioned) {
  int rc = 0;
  int64_t j = 0;
  const abz_ahead& ah = ctx->ahead;
  if (ah.valid && ah.delta == delta_all && ah.alive == alive && ah.N == N && ah.n_prev == n_prev && ah.alpha == alpha &&
      ah.eps_prev == eps_prev && ah.eps_target == eps_target) {
    j = ah.j;                         /* the select (and the extrema, and eps) were enqueued behind the sweeps of the generation before */
    ctx->n_select_reused++;
  } else {
    ctx->n_select_inline++;
    rc = abz_prologue_select_enqueue(ctx, delta_all, alive, N, n_prev, alpha, eps_prev, eps_target, &j);
    if (rc) return rc;
  }
  ctx->ahead.valid = false;
  unsigned long long seq = 0;
  /* indicator kernel on uniform weights: closed forms, two launches instead of four, the weights filled by the partition */
  const bool fast = ctx->w_uniform && (ctx->h_model.abck == ABZ_K_INDICATOR || ctx->h_model.abck == ABZ_K_INDICATOR_STRICT);
  ctx->w_uniform = fast;              /* the general path leaves the weights as its floating sums made them */
  ctx->n_reweight_fast += fast ? 1 : 0;
  if (fast) {
    rc = abz_ind_reweight_partition(ctx, delta_all, alive, N, n_prev, ess_min, bits, bits_other, slot0, slot1, logpi, delta_rw, wns, &seq);
    if (rc) return rc;
  } else {
    rc = reweight_enqueue(ctx, delta_all, wns, alive, n_prev, eps_k_old, 0.0, (const double*)(ctx->d_scal + ABZ_S_EPS), &seq);
    if (rc) return rc;
    /* everything the host needs is known here: the scalars are published (by the launch that finishes the reweight's sums) BEFORE
     * the partition is enqueued, and the host returns (and enqueues the generation's sweeps) while the partition kernels are still
     * running -- the round trip hides behind them.  A partition error (flags that do not describe a prefix) is reported by the
     * next counter read-back instead. */
    rc = abz_partition_impl(ctx, alive, N, n_prev, 0, bits, bits_other, slot0, slot1, logpi, delta_rw, wns, ctx->d_scal, ess_min, false);
    if (rc) return rc;
  }
  rc = abz_publish_wait(ctx, ABZ_S_SCALARS, seq);
  if (rc) return rc;
  rc = select_finish(ctx, j - 1, nullptr, nullptr, nullptr);
  if (rc) return rc;
  out[0] = scal_f64(ctx, ABZ_S_EPS);
  out[1] = scal_f64(ctx, ABZ_S_QVAL);
  out[2] = scal_f64(ctx, ABZ_S_WNORM);
  out[3] = 1.0 / scal_f64(ctx, ABZ_S_SUMSQ);
  out[4] = f64_from_order_key(ctx->h_scal[ABZ_S_MIN]);
  out[5] = f64_from_order_key(ctx->h_scal[ABZ_S_MAX]);
  *n_alive = (int64_t)scal_f64(ctx, ABZ_S_NALIVE);
  *partitioned = !(*n_alive > 0 && out[3] < ess_min);
  return 0;
}

/* ================================================================ spec arithmetic on the device (test hook) */
__global__ __launch_bounds__(ABZ_BLOCK) void math_eval_kernel(int fn, const abz_tables* __restrict__ T,
                                                              const abz_model* __restrict__ M,
                                                              const double* __restrict__ x, double* __restrict__ y,
                                                              double* __restrict__ y2, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * ABZ_BLOCK + threadIdx.x;
  if (i >= n) return;
  switch (fn) {
    case 0: y[i] = abz_log(x[i]); break;
    case 1: y[i] = abz_exp(x[i]); break;
    case 2: { double s, c; abz_sincos2pi(x[i], &s, &c); y[i] = s; y2[i] = c; break; }
    case 3: y[i] = abz_rint(x[i]); break;
    case 4: y[i] = abz_floor(x[i]); break;
    case 5: y[i] = abz_sqrt(x[i]); break;
    case 7: y[i] = abz_log_tab(x[i], T); break;
    case 8: { double s, c; abz_sincos2pi_tab(x[i], T, &s, &c); y[i] = s; y2[i] = c; break; }
    case 9: y[i] = abz_sqrt_pn(x[i]); break;
    case 10: y[i] = abz_lgamma(x[i]); break;
    case 11: y[i] = abz_prior_logpdf1x(&M->prior[(int)y2[i]], x[i], M->ext); break;   /* log-density of prior factor y2[i] at x[i] */
    default: y[i] = x[i] / y2[i]; break;
  }
}
int abz_math_eval_impl(abcdez_ctx* ctx, int fn, const double* x, double* y, double* y2, int64_t n) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(math_eval_kernel, dim3((unsigned)((n + ABZ_BLOCK - 1) / ABZ_BLOCK)), dim3(ABZ_BLOCK), 0,
                     ctx->stream, fn, ctx->d_tables, ctx->d_model, x, y, y2, n);
  ABZ_HIP_CHECK(hipGetLastError());
  ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}

/* ================================================================ per-particle scalar draws of a sweep (test hook)
 * Runs particle_draws<L> (abz_device.h) -- the code the sweep / replay kernels call -- for particles
 * [i0, i0+n) with alive rank = particle index in a pool of n_pool, for any lane-group width L, so that the
 * tests can check bit for bit that (donor ranks, gamma, log u) do not depend on the lane shape (smc:119-128,145). */
template <int L>
__global__ __launch_bounds__(ABZ_BLOCK) void draws_eval_kernel(const HotModel M, uint32_t i0, uint32_t n, uint32_t n_pool,
                                                               uint32_t sweep, double gamma0, double gsig,
                                                               uint32_t* __restrict__ ra, uint32_t* __restrict__ rb,
                                                               double* __restrict__ g, double* __restrict__ log_u) {
  __shared__ abz_tables s_tab;
  {
    TabStage st;
    st.load(M);
    st.store(s_tab);
  }
  __syncthreads();
  const uint32_t gid = blockIdx.x * ABZ_BLOCK + threadIdx.x;
  const uint32_t grp = gid / L;
  const int j = (int)(gid % L);
  const bool active = grp < n;                       /* inactive groups still take part in the group shuffles */
  const uint32_t i = i0 + (active ? grp : 0u);
  uint32_t a, b;
  double gg, lu;
  particle_draws<L>(&s_tab, M.seed, i, sweep, j, n_pool, i, gamma0, gsig, &a, &b, &gg, &lu);
  if (active && j == 0) { ra[grp] = a; rb[grp] = b; g[grp] = gg; log_u[grp] = lu; }
}
int abz_draws_eval_impl(abcdez_ctx* ctx, int lanes, uint32_t i0, uint32_t n, uint32_t n_pool, uint32_t sweep,
                        double gamma0, double gsig, uint32_t* ra, uint32_t* rb, double* g, double* log_u) {
  if (n == 0) return 0;
  const unsigned grid = (unsigned)(((uint64_t)n * (uint64_t)lanes + ABZ_BLOCK - 1) / ABZ_BLOCK);
#define ABZ_DRAWS(LL)                                                                                                    \
  case LL:                                                                                                               \
    hipLaunchKernelGGL((draws_eval_kernel<LL>), dim3(grid), dim3(ABZ_BLOCK), 0, ctx->stream, ctx->hot, i0, n, n_pool,    \
                       sweep, gamma0, gsig, ra, rb, g, log_u);                                                           \
    break;
  switch (lanes) {
    ABZ_DRAWS(1) ABZ_DRAWS(2) ABZ_DRAWS(4) ABZ_DRAWS(8) ABZ_DRAWS(16)
    default: abz_set_error("draws_eval: lanes must be 1, 2, 4, 8 or 16"); return -1;
  }
#undef ABZ_DRAWS
  ABZ_HIP_CHECK(hipGetLastError());
  ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}
