/* abz_ctx.h -- internal context of libabcdez_hip.so (not part of the ABI). */
#ifndef ABZ_CTX_H
#define ABZ_CTX_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "abcdez_spec.h"
#include "abz_hotmodel.h"

#define ABZ_GROUP_MAX 16    /* sweeps per abcdez_smc_sweeps_packed call */
#define ABZ_MC_RING 8       /* abcdemc generations in flight (abcdez_mc_generation_async) */
/* per generation: [0] #(Ds > eps_target), [1] nsim -- cumulative totals of this GPU's counter slots, or, for a generation of a sharded
 * population ([7] != 0), that generation's counts over ALL ranks --, [2] min key, [3] max key, [4] eps_pop, [5] tail length, [6] error
 * word, [7] sharded, [8] [9] this GPU's cumulative slot totals (sharded generations: the host's baselines), [10] spare, [11] ticket + 1 */
#define ABZ_RING_WORDS 12
#define ABZ_MAX_N 0x7FFFFFFFll /* particle indices are 32-bit on the device */
enum { ABZ_COMM_NONE = 0, ABZ_COMM_RCCL = 1, ABZ_COMM_HOST = 2 };   /* abcdez_comm_kind */

/* abcdez_smc_select_ahead: armed = start the next generation's select behind the next grouped sweeps; valid = it has been
 * enqueued for exactly these arguments and nothing has touched the distances / flags since */
struct abz_ahead {
  bool armed = false, valid = false;
  const double* delta = nullptr;
  const uint8_t* alive = nullptr;
  int64_t N = 0, n_prev = 0, j = 0;
  double alpha = 0.0, eps_prev = 0.0, eps_target = 0.0;
};

/* One abcdemc generation (rank pass + sweep + snapshot: up to 15 dependent launches, no data-dependent host decision,
 * src/abcdez_mc.jl:134-161) captured as a HIP graph.  A chain of dependent small kernels runs at ~3.2 us per kernel as stream
 * launches and ~2.2 us as a graph (profiles/r03_launch_floor.jsonl).  Everything that changes from one generation to the next
 * is either part of the key (buffer parity, the min / max bank, which sorts the rank pass launches and how large) or read on
 * the device (the generation's sequence number ABZ_S_MCSEQ -> RNG epoch, ring slot, ticket): a replay needs no patching. */
struct abz_mc_graph_key {
  const void *theta, *logpi, *delta, *ntheta, *nlogpi, *ndelta, *order, *sorted_delta, *cnt, *stamp_cur, *stamp_nxt, *ws, *stream;
  int64_t N;
  double alpha, eps_target, gamma0, gsig;
  uint32_t sweep_base, ltiles;
  int32_t do_rank, path, mm_bank, L, C;
};
struct abz_mc_graph {
  abz_mc_graph_key key;
  hipGraphExec_t exec = nullptr;
  const uint32_t* rank_state = nullptr;
  uint32_t rank_limit = 0xFFFFFFFFu;
};

struct abcdez_ctx {
  int device = 0;
  /* abcdez_ctx_set_graphs: replay abcdemc generations as HIP graphs.  DEFAULT OFF: measured slower on ROCm 7.2 / MI355X -- a graph's
   * kernel nodes take what the same kernels take as stream launches (the 4-5 us of a small dependent kernel are the dispatch itself,
   * not the host) and every graph launch adds ~16 us before its first node: 0.140 against 0.125 ms per generation
   * (profiles/r04_mc1d_graph_replay_ab.json, r04_mc1d_generation_timeline_graph_replay.txt).  ABZ_GRAPHS=1 turns it on. */
  bool graphs_on = false;
  std::vector<abz_mc_graph> mc_graphs;
  long long n_graph_replays = 0, n_graph_captures = 0, n_graph_direct = 0;
  bool mc_seq_dirty = false;            /* a generation failed half way: ABZ_S_MCSEQ must be set to mc_issued again */
  /* the alive particles' weights are uniform, 1 / n_alive (abcdez_ctx_set_uniform_weights; kept by the indicator fast path of the
   * prologue, set by a resampling, cleared by a general reweight): lets the prologue use the closed forms for indicator kernels */
  bool w_uniform = false;
  long long n_reweight_fast = 0;        /* prologues that took the closed forms of the indicator reweight */
  long long n_select_reused = 0, n_select_inline = 0;   /* prologues that found their select enqueued ahead / ran it themselves */
  HotModel hot;                   /* by-value kernel argument, pointers are device pointers */
  hipStream_t stream = nullptr;
  abz_model h_model;              /* host copy; .data points at d_data          */
  abz_model* d_model = nullptr;
  double* d_data = nullptr;
  double* d_mv = nullptr;             /* maps of a correlated Normal prior (abz_model.mv), else null */
  double* d_ext = nullptr;            /* records of the wrapper prior families (abz_model.ext), else null */
  abz_tables* d_tables = nullptr;
  int n_cu = 1;                       /* compute units of the device */
  std::unordered_map<const void*, int> occ;   /* kernel -> resident workgroups per CU (abz_persistent_grid) */
  int L = 1, C = 1;               /* lane-group shape: ld = L*C                 */
  abz_ahead ahead;
  bool prior_plain = false;       /* all real dimensions continuous Normal priors (abz_api.hip)                 */
  /* device scalars + pinned host mirror */
  void* d_block = nullptr;                /* ONE allocation behind d_scal, d_sync, d_model, d_tables, d_data, d_mv (abcdez_ctx_create) */
  unsigned long long* d_scal = nullptr;   /* ABZ_S_N x u64                      */
  unsigned int* d_sync = nullptr;         /* ABZ_SYNC_N tickets of "the last block to finish does X" kernels, zero between launches */
  unsigned long long* h_scal = nullptr;   /* pinned + mapped: ABZ_S_N words + the sequence word of abz_publish */
  unsigned long long* h_scal_dev = nullptr;   /* the same memory as the device sees it */
  unsigned long long pub_seq = 0;
  /* growable workspace */
  void* ws = nullptr;
  size_t ws_bytes = 0;
  void* lv_hand = nullptr;              /* two-launch sweep (Lotka-Volterra, user simulators): the list between the launches (abz_kernels.h, LvHandList), 8 ld + 28 B per position */
  size_t lv_hand_cap = 0;               /* positions the list has room for */
  unsigned long long lv_seq = 0;        /* sweeps launched: parity picks the list counter (abz_smc_swarm.hip) */
  bool user_one_kernel = false;         /* ABZ_USER_ONE_KERNEL=1: user simulators stay on the one-kernel two-phase sweep */
  /* hiprtc-compiled kernels (abz_jit.hip): of a user-supplied simulator, or of a built-in one whose model has prior factors of the
   * wrapper families (truncated(...), MixtureModel) -- the statically compiled sweeps do not carry those; else null */
  void* user_module = nullptr;
  /* quantile select: its own histogram (left zeroed by every call) and the arrays the device-side window belongs to */
  uint32_t* sel_hist = nullptr;
  bool sel_clean = false;
  const void* sel_delta = nullptr;
  const void* sel_alive = nullptr;
  int64_t sel_N = 0;
  /* running totals of the cumulative counter slots at the last read-back (ABZ_S_CSLOT0), by counter class */
  unsigned long long cnt_prev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  /* min / max slot bank the NEXT abcdemc sweep reduces into (the sweep resets the other one) */
  int mm_bank = 0;
  /* blob stamps of the current / next generation (abcdez_ctx_set_stamps); null = blobs off.  The launchers that
   * take (logpi, nlogpi)-style pairs use (stamp_cur, stamp_nxt) alongside; the packed sweeps update stamp_cur in place */
  uint64_t* stamp_cur = nullptr;
  uint64_t* stamp_nxt = nullptr;
  /* optional HIP-event timing of the sweep kernel (bench.py's roofline figure) */
  bool timing = false, timing_first_only = false;
  bool timing_group = false;                      /* mode 3: ONE pair around all the sweeps of an abcdez_smc_sweeps_packed call */
  int ev_group[ABZ_GROUP_MAX] = {0};              /* the pair brackets a whole group: counts `done` launches of ev_units updates each */
  int timing_stride = 1;                          /* of the launches that would be bracketed only every timing_stride-th is */
  long long timing_seq = 0, timing_rot = 0;
  bool ring_timed[ABZ_MC_RING] = {false};         /* was the sweep of that asynchronous generation bracketed */
  long long ev_head = 0, ev_tail = 0;             /* FIFO of sweeps timed but not yet read: pair k lives in slot k % ABZ_GROUP_MAX */
  hipEvent_t ev[2 * ABZ_GROUP_MAX] = {nullptr};
  long long ev_units[ABZ_GROUP_MAX] = {0};
  int ev_sweep[ABZ_GROUP_MAX] = {0};              /* which sweep of its group the pair brackets (0 outside groups) */
  int cur_sweep_k = 0;                            /* set by the callers of the sweep launcher */
  /* abcdemc generations enqueued without a host synchronisation (abcdez_mc_generation_async): a ring of pinned snapshots of
   * the scalar area, one per generation in flight, each behind its own event; tickets are issued and redeemed in order */
  unsigned long long* h_ring = nullptr;           /* ABZ_MC_RING x ABZ_RING_WORDS u64, pinned + mapped: written BY a kernel */
  unsigned long long* d_ring = nullptr;           /* the same memory as the device sees it */
  bool ring_folded[ABZ_MC_RING] = {false};
  bool mc_chain_sharded = false;                  /* the chain's generations sweep this rank's particles only (abcdez_mc_generation_sharded_async) */
  long long ring_res[ABZ_MC_RING][2] = {{0, 0}};  /* (nsim, #above target) once folded */
  long long ring_chain[ABZ_MC_RING] = {0};        /* which chain of generations the ticket belongs to (mc_chain when it was issued) */
  double ring_eps_target[ABZ_MC_RING] = {0.0};
  /* A chain = consecutive asynchronous generations on one population with one (alpha, eps_target); anything else that writes
   * distances breaks it.  Once a generation of the chain ran with eps_pop == eps_target, the particles that draw (Ds > eps_pop)
   * can only become fewer: the converged ones never leave (they accept only dp <= eps_target, mc:52), the others accept only
   * dp <= Ds[i], and lo + alpha (hi - lo) does not grow -- so that generation's tail length BOUNDS every later one's, and the
   * rank pass launches only the path that bound calls for (abz_sort.hip). */
  long long mc_chain = 0;
  const void* mc_last_out = nullptr;              /* distances the last asynchronous generation wrote, and their length */
  int64_t mc_last_N = 0;
  long long n_rank_paths[3] = {0, 0, 0};          /* rank passes that launched both sorts / only the LDS sort / only the radix sort */
  long long mc_tail_bound = -1;                   /* proved upper bound of the tail length of the chain's next generations; -1 = none */
  long long mc_issued = 0, mc_waited = 0;
  bool mc_have_bank = false;
  bool mc_window_ready = false;                   /* the last snapshot kernel left the next generation's window for (mc_alpha, mc_eps_target) */
  double mc_alpha = 0.0, mc_eps_target = 0.0;                      /* a sweep of this context has left extrema in a bank */
  /* an open group of sharded sweeps (abcdez_smc_group_begin .. _end): sweeps enqueued so far (-1 = none open) */
  int grp_k = -1;
  int64_t grp_n_alive = 0;
  double grp_kmin = 0.0;
  unsigned long long grp_base_acc = 0, grp_base_sim = 0, grp_pub = 0;
  const uint32_t* mc_rank_state = nullptr;        /* state words of the last rank pass (abz_sort.hip), in the workspace */
  uint32_t mc_rank_limit = 0xFFFFFFFFu;           /* longest tail the sorts that pass launched can handle (only the LDS sort: 4096) */
  long long mc_tail_hint = -1;                    /* particles that drew in the last generation the host has seen; -1 = unknown */
  /* ABZ_S_MC_NABOVE describes the population of this chain (-1: count it before the next asynchronous generation) */
  long long mc_nabove_chain = -1;
  /* a redeemed generation of the chain left at least 1 / 16 of the particles at or below eps_target: from then on every generation
   * of the chain draws by rejection (the count never grows, include/abcdez_spec.h) and no rank pass is launched */
  bool mc_reject_known = false;
  long long n_mc_reject_gens = 0;                 /* asynchronous generations issued without a rank pass for that reason */
  /* the last abcdez_count_gt (the driver's mc:133, right before its loop): if the next chain starts from these very distances
   * with thr as its eps_target, the host side knows the chain's first count without waiting for a generation to be redeemed */
  struct { const void* delta = nullptr; int64_t N = 0; double thr = 0.0; int64_t count = -1; long long chain = -1; } mc_count_seen;
  double swarm_ms = 0.0;
  long long swarm_launches = 0, swarm_units = 0;
  /* serpentine sweeps: every other launch of the packed sweep walks the prefix from its end (abz_kernels.h, SmcPackedArgs.rev);
   * ABZ_SERPENTINE=0 in the environment keeps every launch front to back (same results; A/B measurements) */
  bool serpentine = true;
  long long sweep_launch_seq = 0;
  /* multi-GPU (abz_comm.hip): the transport of this context's rank -- RCCL (comm = an ncclComm_t) or the host's callbacks
   * (abcdez_comm_init_host; hc_stage = the page-locked block the pieces travel through) --, none on a single GPU.  comm_broken: a
   * sharded call failed on this rank between its collectives; every later collective returns an error (abz_comm_abort_after_failure) */
  void* comm = nullptr;
  int comm_kind = ABZ_COMM_NONE;
  bool comm_broken = false;
  int comm_rank = 0, comm_world = 1;
  int (*hc_allgather)(void*, void*, int64_t) = nullptr;
  int (*hc_allreduce)(void*, void*, int64_t, int32_t, int32_t) = nullptr;
  void* hc_user = nullptr;
  void* hc_stage = nullptr;
  size_t hc_stage_bytes = 0;
};

void abz_set_error(const std::string& msg);

/* Thread indices are 32-bit on the device (gid = tile * BLOCK + threadIdx.x in the sweep, init, gather, partition and replay bodies;
 * L lanes of a wavefront own one particle): a launch over n particles must stay below 2^32 THREADS, which n <= ABZ_MAX_N alone
 * does not guarantee (2^30 particles at 8 lanes wrap silently).  Checked by every entry point that launches lane-group kernels,
 * before anything is enqueued. */
#define ABZ_REQUIRE_LANES(ctx, n, who)                                                                                        \
  do {                                                                                                                        \
    if ((uint64_t)(n) * (uint64_t)(ctx)->L >= (1ull << 32)) {                                                                \
      abz_set_error(std::string(who) + ": " + std::to_string((long long)(n)) + " particles x " + std::to_string((ctx)->L) +    \
                    " lanes per particle is 2^32 threads or more -- thread indices are 32-bit on the device: use fewer lanes " \
                    "(abcdez_ctx_set_lanes) or shard the population over more GPUs");                                          \
      return -1;                                                                                                              \
    }                                                                                                                         \
  } while (0)

#define ABZ_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      abz_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                  \
      return -2;                                                                         \
    }                                                                                    \
  } while (0)

/* Grid of a kernel whose workgroups loop over tiles (ABZ_TILE_LOOP): as many workgroups as the device holds at once --
 * compute units x the occupancy the runtime reports for this kernel -- each with the same number of tiles (+-1), so the
 * model tables are staged into LDS once per resident workgroup.  Used by the kernels that are not memory-bound on long rows
 * (initial population, abcdemc sweep, blobs); the SMC sweep keeps one tile per workgroup: the hardware's dynamic dispatch of
 * many small workgroups keeps every CU busy to the end of the launch, which a static deal of ~18 tiles per wavefront does not
 * (profiles/r03_bench_ab_round2_vs_icdf_persistent.json). */
static inline unsigned abz_tiles_to_grid(uint64_t ntiles, uint64_t resident) {
  if (resident < 1) resident = 1;
  if (ntiles <= resident) return (unsigned)(ntiles ? ntiles : 1);
  const uint64_t per = (ntiles + resident - 1) / resident;
  return (unsigned)((ntiles + per - 1) / per);
}
template <class K>
static inline unsigned abz_persistent_grid(abcdez_ctx* ctx, K kernel, uint64_t ntiles, int block) {
  const void* key = reinterpret_cast<const void*>(kernel);
  auto it = ctx->occ.find(key);
  int per_cu;
  if (it != ctx->occ.end()) per_cu = it->second;
  else {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess || nb < 1) nb = 1;
    per_cu = nb;
    ctx->occ.emplace(key, per_cu);
  }
  return abz_tiles_to_grid(ntiles, (uint64_t)ctx->n_cu * (uint64_t)per_cu);
}

/* workspace: returns a device pointer to at least `bytes` (256-B aligned) */
int abz_ws_reserve(abcdez_ctx* ctx, size_t bytes);
int abz_lv_hand_reserve(abcdez_ctx* ctx, size_t positions);
bool abz_sweep_in_two_launches(const abcdez_ctx* ctx);

static inline size_t abz_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

/* tickets in d_sync */
enum { ABZ_SYNC_TILE = 0, ABZ_SYNC_N = 16 };

/* scalar slots in d_scal / h_scal */
enum {
  ABZ_S_NACC = 0, ABZ_S_NSIM = 1, ABZ_S_WNORM = 2, ABZ_S_SUMSQ = 3, ABZ_S_NALIVE = 4,
  ABZ_S_MIN = 5, ABZ_S_MAX = 6, ABZ_S_COUNT = 7, ABZ_S_LASTPOS = 8, ABZ_S_SUM = 9,
  /* state of the quantile select (abz_population.hip): result (key of the rank, #keys below, #equal, next larger key),
   * rank inside the selected bin, that bin, buffer fill (reset by every call), error flag, and the binning window [HLO, HHI]
   * carried from call to call */
  ABZ_S_SEL_PREFIX = 10, ABZ_S_SEL_K = 11, ABZ_S_SEL_LESS = 12, ABZ_S_SEL_EQ = 13, ABZ_S_SEL_NEXT = 14,
  ABZ_S_SEL_NBUF = 15, ABZ_S_SEL_PAD = 16, ABZ_S_SEL_HLO = 17, ABZ_S_SEL_HHI = 18, ABZ_S_SEL_BIN = 19,
  ABZ_S_SEL_END = 20,
  ABZ_S_INITBAD = 20,
  ABZ_S_RACC = 21, ABZ_S_RSIM = 22,      /* counters of the replayed ranks (sharded packed population) */
  ABZ_S_MCGT = 23, ABZ_S_MCMIN = 24, ABZ_S_MCMAX = 25,
  ABZ_S_PART_H = 26, ABZ_S_PART_F = 27, ABZ_S_PART_ERR = 28,
  ABZ_S_EPS = 29, ABZ_S_QVAL = 30,       /* fused prologue: eps of smc:301 and the quantile behind it (f64) */   /* partition: #holes, #fillers (must agree), error flag */   /* abcdemc sweep: #(Ds > eps_target), extrema of the new distances */
  ABZ_S_PARITY = 31,                     /* resampling of double-buffered rows: the slot parity every position gets */
  ABZ_S_SCALARS = 32,
  /* Sweep counters (smc:138,150,352; mc:44,156): every block of a sweep / replay kernel ADDS its counts to one of
   * ABZ_CSLOTS slots (one 64-byte line each, chosen by block index; agent-scope atomics, fire and forget).  The
   * slots are cumulative -- never zeroed -- and the host takes differences of their totals between read-backs, so
   * a sweep costs no memset, no reduction launch and no second pass (abz_device.h: block_count2).              */
  ABZ_S_CSLOT0 = ABZ_S_SCALARS,
  /* min / max of the distances an abcdemc sweep leaves (mc:146,163): two banks of ABZ_MMSLOTS (min key, max key)
   * pairs; a sweep reduces into one bank and resets the other for its successor                                */
  ABZ_S_MM0 = ABZ_S_CSLOT0 + ABZ_CSLOTS * ABZ_CSTRIDE,
  /* a group of sweeps enqueued behind the device-side test of smc:352 (abcdez_smc_sweeps_packed): stop flag, number of
   * sweeps that ran, and the totals of the (nacc, nsim) counter slots after each of them                             */
  ABZ_S_GRP_STOP = ABZ_S_MM0 + 2 * ABZ_MMSLOTS * 2,
  ABZ_S_GRP_DONE = ABZ_S_GRP_STOP + 1,
  ABZ_S_GRP_SNAP = ABZ_S_GRP_STOP + 2,
  /* abcdemc generation enqueued without host values (abcdez_mc_generation_async): eps_pop of mc:147 (f64), the binning
   * window of the rank pass (key of eps_pop, shift) and the extrema it was made from (f64 lo, hi)                     */
  ABZ_S_MCW_EPS = ABZ_S_GRP_SNAP + 2 * ABZ_GROUP_MAX,
  ABZ_S_MCW_KLO = ABZ_S_MCW_EPS + 1, ABZ_S_MCW_SHIFT = ABZ_S_MCW_EPS + 2, ABZ_S_MCW_LO = ABZ_S_MCW_EPS + 3,
  ABZ_S_MCW_HI = ABZ_S_MCW_EPS + 4,
  /* number of asynchronous abcdemc generations whose snapshot kernel has run (== the host's mc_issued when the next one
   * executes): ring slot and ticket of the snapshot, RNG epoch of the sweep = base + this */
  ABZ_S_MCSEQ = ABZ_S_MCW_EPS + 5,
  /* how the next asynchronous generation draws its better particles (include/abcdez_spec.h, abz_mc_draws_by_rejection):
   * #(Ds > eps_target) of the distances it reads -- counted at the start of a chain, afterwards the growth of the cumulative
   * ABZ_C_MCGT slots over the sweep before (their total at the last snapshot is kept next to it) */
  ABZ_S_MC_NABOVE = ABZ_S_MCW_EPS + 6, ABZ_S_MC_TGPREV = ABZ_S_MCW_EPS + 7,
  ABZ_S_MC_REJFAIL = ABZ_S_MCW_EPS + 8,   /* a sweep drawing by rejection ran out of trials: the population was not what the rule assumed */
  /* asynchronous generations of a SHARDED population (abcdez_mc_generation_sharded_async): this GPU's ABZ_C_MCSIM total at the last
   * snapshot (TGPREV holds the ABZ_C_MCGT one) and the words the ranks exchange per generation -- [0] [1] this generation's counts
   * (all-reduced by sum), [2] min key, [3] ~max key, [4] ~fail word (all-reduced by min), [5] [6] the local totals (not reduced) */
  ABZ_S_MC_TSPREV = ABZ_S_MCW_EPS + 9,
  ABZ_S_MC_PART = ABZ_S_MCW_EPS + 10,
  ABZ_S_N = ABZ_S_MCW_EPS + 18
};

/* kernel launchers implemented across the .hip files */
int abz_launch_init(abcdez_ctx*, double*, double*, double*, int64_t, int64_t);
/* HIP-event timing of the sweep kernels: bracket a launch; the pairs are read at the next counter read-back */
static inline int abz_time_begin(abcdez_ctx* ctx) {
  if (!ctx->timing || ctx->ev_tail - ctx->ev_head >= ABZ_GROUP_MAX) return -1;
  if (ctx->timing_stride > 1 && (ctx->timing_seq++ % ctx->timing_stride) != 0) return -1;
  const int k = (int)(ctx->ev_tail % ABZ_GROUP_MAX);
  (void)hipEventRecord(ctx->ev[2 * k], ctx->stream);
  return k;
}
static inline void abz_time_end(abcdez_ctx* ctx, int k, long long units) {
  if (k < 0) return;
  (void)hipEventRecord(ctx->ev[2 * k + 1], ctx->stream);
  ctx->ev_units[k] = units;
  ctx->ev_group[k] = 0;
  ctx->ev_sweep[k] = ctx->cur_sweep_k;
  ctx->ev_tail += 1;
}
int abz_launch_mc_swarm(abcdez_ctx*, const uint32_t*, const uint32_t*, uint32_t, const double*, const double*,
                        const double*, double*, double*, double*, double, double, double, double,
                        uint32_t, uint32_t, uint32_t, const unsigned long long*, const unsigned long long* seq_dev = nullptr,
                        const unsigned long long* nabove_dev = nullptr, int rank_built = -1 /* -1: iff order and cnt are given */);
int abz_launch_mc_window(abcdez_ctx*, int, double, double, double, double);
int abz_launch_mc_snapshot(abcdez_ctx*, int bank, unsigned long long* d_ring, double alpha, double eps_target,
                           const uint32_t* rank_state, uint32_t N, int sharded = 0);
int abz_launch_mc_partial(abcdez_ctx*, int bank);
/* abz_comm.hip: what the ranks exchange after the own-range sweep of an abcdemc generation -- the new rows / log-priors / distances
 * (/ blob stamps) of every rank's particles, in place, and the ABZ_S_MC_PART words -- as ONE group of collectives on the stream */
int abz_comm_mc_exchange(abcdez_ctx*, double* ntheta, double* nlogpi, double* ndelta, uint64_t* nstamp, int64_t n_local, int ld);
void abz_comm_abort_after_failure(abcdez_ctx*);
int abz_launch_mc_chain_start(abcdez_ctx*, const double* delta, int64_t N, double eps_target);
/* which sorts a rank pass launches for a tail of the hinted / bounded length, and the grid of the long-tail kernels (a power of two
 * of wave-tiles: the kernels stride, so any grid is correct -- few distinct values keep the graph cache small) */
struct abz_rank_plan { bool small_path, long_path; uint32_t ltiles; size_t ws_bytes; };
abz_rank_plan abz_rank_plan_for(int64_t N, int64_t tail_hint, int64_t tail_bound);
int abz_launch_push_p(abcdez_ctx*, const double*, int64_t, double*);
void abz_fold_counters(abcdez_ctx*);
/* Read-back of the first `nwords` device scalars WITHOUT the copy engine and without a stream synchronisation: a one-block
 * kernel writes them straight into the pinned host mirror and stores a sequence word last (system-scope release); the
 * host polls that word.  Returns when everything enqueued before it has completed.                                  */
int abz_publish(abcdez_ctx* ctx, int nwords);
/* Host side of a read-back through pinned memory: waits until *word == expected.  Spins with a pause instruction for up to
 * 4 ms (the waits of the hot loop), then yields the core between looks, asks the stream now and then, and after 2 s blocks in
 * hipStreamSynchronize like the copy-engine path would, with no limit of its own.  0 = the word arrived, 1 = the stream
 * drained without it (a failed launch), < 0 = HIP error. */
int abz_poll_word(abcdez_ctx* ctx, const unsigned long long* word, unsigned long long expected);
int abz_publish_launch(abcdez_ctx* ctx, int nwords, unsigned long long* seq_out);
int abz_publish_wait(abcdez_ctx* ctx, int nwords, unsigned long long seq);
int abz_prologue_select_enqueue(abcdez_ctx* ctx, const double* delta_all, const uint8_t* alive, int64_t N, int64_t n_prev,
                                double alpha, double eps_prev, double eps_target, int64_t* j_out);
void abz_fold_minmax(abcdez_ctx*, int bank, double* lo, double* hi);
int abz_jit_build(abcdez_ctx*, const char* user_source);
void abz_jit_destroy(abcdez_ctx*);
int abz_jit_launch_init(abcdez_ctx*, double*, double*, double*, uint32_t, uint32_t, unsigned long long*, uint64_t* stamp);
int abz_jit_launch_blob(abcdez_ctx*, const double* theta, const uint64_t* stamp, uint32_t n, double* blob,
                        double* delta_out, uint32_t nbw);
int abz_launch_blob_eval(abcdez_ctx*, const double* theta, const uint64_t* stamp, int64_t n, double* blob,
                         double* delta_out, uint32_t nbw);
int abz_jit_launch_mc(abcdez_ctx*, const void* args, unsigned ntiles);
int abz_jit_launch_smc_packed(abcdez_ctx*, const void* args, unsigned nblocks);
bool abz_jit_has_smc_split(abcdez_ctx*);
bool abz_jit_has_rounds(abcdez_ctx*);
bool abz_jit_has_replay(abcdez_ctx*);
int abz_jit_launch_replay(abcdez_ctx*, const void* args, unsigned nblocks);
unsigned abz_jit_smc_block(abcdez_ctx*);
int abz_jit_launch_smc_split(abcdez_ctx*, const void* args, const void* list, unsigned nblocks);

#endif
