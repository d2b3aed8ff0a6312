/*
 * abz_comm.hip -- the multi-GPU exchange steps of the population loop behind the C ABI, over one of TWO transports:
 *
 *   RCCL over xGMI (abcdez_comm_init): collectives issued on the context's OWN stream between the kernels they depend on (no
 *     second stream, no event hand-over, no host framework).  librccl is opened lazily (dlopen) by the first call that needs it:
 *     a single-GPU host, or one that exchanges through its own transport, loads libabcdez_hip.so without RCCL installed.
 *   a host-supplied transport (abcdez_comm_init_host): the library stages this rank's piece through page-locked memory,
 *     waits for its stream, calls the host's all-gather (MPI.Allgather!, gloo, shared memory ...) on HOST memory and copies the
 *     other ranks' pieces back -- the same in-place semantics at the same points of the same entry points.  For hosts without
 *     RCCL, for ranks that share one GPU (RCCL refuses two ranks on one device: how this repository's tests run worlds of 2 and
 *     4 on a one-GPU box), and as the reference transport the RCCL path is checked against.
 *
 * The reference parallelises its loops over threads of one process (`@floop ex for i in 1:nparticles`, src/abcdez_smc.jl:110,
 * src/abcdez_mc.jl:7, src/abcdez_init.jl:6); here the loop is sharded over GPUs, one process per GPU, and what the threads of
 * the reference share through memory -- every particle may be anybody's donor (smc:119-126, mc:23-32) -- is exchanged once per
 * sweep.  SURVEY.md section 8e / DESIGN.md section 7: every rank keeps the whole population; per sweep ONE BYTE per alive
 * position crosses the fabric (the accept flag; the accepted rows are rebuilt on every replica), per generation the distances.
 *
 *   abcdez_comm_unique_id  -- rank 0 makes the 128-byte id; the host carries it to the other ranks by its own means
 *                             (MPI.bcast in Julia, the process group's store under torch.distributed)
 *   abcdez_comm_init       -- every rank: ncclCommInitRank on the context's device
 *   abcdez_comm_init_host  -- every rank: the host's all-gather (and, optionally, all-reduce) callbacks
 *   abcdez_comm_allgather / _allreduce -- the raw in-place collectives on the context's stream (initial population, abcdemc)
 *   abcdez_smc_sweeps_sharded -- the Kmcmc sweeps of one generation of abcdesmc! on a sharded population in ONE call:
 *                             own chunk sweep -> flag all-gather -> replay (+ the device-side test of smc:352), k_max times,
 *                             read-back, distance all-gather
 *
 * The sharded entry points are COLLECTIVE calls: every rank of the communicator must make them with the same arguments.  What can
 * fail on one rank alone (allocations, argument checks) is done before the call's first collective; a failure after that point
 * aborts the communicator (ncclCommAbort / the host transport is marked broken and every later collective of this context
 * returns an error) so that the peers fail instead of waiting for ever.
 */
#include <rccl/rccl.h>          /* types and prototypes only: the functions are looked up with dlsym (abz_rccl) */

#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "abz_ctx.h"
#include "../../include/abcdez_hip.h"

#define ABZ_REQUIRE(cond, msg) \
  do { if (!(cond)) { abz_set_error(msg); return -1; } } while (0)
#define ABZ_NCCL_CHECK(expr)                                                             \
  do {                                                                                   \
    ncclResult_t _r = (expr);                                                            \
    if (_r != ncclSuccess) {                                                             \
      abz_set_error(std::string(#expr) + ": " + R.GetErrorString(_r));                   \
      return -4;                                                                         \
    }                                                                                    \
  } while (0)

static_assert(NCCL_UNIQUE_ID_BYTES == ABCDEZ_COMM_ID_BYTES, "abcdez_hip.h promises the hosts a 128-byte id");

/* ---- librccl, opened on first use ---- */
namespace {
struct abz_rccl {
  void* handle = nullptr;
  bool tried = false;
  std::string why;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
};
abz_rccl R;

template <class F>
bool abz_sym(F& f, const char* name) {
  f = reinterpret_cast<F>(dlsym(R.handle, name));
  if (!f) R.why = std::string("librccl has no symbol ") + name;
  return f != nullptr;
}

int abz_rccl_load() {
  static std::mutex once;                                     /* contexts of several host threads may come here together */
  std::lock_guard<std::mutex> hold(once);
  if (R.handle && R.GroupEnd) return 0;
  if (!R.tried) {
    R.tried = true;
    const char* names[] = {getenv("ABCDEZ_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      R.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (R.handle) break;
      R.why = dlerror();
    }
    if (R.handle) {
      const bool ok = abz_sym(R.GetUniqueId, "ncclGetUniqueId") && abz_sym(R.CommInitRank, "ncclCommInitRank") &&
                      abz_sym(R.CommDestroy, "ncclCommDestroy") && abz_sym(R.CommAbort, "ncclCommAbort") &&
                      abz_sym(R.GetErrorString, "ncclGetErrorString") && abz_sym(R.AllGather, "ncclAllGather") &&
                      abz_sym(R.AllReduce, "ncclAllReduce") && abz_sym(R.GroupStart, "ncclGroupStart") &&
                      abz_sym(R.GroupEnd, "ncclGroupEnd");
      if (!ok) { dlclose(R.handle); R.handle = nullptr; R.GroupEnd = nullptr; }
    }
  }
  if (R.handle && R.GroupEnd) return 0;
  abz_set_error("RCCL is not available on this host (" + R.why + "): use abcdez_comm_init_host with the host's own all-gather");
  return -4;
}

/* ---- the host transport ---- */
int abz_stage_reserve(abcdez_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->hc_stage_bytes) return 0;
  ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));          /* copies out of the old block may still be in flight */
  if (ctx->hc_stage) (void)hipHostFree(ctx->hc_stage);
  ctx->hc_stage = nullptr; ctx->hc_stage_bytes = 0;
  const size_t want = abz_align(bytes + bytes / 4, 4096);
  ABZ_HIP_CHECK(hipHostMalloc(&ctx->hc_stage, want, hipHostMallocDefault));
  ctx->hc_stage_bytes = want;
  return 0;
}

int abz_host_cb_failed(abcdez_ctx* ctx, const char* what, int rc) {
  ctx->comm_broken = true;
  abz_set_error(std::string(what) + ": the host transport's callback returned " + std::to_string(rc) +
                " (the communicator of this context is broken: abcdez_comm_destroy, then initialise it again)");
  return -4;
}

/* one in-place all-gather of a device array through the staging block at offset `off`; the pieces of the other ranks are copied
 * back asynchronously -- the block is not written again before the stream has been waited for (the next collective does) */
struct abz_host_piece { char* dev; size_t piece; size_t off; };

int abz_host_allgather_many(abcdez_ctx* ctx, const abz_host_piece* p, int n) {
  const size_t G = (size_t)ctx->comm_world, r = (size_t)ctx->comm_rank;
  char* st = (char*)ctx->hc_stage;
  for (int i = 0; i < n; ++i)
    if (p[i].piece) ABZ_HIP_CHECK(hipMemcpyAsync(st + p[i].off + r * p[i].piece, p[i].dev + r * p[i].piece, p[i].piece, hipMemcpyDeviceToHost, ctx->stream));
  ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n; ++i) {
    if (!p[i].piece) continue;
    const int rc = ctx->hc_allgather(ctx->hc_user, st + p[i].off, (int64_t)p[i].piece);
    if (rc) return abz_host_cb_failed(ctx, "comm_allgather", rc);
  }
  for (int i = 0; i < n; ++i) {
    if (!p[i].piece) continue;
    if (r > 0) ABZ_HIP_CHECK(hipMemcpyAsync(p[i].dev, st + p[i].off, r * p[i].piece, hipMemcpyHostToDevice, ctx->stream));
    if (r + 1 < G)
      ABZ_HIP_CHECK(hipMemcpyAsync(p[i].dev + (r + 1) * p[i].piece, st + p[i].off + (r + 1) * p[i].piece, (G - r - 1) * p[i].piece,
                                   hipMemcpyHostToDevice, ctx->stream));
  }
  return 0;
}

template <class T>
void abz_reduce_rows(T* out, const T* all, size_t n, size_t G, int op) {
  for (size_t j = 0; j < n; ++j) {
    T v = all[j];
    for (size_t g = 1; g < G; ++g) {                  /* rank order: the same result on every rank, whatever the host's transport does */
      const T w = all[g * n + j];
      v = op == 0 ? (T)(v + w) : op == 1 ? (w < v ? w : v) : (w > v ? w : v);
    }
    out[j] = v;
  }
}

/* all-reduce of n 8-byte words ALREADY IN HOST MEMORY at `host` (dtype 0 int64, 1 float64, 2 uint64; op 0 sum, 1 min, 2 max): the
 * host's own all-reduce if it gave one, else an all-gather of the words and a reduction in rank order.  `scratch` has room for
 * world * n words. */
int abz_host_allreduce_words(abcdez_ctx* ctx, void* host, size_t n, int dtype, int op, void* scratch) {
  if (ctx->hc_allreduce) {
    const int rc = ctx->hc_allreduce(ctx->hc_user, host, (int64_t)n, dtype, op);
    return rc ? abz_host_cb_failed(ctx, "comm_allreduce", rc) : 0;
  }
  const size_t G = (size_t)ctx->comm_world, r = (size_t)ctx->comm_rank;
  memcpy((char*)scratch + r * n * 8, host, n * 8);
  const int rc = ctx->hc_allgather(ctx->hc_user, scratch, (int64_t)(n * 8));
  if (rc) return abz_host_cb_failed(ctx, "comm_allreduce (by all-gather)", rc);
  if (dtype == 0) abz_reduce_rows((int64_t*)host, (const int64_t*)scratch, n, G, op);
  else if (dtype == 1) abz_reduce_rows((double*)host, (const double*)scratch, n, G, op);
  else abz_reduce_rows((uint64_t*)host, (const uint64_t*)scratch, n, G, op);
  return 0;
}

int abz_comm_usable(abcdez_ctx* ctx, const char* who) {
  if (ctx->comm_kind == ABZ_COMM_NONE) { abz_set_error(std::string(who) + ": no communicator (abcdez_comm_init / abcdez_comm_init_host)"); return -1; }
  if (ctx->comm_broken) { abz_set_error(std::string(who) + ": the communicator was aborted after a failure on this rank (abcdez_comm_destroy, then initialise it again)"); return -4; }
  return 0;
}

int abz_allgather_impl(abcdez_ctx* ctx, void* buf, size_t piece_bytes) {
  if (piece_bytes == 0) return 0;
  char* b = (char*)buf;
  if (ctx->comm_kind == ABZ_COMM_RCCL) {
    ABZ_NCCL_CHECK(R.AllGather(b + (size_t)ctx->comm_rank * piece_bytes, b, piece_bytes, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream));
    return 0;
  }
  if (int rc = abz_stage_reserve(ctx, piece_bytes * (size_t)ctx->comm_world)) return rc;
  const abz_host_piece p{b, piece_bytes, 0};
  return abz_host_allgather_many(ctx, &p, 1);
}

int abz_allreduce_impl(abcdez_ctx* ctx, void* buf, size_t n, int dtype, int op) {
  if (n == 0) return 0;
  if (ctx->comm_kind == ABZ_COMM_RCCL) {
    const ncclRedOp_t ops[3] = {ncclSum, ncclMin, ncclMax};
    const ncclDataType_t dts[3] = {ncclInt64, ncclFloat64, ncclUint64};
    ABZ_NCCL_CHECK(R.AllReduce(buf, buf, n, dts[dtype], ops[op], (ncclComm_t)ctx->comm, ctx->stream));
    return 0;
  }
  const size_t G = (size_t)ctx->comm_world;
  if (int rc = abz_stage_reserve(ctx, n * 8 * (G + 1))) return rc;
  char* st = (char*)ctx->hc_stage;
  ABZ_HIP_CHECK(hipMemcpyAsync(st, buf, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  ABZ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (int rc = abz_host_allreduce_words(ctx, st, n, dtype, op, st + n * 8)) return rc;
  ABZ_HIP_CHECK(hipMemcpyAsync(buf, st, n * 8, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}
}  // namespace

/* A rank failed between the collectives of a sharded call: its peers are (or will be) waiting in a collective this rank never
 * enqueues.  RCCL: abort the communicator, which fails their pending operations; host transport: nothing of the library's is
 * pending on the peers -- the host owns the transport and its time-outs -- but this context refuses further collectives. */
void abz_comm_abort_after_failure(abcdez_ctx* ctx) {
  if (ctx->comm_kind == ABZ_COMM_NONE || ctx->comm_broken) return;
  ctx->comm_broken = true;
  if (ctx->comm_kind == ABZ_COMM_RCCL && ctx->comm && R.CommAbort) {
    (void)R.CommAbort((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
  }
}

int abz_comm_mc_exchange(abcdez_ctx* ctx, double* ntheta, double* nlogpi, double* ndelta, uint64_t* nstamp, int64_t n_local, int ld) {
  if (int rc = abz_comm_usable(ctx, "mc_generation_sharded_async")) return rc;
  const size_t r = (size_t)ctx->comm_rank, nl = (size_t)n_local, G = (size_t)ctx->comm_world;
  unsigned long long* part = ctx->d_scal + ABZ_S_MC_PART;
  if (ctx->comm_kind == ABZ_COMM_RCCL) {
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    ABZ_NCCL_CHECK(R.GroupStart());
    ncclResult_t rc = R.AllGather(ntheta + r * nl * (size_t)ld, ntheta, nl * (size_t)ld, ncclFloat64, comm, ctx->stream);
    if (rc == ncclSuccess) rc = R.AllGather(nlogpi + r * nl, nlogpi, nl, ncclFloat64, comm, ctx->stream);
    if (rc == ncclSuccess) rc = R.AllGather(ndelta + r * nl, ndelta, nl, ncclFloat64, comm, ctx->stream);
    if (rc == ncclSuccess && nstamp) rc = R.AllGather(nstamp + r * nl, nstamp, nl, ncclUint64, comm, ctx->stream);
    if (rc == ncclSuccess) rc = R.AllReduce(part, part, 2, ncclUint64, ncclSum, comm, ctx->stream);
    if (rc == ncclSuccess) rc = R.AllReduce(part + 2, part + 2, 3, ncclUint64, ncclMin, comm, ctx->stream);
    const ncclResult_t re = R.GroupEnd();
    ABZ_NCCL_CHECK(rc);
    ABZ_NCCL_CHECK(re);
    return 0;
  }
  /* host transport: the same exchange as ONE wait for the stream -- every piece staged, the callbacks, everything copied back */
  abz_host_piece p[4];
  int np = 0;
  size_t off = 0;
  auto add = [&](void* dev, size_t piece) { p[np++] = abz_host_piece{(char*)dev, piece, off}; off += abz_align(piece * G, 64); };
  add(ntheta, nl * (size_t)ld * 8); add(nlogpi, nl * 8); add(ndelta, nl * 8);
  if (nstamp) add(nstamp, nl * 8);
  const size_t off_part = off, off_scr = off + 64;
  if (int rc = abz_stage_reserve(ctx, off_scr + 8 * 3 * G + 64)) return rc;
  char* st = (char*)ctx->hc_stage;
  ABZ_HIP_CHECK(hipMemcpyAsync(st + off_part, part, 5 * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (int rc = abz_host_allgather_many(ctx, p, np)) return rc;       /* (waits for the stream: the five words are here too) */
  if (int rc = abz_host_allreduce_words(ctx, st + off_part, 2, 2, 0, st + off_scr)) return rc;
  if (int rc = abz_host_allreduce_words(ctx, st + off_part + 16, 3, 2, 1, st + off_scr)) return rc;
  ABZ_HIP_CHECK(hipMemcpyAsync(part, st + off_part, 5 * 8, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

extern "C" {

int abcdez_comm_unique_id(void* id_out, size_t bytes) {
  ABZ_REQUIRE(id_out && bytes >= (size_t)NCCL_UNIQUE_ID_BYTES, "comm_unique_id: needs a buffer of 128 bytes");
  if (int rc = abz_rccl_load()) return rc;
  ncclUniqueId id;
  ABZ_NCCL_CHECK(R.GetUniqueId(&id));
  memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

int abcdez_comm_init(abcdez_ctx* ctx, const void* unique_id, size_t bytes, int rank, int world) {
  ABZ_REQUIRE(ctx && unique_id && bytes >= (size_t)NCCL_UNIQUE_ID_BYTES, "comm_init: null argument / id shorter than 128 bytes");
  ABZ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: need 0 <= rank < world");
  ABZ_REQUIRE(ctx->comm_kind == ABZ_COMM_NONE, "comm_init: the context already has a communicator (abcdez_comm_destroy first)");
  if (int rc = abz_rccl_load()) return rc;
  ABZ_HIP_CHECK(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  ABZ_NCCL_CHECK(R.CommInitRank(&comm, world, id, rank));
  ctx->comm = (void*)comm; ctx->comm_kind = ABZ_COMM_RCCL; ctx->comm_broken = false; ctx->comm_rank = rank; ctx->comm_world = world;
  return 0;
}

int abcdez_comm_init_host(abcdez_ctx* ctx, int rank, int world, abcdez_host_allgather_fn allgather, abcdez_host_allreduce_fn allreduce,
                          void* user) {
  ABZ_REQUIRE(ctx && allgather, "comm_init_host: null context / the all-gather callback is required");
  ABZ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init_host: need 0 <= rank < world");
  ABZ_REQUIRE(ctx->comm_kind == ABZ_COMM_NONE, "comm_init_host: the context already has a communicator (abcdez_comm_destroy first)");
  ctx->hc_allgather = allgather; ctx->hc_allreduce = allreduce; ctx->hc_user = user;
  ctx->comm = nullptr; ctx->comm_kind = ABZ_COMM_HOST; ctx->comm_broken = false; ctx->comm_rank = rank; ctx->comm_world = world;
  return 0;
}

int abcdez_comm_destroy(abcdez_ctx* ctx) {
  ABZ_REQUIRE(ctx, "comm_destroy: null context");
  if (ctx->comm_kind == ABZ_COMM_NONE) return 0;
  (void)hipStreamSynchronize(ctx->stream);
  ncclComm_t comm = (ncclComm_t)ctx->comm;
  const int kind = ctx->comm_kind;
  ctx->comm = nullptr; ctx->comm_kind = ABZ_COMM_NONE; ctx->comm_broken = false; ctx->comm_rank = 0; ctx->comm_world = 1;
  ctx->hc_allgather = nullptr; ctx->hc_allreduce = nullptr; ctx->hc_user = nullptr;
  if (ctx->hc_stage) { (void)hipHostFree(ctx->hc_stage); ctx->hc_stage = nullptr; ctx->hc_stage_bytes = 0; }
  if (kind == ABZ_COMM_RCCL && comm) ABZ_NCCL_CHECK(R.CommDestroy(comm));
  return 0;
}

int abcdez_comm_rank(abcdez_ctx* ctx, int32_t* rank, int32_t* world) {
  ABZ_REQUIRE(ctx && rank && world, "comm_rank: null argument");
  *rank = ctx->comm_rank; *world = ctx->comm_world;
  return ctx->comm_kind != ABZ_COMM_NONE ? 0 : 1;          /* 1: no communicator (a single-GPU context): rank 0 of 1 */
}

/* which transport: 0 none, 1 RCCL, 2 host callbacks */
int abcdez_comm_kind(abcdez_ctx* ctx, int32_t* kind) {
  ABZ_REQUIRE(ctx && kind, "comm_kind: null argument");
  *kind = ctx->comm_kind;
  return 0;
}

/* in place: rank r's piece is buf[r piece_bytes .. (r + 1) piece_bytes); afterwards every rank holds all `world` pieces.
 * RCCL: enqueued on the context's stream behind the kernels that wrote the piece; does not wait.  Host transport: waits for
 * the stream, calls the host, enqueues the copies back. */
int abcdez_comm_allgather(abcdez_ctx* ctx, void* buf, int64_t piece_bytes) {
  ABZ_REQUIRE(ctx && buf && piece_bytes >= 0, "comm_allgather: null argument");
  if (int rc = abz_comm_usable(ctx, "comm_allgather")) return rc;
  return abz_allgather_impl(ctx, buf, (size_t)piece_bytes);
}

/* in place over n elements of dtype (0: int64, 1: float64, 2: uint64) with op (0: sum, 1: min, 2: max) on the context's stream */
int abcdez_comm_allreduce(abcdez_ctx* ctx, void* buf, int64_t n, int dtype, int op) {
  ABZ_REQUIRE(ctx && buf && n >= 0, "comm_allreduce: null argument");
  ABZ_REQUIRE(dtype >= 0 && dtype <= 2 && op >= 0 && op <= 2, "comm_allreduce: dtype 0 (int64) / 1 (float64) / 2 (uint64), op 0 (sum) / 1 (min) / 2 (max)");
  if (int rc = abz_comm_usable(ctx, "comm_allreduce")) return rc;
  return abz_allreduce_impl(ctx, buf, (size_t)n, dtype, op);
}

/* The sweeps of one generation (smc:336-353) on a population sharded by position, in one call (RCCL: one host synchronisation).
 * chunk = positions per rank (a multiple of 64 with world * chunk >= n_alive); this rank sweeps [rank chunk, (rank + 1) chunk)
 * clipped to n_alive.  flags and delta need room for world * chunk entries.  A COLLECTIVE call. */
int abcdez_smc_sweeps_sharded(abcdez_ctx* ctx, uint32_t* bits_a, uint32_t* bits_b, int64_t n_alive, int64_t chunk, double* slot0,
                              double* slot1, double* logpi, double* delta, uint8_t* flags, double eps, double gamma0,
                              double gamma_sigma, uint32_t sweep0, int32_t k_max, double kmcmc_min, int64_t* nacc, int64_t* nsim,
                              int32_t* k_done) {
  ABZ_REQUIRE(ctx && bits_a && bits_b && slot0 && slot1 && logpi && delta && flags && nacc && nsim && k_done, "smc_sweeps_sharded: null argument");
  if (int rc = abz_comm_usable(ctx, "smc_sweeps_sharded")) return rc;
  ABZ_REQUIRE(1 <= k_max && k_max <= ABZ_GROUP_MAX, "smc_sweeps_sharded: 1 <= k_max <= 16 sweeps per call");
  const int64_t G = ctx->comm_world, r = ctx->comm_rank;
  ABZ_REQUIRE(chunk > 0 && chunk % 64 == 0 && G * chunk >= n_alive, "smc_sweeps_sharded: chunk must be a multiple of 64 with world * chunk >= n_alive");
  /* everything below this line that can fail on this rank alone happens BEFORE the first collective: the arguments the sweep and the
   * replay check (the same on every rank), the hand-over list of a two-launch sweep, the staging block of the host transport */
  ABZ_REQUIRE(n_alive >= 3 && n_alive <= ABZ_MAX_N && slot0 != slot1 && bits_a != bits_b && kmcmc_min >= 0.0,
              "smc_sweeps_sharded: needs at least 3 alive particles, two different slots / bit arrays and Kmcmc_min >= 0");
  const int64_t r_lo = r * chunk < n_alive ? r * chunk : n_alive;
  const int64_t r_hi = r_lo + chunk < n_alive ? r_lo + chunk : n_alive;
  if (abz_sweep_in_two_launches(ctx) && r_hi > r_lo)
    if (int rc = abz_lv_hand_reserve(ctx, (size_t)(r_hi - r_lo))) return rc;
  if (ctx->comm_kind == ABZ_COMM_HOST)
    if (int rc = abz_stage_reserve(ctx, (size_t)chunk * 8 * (size_t)G)) return rc;
  int rc = abcdez_smc_group_begin(ctx, n_alive, kmcmc_min);
  if (rc) return rc;
  for (int k = 0; k < k_max && rc == 0; ++k) {
    uint32_t* in = (k & 1) ? bits_b : bits_a;
    uint32_t* out = (k & 1) ? bits_a : bits_b;
    rc = abcdez_smc_swarm_packed(ctx, in, out, n_alive, r_lo, r_hi, slot0, slot1, logpi, delta, flags, eps, gamma0, gamma_sigma,
                                 sweep0 + (uint32_t)k, nullptr, nullptr);
    /* every rank executes the collective whether or not the sweep ran (the test of smc:352 is evaluated on the device) */
    if (rc == 0) rc = abz_allgather_impl(ctx, flags, (size_t)chunk);
    if (rc == 0) rc = abcdez_smc_group_replay(ctx, in, out, r_lo, r_hi, slot0, slot1, logpi, flags, gamma0, gamma_sigma, sweep0 + (uint32_t)k);
  }
  if (rc == 0) rc = abcdez_smc_group_publish(ctx);
  /* the owners' distances are final: their exchange travels behind the read-back, while the host applies its rules */
  if (rc == 0) rc = abz_allgather_impl(ctx, delta, (size_t)chunk * 8);
  if (rc) {
    abz_comm_abort_after_failure(ctx);           /* the peers are in a collective this rank will not reach */
    (void)abcdez_smc_group_abort(ctx);
    return rc;
  }
  return abcdez_smc_group_end(ctx, nacc, nsim, k_done);
}

} /* extern "C" */
