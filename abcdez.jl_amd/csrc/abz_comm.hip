/*
 * abz_comm.hip -- the multi-GPU exchange steps of the population loop behind the C ABI: RCCL collectives over xGMI, issued on
 * the context's OWN stream between the kernels they depend on (no second stream, no event hand-over, no host framework).
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
 *   abcdez_comm_allgather / _allreduce -- the raw in-place collectives on the context's stream (initial population, abcdemc)
 *   abcdez_smc_sweeps_sharded -- the Kmcmc sweeps of one generation of abcdesmc! on a sharded population in ONE call:
 *                             own chunk sweep -> flag all-gather -> replay (+ the device-side test of smc:352), k_max times,
 *                             read-back, distance all-gather
 */
#include <rccl/rccl.h>

#include <string.h>

#include "abz_ctx.h"
#include "../../include/abcdez_hip.h"

#define ABZ_REQUIRE(cond, msg) \
  do { if (!(cond)) { abz_set_error(msg); return -1; } } while (0)
#define ABZ_NCCL_CHECK(expr)                                                             \
  do {                                                                                   \
    ncclResult_t _r = (expr);                                                            \
    if (_r != ncclSuccess) {                                                             \
      abz_set_error(std::string(#expr) + ": " + ncclGetErrorString(_r));                 \
      return -4;                                                                         \
    }                                                                                    \
  } while (0)

static_assert(NCCL_UNIQUE_ID_BYTES == ABCDEZ_COMM_ID_BYTES, "abcdez_hip.h promises the hosts a 128-byte id");

int abz_comm_mc_exchange(abcdez_ctx* ctx, double* ntheta, double* nlogpi, double* ndelta, uint64_t* nstamp, int64_t n_local, int ld) {
  ABZ_REQUIRE(ctx->comm, "mc_generation_sharded_async: no communicator (abcdez_comm_init)");
  ncclComm_t comm = (ncclComm_t)ctx->comm;
  const size_t r = (size_t)ctx->comm_rank, nl = (size_t)n_local;
  unsigned long long* part = ctx->d_scal + ABZ_S_MC_PART;
  ABZ_NCCL_CHECK(ncclGroupStart());
  ncclResult_t rc = ncclAllGather(ntheta + r * nl * (size_t)ld, ntheta, nl * (size_t)ld, ncclFloat64, comm, ctx->stream);
  if (rc == ncclSuccess) rc = ncclAllGather(nlogpi + r * nl, nlogpi, nl, ncclFloat64, comm, ctx->stream);
  if (rc == ncclSuccess) rc = ncclAllGather(ndelta + r * nl, ndelta, nl, ncclFloat64, comm, ctx->stream);
  if (rc == ncclSuccess && nstamp) rc = ncclAllGather(nstamp + r * nl, nstamp, nl, ncclUint64, comm, ctx->stream);
  if (rc == ncclSuccess) rc = ncclAllReduce(part, part, 2, ncclUint64, ncclSum, comm, ctx->stream);
  if (rc == ncclSuccess) rc = ncclAllReduce(part + 2, part + 2, 3, ncclUint64, ncclMin, comm, ctx->stream);
  const ncclResult_t re = ncclGroupEnd();
  ABZ_NCCL_CHECK(rc);
  ABZ_NCCL_CHECK(re);
  return 0;
}

extern "C" {

int abcdez_comm_unique_id(void* id_out, size_t bytes) {
  ABZ_REQUIRE(id_out && bytes >= (size_t)NCCL_UNIQUE_ID_BYTES, "comm_unique_id: needs a buffer of 128 bytes");
  ncclUniqueId id;
  ABZ_NCCL_CHECK(ncclGetUniqueId(&id));
  memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

int abcdez_comm_init(abcdez_ctx* ctx, const void* unique_id, size_t bytes, int rank, int world) {
  ABZ_REQUIRE(ctx && unique_id && bytes >= (size_t)NCCL_UNIQUE_ID_BYTES, "comm_init: null argument / id shorter than 128 bytes");
  ABZ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: need 0 <= rank < world");
  ABZ_REQUIRE(ctx->comm == nullptr, "comm_init: the context already has a communicator (abcdez_comm_destroy first)");
  ABZ_HIP_CHECK(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  ABZ_NCCL_CHECK(ncclCommInitRank(&comm, world, id, rank));
  ctx->comm = (void*)comm; ctx->comm_rank = rank; ctx->comm_world = world;
  return 0;
}

int abcdez_comm_destroy(abcdez_ctx* ctx) {
  ABZ_REQUIRE(ctx, "comm_destroy: null context");
  if (!ctx->comm) return 0;
  (void)hipStreamSynchronize(ctx->stream);
  ncclComm_t comm = (ncclComm_t)ctx->comm;
  ctx->comm = nullptr; ctx->comm_rank = 0; ctx->comm_world = 1;
  ABZ_NCCL_CHECK(ncclCommDestroy(comm));
  return 0;
}

int abcdez_comm_rank(abcdez_ctx* ctx, int32_t* rank, int32_t* world) {
  ABZ_REQUIRE(ctx && rank && world, "comm_rank: null argument");
  *rank = ctx->comm_rank; *world = ctx->comm_world;
  return ctx->comm ? 0 : 1;          /* 1: no communicator (a single-GPU context): rank 0 of 1 */
}

/* in place: rank r's piece is buf[r piece_bytes .. (r + 1) piece_bytes); afterwards every rank holds all `world` pieces.
 * Enqueued on the context's stream behind the kernels that wrote the piece; does not wait. */
int abcdez_comm_allgather(abcdez_ctx* ctx, void* buf, int64_t piece_bytes) {
  ABZ_REQUIRE(ctx && buf && piece_bytes >= 0, "comm_allgather: null argument");
  ABZ_REQUIRE(ctx->comm, "comm_allgather: no communicator (abcdez_comm_init)");
  if (piece_bytes == 0) return 0;
  char* b = (char*)buf;
  ABZ_NCCL_CHECK(ncclAllGather(b + (size_t)ctx->comm_rank * (size_t)piece_bytes, b, (size_t)piece_bytes, ncclUint8,
                               (ncclComm_t)ctx->comm, ctx->stream));
  return 0;
}

/* in place over n elements of dtype (0: int64, 1: float64) with op (0: sum, 1: min, 2: max) on the context's stream */
int abcdez_comm_allreduce(abcdez_ctx* ctx, void* buf, int64_t n, int dtype, int op) {
  ABZ_REQUIRE(ctx && buf && n >= 0, "comm_allreduce: null argument");
  ABZ_REQUIRE(ctx->comm, "comm_allreduce: no communicator (abcdez_comm_init)");
  ABZ_REQUIRE((dtype == 0 || dtype == 1) && op >= 0 && op <= 2, "comm_allreduce: dtype 0 (int64) / 1 (float64), op 0 (sum) / 1 (min) / 2 (max)");
  if (n == 0) return 0;
  const ncclRedOp_t ops[3] = {ncclSum, ncclMin, ncclMax};
  ABZ_NCCL_CHECK(ncclAllReduce(buf, buf, (size_t)n, dtype == 0 ? ncclInt64 : ncclFloat64, ops[op], (ncclComm_t)ctx->comm, ctx->stream));
  return 0;
}

/* The sweeps of one generation (smc:336-353) on a population sharded by position, in one call and one host synchronisation.
 * chunk = positions per rank (a multiple of 64 with world * chunk >= n_alive); this rank sweeps [rank chunk, (rank + 1) chunk)
 * clipped to n_alive.  flags and delta need room for world * chunk entries. */
int abcdez_smc_sweeps_sharded(abcdez_ctx* ctx, uint32_t* bits_a, uint32_t* bits_b, int64_t n_alive, int64_t chunk, double* slot0,
                              double* slot1, double* logpi, double* delta, uint8_t* flags, double eps, double gamma0,
                              double gamma_sigma, uint32_t sweep0, int32_t k_max, double kmcmc_min, int64_t* nacc, int64_t* nsim,
                              int32_t* k_done) {
  ABZ_REQUIRE(ctx && bits_a && bits_b && slot0 && slot1 && logpi && delta && flags && nacc && nsim && k_done, "smc_sweeps_sharded: null argument");
  ABZ_REQUIRE(ctx->comm, "smc_sweeps_sharded: no communicator (abcdez_comm_init)");
  ABZ_REQUIRE(1 <= k_max && k_max <= ABZ_GROUP_MAX, "smc_sweeps_sharded: 1 <= k_max <= 16 sweeps per call");
  const int64_t G = ctx->comm_world, r = ctx->comm_rank;
  ABZ_REQUIRE(chunk > 0 && chunk % 64 == 0 && G * chunk >= n_alive, "smc_sweeps_sharded: chunk must be a multiple of 64 with world * chunk >= n_alive");
  const int64_t r_lo = r * chunk < n_alive ? r * chunk : n_alive;
  const int64_t r_hi = r_lo + chunk < n_alive ? r_lo + chunk : n_alive;
  int rc = abcdez_smc_group_begin(ctx, n_alive, kmcmc_min);
  if (rc) return rc;
  for (int k = 0; k < k_max && rc == 0; ++k) {
    uint32_t* in = (k & 1) ? bits_b : bits_a;
    uint32_t* out = (k & 1) ? bits_a : bits_b;
    rc = abcdez_smc_swarm_packed(ctx, in, out, n_alive, r_lo, r_hi, slot0, slot1, logpi, delta, flags, eps, gamma0, gamma_sigma,
                                 sweep0 + (uint32_t)k, nullptr, nullptr);
    /* every rank executes the collective whether or not the sweep ran (the test of smc:352 is evaluated on the device) */
    if (rc == 0) rc = abcdez_comm_allgather(ctx, flags, chunk);
    if (rc == 0) rc = abcdez_smc_group_replay(ctx, in, out, r_lo, r_hi, slot0, slot1, logpi, flags, gamma0, gamma_sigma, sweep0 + (uint32_t)k);
  }
  if (rc == 0) rc = abcdez_smc_group_publish(ctx);
  /* the owners' distances are final: their exchange travels behind the read-back, while the host applies its rules */
  if (rc == 0) rc = abcdez_comm_allgather(ctx, delta, chunk * 8);
  if (rc) { (void)abcdez_smc_group_abort(ctx); return rc; }
  return abcdez_smc_group_end(ctx, nacc, nsim, k_done);
}

} /* extern "C" */
