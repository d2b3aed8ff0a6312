/*
 * abcdez_hip.h -- C ABI of libabcdez_hip.so: the MI355X (gfx950) engine behind
 * ABCdeZ.jl's per-generation population loop.
 *
 * The reference has no FFI; its internal seam is a set of Julia functions that
 * mutate caller-owned arrays.  Each entry point below
 * replaces one of them and keeps its argument meaning.  Conventions:
 *   - every array argument is a DEVICE pointer to a caller-owned buffer (allocate
 *     with the host's own GPU array package, with torch, or with abcdez_dev_alloc);
 *   - arrays are FULL population arrays of N elements (theta / row slots: N rows of `ld`
 *     doubles, row-major); functions that shard take a global range of particles / positions;
 *   - scalar results come back through HOST pointers; such functions synchronise
 *     the context's stream before returning, the others only enqueue;
 *   - return value: 0 = ok, negative = error (text via abcdez_last_error());
 *     no exceptions cross the boundary;
 *   - one context per host thread; a context is not re-entrant.
 *
 * Model descriptor (prior, dist!, ABCk, rng seed): struct abz_model in
 * abcdez_spec.h (shipped next to this header as include/abcdez_spec.h).
 */
#ifndef ABCDEZ_HIP_H
#define ABCDEZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "abcdez_spec.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ABCDEZ_API __attribute__((visibility("default")))

typedef struct abcdez_ctx abcdez_ctx;

ABCDEZ_API int abcdez_version(void);
/* Rounds of the Philox4x32 stream this library was built with (ABZ_PHILOX_ROUNDS, abcdez_spec.h: 10).  Every random number of a
 * run depends on it, so hosts store it in their checkpoints next to abcdez_version() and refuse to resume across a change. */
ABCDEZ_API int abcdez_rng_rounds(void);
/* Layout of the structs that cross the boundary: fills out[0 .. n) with { sizeof(abz_prior_dim), offsetof of its 7 fields
 * in declaration order, sizeof(abz_model), offsetof of its 14 fields in declaration order } and returns how many
 * values there are (23).  Hosts that mirror the structs by hand assert this (julia/ABCdeZHIP.jl, tests/test_host_api.py). */
ABCDEZ_API int abcdez_abi_layout(int32_t* out, int n);
ABCDEZ_API const char* abcdez_last_error(void);

/* Context = (prior, dist!, varexternal, rng, ex) of the reference signatures
 * (src/abcdez_smc.jl:106-109, src/abcdez_mc.jl:5-6, src/abcdez_init.jl:2).
 * `model->data` is a HOST pointer here; it is copied to the device.             */
ABCDEZ_API int abcdez_ctx_create(const abz_model* model, int device, abcdez_ctx** out);
/* The same with a user-supplied simulator (model->sim_id == ABZ_SIM_USER): `user_source` is HIP source text defining ONE of
 *   d <= 16 -- the whole row in one thread:
 *     __device__ double abz_user_dist(const double* theta, int d, const double* data, int n_data,
 *                                     const double* sim_p, abz_user_rng& rng);
 *   17 <= d <= 256 -- the row spread over the lanes of a wavefront (as the built-in d-dimensional Normal simulator: 8 components on each
 *   of 4 or 8 lanes up to 64 parameters, 16 or 32 components on each of 8 lanes beyond), every lane
 *   of the group calling
 *     __device__ double abz_user_dist_lanes(const double* theta, const abz_user_lanes& g, int d, const double* data, int n_data,
 *                                           const double* sim_p, abz_user_rng& rng);
 *   with ITS ABZ_USER_C components (g.comp(q) = index in the row; g.sum(v) = the canonical tree sum over the group; draws addressed
 *   by component: rng.normal_pair_at(k, z0, z1), rng.uniform_at(k)) and returning the distance on every lane;
 *   the STAGED form (d <= 16) -- #define ABZ_USER_ROUNDS / ABZ_USER_STATE and
 *     __device__ double abz_user_round(const double* theta, int d, const double* data, int n_data, const double* sim_p,
 *                                      abz_user_rng& rng, int round, double* state);
 *   returning after every step a lower bound of the final distance that never decreases: proposals whose bound has passed eps leave
 *   the simulation early (csrc/abz_user_rounds.h), results bit for bit those of running every round.
 *   With 3 to 16 parameters (rows of 4, 8 or 16 doubles) either one-thread form sweeps in TWO launches: the simulator runs in a launch of
 *   its own over the dense list of the proposals the prior ratio has not already rejected (that is where the staged form's early exit
 *   acts); ABZ_USER_ONE_KERNEL=1 in the environment keeps it inside the sweep kernel.
 * (theta push_p-cast; rng.uniform() / rng.normal() / rng.normal_pair(z0, z1) / rng.bits() hand out the
 * particle's Philox stream).  It is compiled with hiprtc for the device's architecture together with the
 * library's own kernel bodies; a compile error comes back as a negative status with the compiler log in
 * abcdez_last_error().  This is the device counterpart of the reference's dist!(theta, ve) closure
 * (src/abcdez_smc.jl:137,166-173, src/abcdez_mc.jl:45, src/abcdez_init.jl:10,17), for any length(prior) the library takes.  */
ABCDEZ_API int abcdez_ctx_create_user(const abz_model* model, const char* user_source, int device, abcdez_ctx** out);
/* Test hook, no device needed: the translation unit abcdez_ctx_create_user would compile for this model and source, and the -D options
 * that go with it (one per line) -- so that a host without a GPU can check a simulator's source with hipcc.  Returns the text's length. */
ABCDEZ_API int abcdez_user_translation_unit(const abz_model* model, const char* user_source, char* tu_out, size_t tu_cap,
                                            char* opts_out, size_t opts_cap);
ABCDEZ_API int abcdez_ctx_destroy(abcdez_ctx* ctx);
/* Optional: size the context's internal workspace for populations of up to N particles now, so that no later call
 * (the first resampling, the first quantile) has to grow it -- growing synchronises the stream and reallocates. */
ABCDEZ_API int abcdez_ctx_reserve(abcdez_ctx* ctx, int64_t N);
ABCDEZ_API int abcdez_ctx_set_stream(abcdez_ctx* ctx, void* hip_stream);
/* lanes per particle (power of two dividing ld, <= 16; 0 = default) -- tuning knob */
ABCDEZ_API int abcdez_ctx_set_lanes(abcdez_ctx* ctx, int lanes);
/* Uniform weights.  In every run of the reference's drivers with an INDICATOR kernel the alive particles' weights are uniform:
 * 1/N at smc:266-270 and after every resampling (smc:102), 1/n_alive after every reweight (smc:310: equal products over their
 * sum).  ws[i] is 1 or 0 (types.jl:26-50), so wnorm = n_new / n_old, Wns = 1 / n_new, 1 / sum(Wns.^2) = n_new in exact arithmetic.
 * A host that has written uniform weights says so with set_uniform_weights(ctx, 1); abcdez_smc_prologue_packed then takes exactly
 * those values (IEEE divisions of integers) instead of floating sums of n_new equal terms, in two launches instead of four, and
 * the library keeps the flag from there (the fast path keeps it, abcdez_smc_resample_gather_packed sets it, a general reweight --
 * another kernel family, or abcdez_smc_reweight -- clears it).  Without the call the general path runs: the reference's
 * statements on whatever weights the arrays hold.  get_uniform_weights: the flag (a checkpoint must carry it for a resumed run
 * to repeat the uninterrupted one bit for bit) and how many prologues took the fast path. */
ABCDEZ_API int abcdez_ctx_set_uniform_weights(abcdez_ctx* ctx, int on);
ABCDEZ_API int abcdez_ctx_get_uniform_weights(abcdez_ctx* ctx, int32_t* on, int64_t* fast_prologues);
/* abcdemc generations enqueued by abcdez_mc_generation_async (up to 15 dependent launches, no host decision in between,
 * src/abcdez_mc.jl:134-161) CAN be captured once per launch shape and replayed as HIP graphs (on = 1, or ABZ_GRAPHS=1 in the
 * environment; needs a stream other than the legacy default stream).  Off by default: on ROCm 7.2 / MI355X the replay is slower
 * than the stream launches it replaces (0.140 against 0.125 ms per generation at 2^20 particles; profiles/r04_mc1d_graph_replay_ab.json).
 * Results do not depend on it.  abcdez_graph_stats: generations replayed / graphs captured / generations enqueued launch by launch. */
ABCDEZ_API int abcdez_ctx_set_graphs(abcdez_ctx* ctx, int on);
ABCDEZ_API int abcdez_graph_stats(abcdez_ctx* ctx, int64_t* replays, int64_t* captures, int64_t* direct);
ABCDEZ_API int abcdez_ctx_get_layout(abcdez_ctx* ctx, int32_t* ld, int32_t* lanes, int32_t* comps_per_lane);
ABCDEZ_API int abcdez_sync(abcdez_ctx* ctx);
/* HIP-event timing of the sweep kernel on the context's stream (measurement only): accumulated kernel milliseconds,
 * launches and particle-updates since set_timing(on).  on = mode + 256 * stride.  mode 1: every sweep launch is bracketed by an
 * event pair; mode 2: of the sweeps of one abcdez_smc_sweeps_packed call only one, the 1st, 2nd, 3rd ... in turn;
 * mode 3: ONE pair around all the sweeps of an abcdez_smc_sweeps_packed call -- it counts as the k_done launches that did work and its
 * milliseconds include the one-block checks between them (an upper bound of the sweeps' duration; the launches run back to back as they
 * do un-instrumented, where a pair of its own makes a launch start on a drained queue).  stride > 1: of the launches the mode
 * selects only every stride-th -- a pair costs ~9 us of queue time (tools/launch_floor.hip), 7 % of an abcdemc generation. */
ABCDEZ_API int abcdez_ctx_set_timing(abcdez_ctx* ctx, int on);
ABCDEZ_API int abcdez_ctx_get_timing(abcdez_ctx* ctx, double* swarm_ms, int64_t* launches, int64_t* units);

/* Device memory helpers for hosts without a GPU array package of their own. */
ABCDEZ_API int abcdez_dev_alloc(size_t bytes, void** out);
ABCDEZ_API int abcdez_dev_free(void* ptr);
ABCDEZ_API int abcdez_memcpy_h2d(abcdez_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
ABCDEZ_API int abcdez_memcpy_d2h(abcdez_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* The one bulk transfer of the path is the result download at the end of a run (P, Wns, C: src/abcdez_smc.jl:382-393, src/abcdez_mc.jl:166-171;
 * 1.2 GB at BASELINE configs[2]).  Into pageable memory the runtime stages it through bounce buffers at ~12 GB/s; into PINNED host memory
 * the copy engine writes at the link's rate.  host_alloc / host_free: page-locked host memory (hipHostMalloc); memcpy_d2h_async: enqueue
 * the copy on the context's stream and return -- several arrays travel back to back, abcdez_sync waits for all of them. */
ABCDEZ_API int abcdez_host_alloc(size_t bytes, void** out);
ABCDEZ_API int abcdez_host_free(void* ptr);
ABCDEZ_API int abcdez_memcpy_d2h_async(abcdez_ctx* ctx, void* dst_host_pinned, const void* src_dev, size_t bytes);

/* S1  abcde_init!(prior, dist!, varexternal, thetas, logpi, Ds, nparticles, rng, ex, blobs)
 *     src/abcdez_init.jl:2-22, including the prior draws and log-priors of
 *     src/abcdez_smc.jl:242-243 / src/abcdez_mc.jl:117-118.  Fills rows [i0, i0+n).  */
ABCDEZ_API int abcdez_init(abcdez_ctx* ctx, double* theta, double* logpi, double* delta, int64_t i0, int64_t n);

/* Blobs -- the second return value of dist!(theta, ve) -> (d, blob) (src/abcdez_smc.jl:137,148, src/abcdez_mc.jl:45,
 * docs/src/index.md:298-324), here: the simulated data behind a particle's current distance (model.n_blob doubles).
 * The population does not store them.  It carries an 8-byte STAMP per particle next to the distance -- (origin
 * particle, RNG epoch, init-or-sweep stream) of the simulator call that produced it (abz_stamp, abcdez_spec.h) --
 * written at init and on every accept, copied / gathered exactly like Ds and blobs in the reference (smc:99,
 * 337-340).  abcdez_ctx_set_stamps names the stamp arrays of the current and the next generation (u64[N] each;
 * the (logpi, nlogpi)-style entry points use them alongside, the packed sweeps update stamp_cur in place; NULL,
 * NULL = off).  abcdez_blob_eval re-runs that one simulator call for each of N particles on dense current rows
 * theta[N][ld] and writes blob[N][width] (abcdez_blob_width: ld for the MVN simulator, n_blob otherwise) and the
 * distance of the re-run, which must equal the stored distance bit for bit.                                    */
ABCDEZ_API int abcdez_ctx_set_stamps(abcdez_ctx* ctx, uint64_t* stamp_cur, uint64_t* stamp_nxt);
ABCDEZ_API int abcdez_blob_width(abcdez_ctx* ctx, int32_t* width);
ABCDEZ_API int abcdez_blob_eval(abcdez_ctx* ctx, const double* theta, const uint64_t* stamp, int64_t N, double* blob,
                     double* delta_out);

/* ---- S2+S3  abcdesmc_swarm!(prior, dist!, varexternal, alive, thetas, logpi, Ds, nthetas, nlogpi, nDs, eps_k_new, gamma0,
 *        gamma_sigma, nparticles, nsims, naccs, rng, ex, nblobs)  src/abcdez_smc.jl:106-153, the identity. copies of
 *        :337-340, and S8 abcdesmc_resample! src/abcdez_smc.jl:85-104 -- on the PACKED population.
 * The alive list wsample(rng, 1:N, alive) scans for (smc:121,125) is replaced by an invariant: after every reweight that kills particles the population is PARTITIONED so that the alive particles are the
 * positions [0, n_alive): the k-th dead position below n_new swaps its whole state with the k-th alive position at or
 * above it (the reference's algorithm is symmetric under relabelling of the particles, src/abcdez_smc.jl:106-153, so
 * the law of every output is unchanged; everything -- random numbers, donor ranks, summation trees -- is keyed by
 * position).  "Alive rank r" then IS "position r": the sweeps need no alive list and no index look-up in front of the
 * donor rows.  Rows live in two slots per position (slot0[N][ld], slot1[N][ld]); one bit per position (bits, 32
 * positions per uint32 word, N/32 words rounded up) names the current slot.  An accepted proposal is written to the
 * position's other slot and its bit flips in bits_out; a rejected one writes nothing; log-prior and distance are
 * updated in place (the copies of smc:337-340 disappear).  The two bit arrays ping-pong with the sweeps and agree
 * wherever no sweep writes.
 *
 * smc_partition: alive[0 .. n_prev) holds the flags after the reweight, n_new = sum(alive) (the reweight's n_alive).
 *   Swaps rows / log-prior / distance / weight / flag / blob stamp; copies bits to bits_other.  Asynchronous.
 * smc_swarm_packed: S2+S3 for the positions [r_lo, r_hi) of the prefix [0, n_alive) (one GPU: 0, n_alive; sharded: a
 *   sub-range whose ends are multiples of 64 or n_alive).  flags (may be NULL): per position, bit 0 accepted, bit 1
 *   simulated.  nacc = nsim = NULL: no counters and no host synchronisation.
 * smc_replay_packed: what a replica does for the positions of the OTHER ranks after the flag exchange: rebuilds the
 *   accepted proposals theta_i + gamma (theta_a - theta_b) (smc:128) and their log-priors from its own rows and the
 *   positions' counter-based random numbers, completes bits_out, and returns sum(naccs), sum(nsims) over the WHOLE
 *   prefix from the flags.  One byte per particle and sweep crosses xGMI instead of 8 ld + 16.
 * smc_resample_gather_packed: S8; source = current row of inds[s], destination = the other slot of s; afterwards
 *   every bit is flipped in both bit arrays; nlogpi / ndelta receive the gathered log-priors / distances.
 * packed_gather: the current rows as one dense array out[N][ld] (results, checkpoints).
 * Rows of at most two doubles (ld <= 2) are kept DOUBLE-BUFFERED: a sweep (and its replay) writes every swept position's
 *   row to the other slot -- the accepted proposal or a copy -- and flips every swept bit; a resampling writes every
 *   position's new row to the slot the alive prefix is not in and gives all N bits that parity.  The alive prefix then
 *   always shares one parity and the sweep takes its donors' slot from the own position's bit (two random look-ups
 *   fewer per update, which is what bounds 8-byte rows).  Callers see no difference: the same arrays, the same calls;
 *   the invariant holds for every population made by abcdez_init + these entry points.                          */
ABCDEZ_API int abcdez_smc_partition(abcdez_ctx* ctx, uint8_t* alive, int64_t N, int64_t n_prev, int64_t n_new,
                                    const uint32_t* bits, uint32_t* bits_other, double* slot0, double* slot1,
                                    double* logpi, double* delta, double* wns);
/* One generation's prologue of the driver loop in ONE call and ONE host synchronisation: extrema(Ds) of the generation
 * that just ended (src/abcdez_smc.jl:364), eps = max(min(quantile(Ds[alive], alpha), eps_prev), eps_target) (:301),
 * abcdesmc_update_ws! + normalisation + alive flags + ESS (:305-311, :323), and -- unless ESS < ess_min, in which case
 * the driver resamples (:324-326) and *partitioned = 0 -- abcdez_smc_partition.  The host keeps the eps schedule and
 * the evidence accumulator: it passes eps_prev, eps_target and the previous kernel width eps_k_old in and gets eps,
 * wnorm (logZ += log(wnorm), :315), ESS and n_alive = sum(alive) back.  q = the raw quantile, (dmin, dmax) = the
 * extrema: may be NULL.  delta / wns / alive are the FULL arrays of N; the first n_prev positions are the alive prefix. */
ABCDEZ_API int abcdez_smc_prologue_packed(abcdez_ctx* ctx, double* delta, double* wns, uint8_t* alive, int64_t N,
                                          int64_t n_prev, double alpha, double eps_prev, double eps_target,
                                          double eps_k_old, double ess_min, const uint32_t* bits, uint32_t* bits_other,
                                          double* slot0, double* slot1, double* logpi, double* eps, double* q,
                                          double* wnorm, double* ess, int64_t* n_alive, int32_t* partitioned,
                                          double* dmin, double* dmax);
ABCDEZ_API int abcdez_smc_swarm_packed(abcdez_ctx* ctx, const uint32_t* bits, uint32_t* bits_out, int64_t n_alive,
                                       int64_t r_lo, int64_t r_hi, double* slot0, double* slot1, double* logpi,
                                       double* delta, uint8_t* flags, double eps, double gamma0, double gamma_sigma,
                                       uint32_t sweep, int64_t* nacc, int64_t* nsim);
/* The sweeps of ONE generation -- `for i in 1:Kmcmc ... (sum(naccs) / n_alive >= Kmcmc_min) && break`
 * (src/abcdez_smc.jl:336-353) -- in one call and one host synchronisation: up to k_max (<= 16) sweeps over the whole
 * prefix, numbered sweep0, sweep0 + 1, ...; sweep i reads bits_a and writes bits_b for odd i (1-based), the other way
 * round for even i.  The early-exit test of :352 is evaluated on the device after every sweep (same IEEE division as
 * the host's) and the sweeps enqueued behind a test that held do nothing.  Returns the per-sweep counters
 * nacc[0 .. k_max), nsim[0 .. k_max) (zero for sweeps that did not run) and *k_done = Ki of :352; the current bit array
 * afterwards is bits_b when *k_done is odd, bits_a when it is even.                                                */
/* Arms the NEXT abcdez_smc_sweeps_packed: behind its last sweep it also enqueues the first third of the next generation's
 * prologue -- extrema(Ds) (smc:364), the rank select and eps = max(min(quantile, eps), eps_target) (smc:301) with eps_prev = the
 * sweeps' eps and n_prev = their n_alive -- which only reads the distances and the alive flags, so the device works on it while
 * the host reads the sweeps' counters and applies its stop rules (smc:375-376).  An abcdez_smc_prologue_packed that follows with
 * the same (delta, alive, N, n_prev, alpha, eps_prev, eps_target) starts at the reweight; every other call discards the work.
 * Results are the same either way. */
ABCDEZ_API int abcdez_smc_select_ahead(abcdez_ctx* ctx, const double* delta, const uint8_t* alive, int64_t N, double alpha,
                                       double eps_target);
/* "The population was written behind the library's back."  Forgets a select armed or enqueued ahead AND ends the chain of
 * asynchronous abcdemc generations (the proved bound of the tail length that lets a rank pass launch a single sort was made
 * for the old distances).  Every library call that changes the distances or the flags does so itself; a host that writes those
 * arrays by other means (a copy, a resumed checkpoint), changes the stream, or ends a run calls this.  (Should a host forget:
 * a rank pass whose bound no longer holds is detected on the device and abcdez_mc_generation_wait returns an error.) */
ABCDEZ_API int abcdez_smc_select_discard(abcdez_ctx* ctx);
/* Diagnostics: prologues of this context that found their select enqueued ahead / that ran it themselves (the first generation,
 * the one after a resample, after a discard).  A steady-state abcdesmc loop reuses one select per generation. */
ABCDEZ_API int abcdez_smc_select_stats(abcdez_ctx* ctx, int64_t* reused, int64_t* inline_runs);
/* Diagnostics: rank passes (abcdez_mc_generation_async / abcdez_mc_rank_prepare) that launched both sorts of the tail, only the
 * one-workgroup LDS sort, only the radix sort.  One alone is launched when the host holds a proved upper bound of the tail's length:
 * once a generation of a chain of asynchronous generations ran with eps_pop == eps_target, the particles that draw only become fewer. */
ABCDEZ_API int abcdez_mc_rank_stats(abcdez_ctx* ctx, int64_t* both, int64_t* small_only, int64_t* long_only);
/* ... and asynchronous generations issued with do_rank != 0 for which no rank pass was launched, because a redeemed generation
 * of the chain had left at least 1 / 16 of the particles at or below eps_target (they draw their better particles by rejection). */
ABCDEZ_API int abcdez_mc_draw_stats(abcdez_ctx* ctx, int64_t* by_rejection_no_rank_pass);
/* The rule itself (include/abcdez_spec.h, abz_mc_draws_by_rejection): 1 if a generation that reads n_above distances above
 * eps_target among N particles draws by rejection, 0 if by rank, -1 for arguments out of range.  A host that drives
 * abcdez_mc_rank_prepare / abcdez_mc_swarm itself (the sharded path) asks here. */
ABCDEZ_API int abcdez_mc_draws_by_rejection(int64_t n_above, int64_t N);
ABCDEZ_API int abcdez_smc_sweeps_packed(abcdez_ctx* ctx, uint32_t* bits_a, uint32_t* bits_b, int64_t n_alive,
                                        double* slot0, double* slot1, double* logpi, double* delta, double eps,
                                        double gamma0, double gamma_sigma, uint32_t sweep0, int32_t k_max,
                                        double kmcmc_min, int64_t* nacc, int64_t* nsim, int32_t* k_done);
/* ONE generation of the loop body of abcdesmc! (src/abcdez_smc.jl:301-353) in one call: abcdez_smc_prologue_packed, then -- when
 * ESS < ess_min, smc:323-326 -- abcdez_wsample_stratified + abcdez_smc_resample_gather_packed (+ abcdez_get_ess of the uniform weights,
 * what the drivers record), then abcdez_smc_sweeps_packed on the population that results; the two decisions between them are taken inside
 * the call.  (bits_cur, logpi_cur, delta_cur) name the current state; after a resampling (*resampled = 1) the OTHER (logpi, delta) pair is
 * the current one and the host swaps its names (the bitmaps keep theirs; k_done odd flips them as after abcdez_smc_sweeps_packed).
 * Outputs: eps, wnorm, ess, n_alive, partitioned, dmin / dmax as abcdez_smc_prologue_packed; n_swept = the alive count the sweeps ran on
 * (N after a resampling; no sweeps below 3); nacc[k], nsim[k], k_done as abcdez_smc_sweeps_packed.  An unsharded population only. */
ABCDEZ_API int abcdez_smc_generation_packed(abcdez_ctx* ctx, int64_t N, int64_t n_prev, uint32_t* bits_cur, uint32_t* bits_oth,
                                            double* slot0, double* slot1, double* logpi_cur, double* delta_cur, double* logpi_oth,
                                            double* delta_oth, double* wns, uint8_t* alive, uint32_t* inds, double alpha,
                                            double eps_prev, double eps_target, double eps_k_old, double ess_min, double gamma0,
                                            double gamma_sigma, uint32_t sweep0, uint32_t draw, int32_t k_max, double kmcmc_min,
                                            int32_t select_ahead, double* eps, double* wnorm, double* ess, int64_t* n_alive,
                                            int32_t* partitioned, int32_t* resampled, double* ess_resampled, int64_t* n_swept,
                                            int64_t* nacc, int64_t* nsim, int32_t* k_done, double* dmin, double* dmax);
ABCDEZ_API int abcdez_smc_replay_packed(abcdez_ctx* ctx, const uint32_t* bits, uint32_t* bits_out, int64_t n_alive,
                                        int64_t skip_lo, int64_t skip_hi, double* slot0, double* slot1, double* logpi,
                                        const uint8_t* flags, double gamma0, double gamma_sigma, uint32_t sweep,
                                        int64_t* nacc, int64_t* nsim);
/* The sweeps of one generation on a SHARDED population with ONE host synchronisation (smc:336-353; the unsharded counterpart is
 * abcdez_smc_sweeps_packed).  Between group_begin and group_end: per sweep abcdez_smc_swarm_packed on the own range with
 * nacc = NULL (sweep k > 0 returns at once when the test of smc:352 held), the host's all-gather of the flag bytes (every
 * rank executes it, stopped or not), abcdez_smc_group_replay (gated the same way; counts both flag bits over the whole prefix and
 * evaluates `sum(naccs) / n_alive >= Kmcmc_min` on the device).  group_publish enqueues the read-back so that the host can put
 * the distance exchange behind it; group_end waits and returns the per-sweep counters and the number of sweeps that ran.
 * Every replica sees the same flags, so all ranks stop after the same sweep without a collective. */
ABCDEZ_API int abcdez_smc_group_begin(abcdez_ctx* ctx, int64_t n_alive, double kmcmc_min);
ABCDEZ_API int abcdez_smc_group_replay(abcdez_ctx* ctx, const uint32_t* bits, uint32_t* bits_out, int64_t skip_lo, int64_t skip_hi,
                                       double* slot0, double* slot1, double* logpi, const uint8_t* flags, double gamma0,
                                       double gamma_sigma, uint32_t sweep);
ABCDEZ_API int abcdez_smc_group_publish(abcdez_ctx* ctx);
/* Abandon an open group after a failure between _begin and _end (a collective raised, a launch failed): waits for the stream,
 * drops the group's timing pairs and re-reads the counter baselines; a no-op when no group is open. */
ABCDEZ_API int abcdez_smc_group_abort(abcdez_ctx* ctx);
ABCDEZ_API int abcdez_smc_group_end(abcdez_ctx* ctx, int64_t* nacc, int64_t* nsim, int32_t* k_done);
/* ---- Multi-GPU: the exchange steps behind the ABI (csrc/abz_comm.hip).
 * The reference parallelises its population loops over the threads of one process (`@floop ex for i in 1:nparticles`,
 * src/abcdez_smc.jl:110, src/abcdez_mc.jl:7, src/abcdez_init.jl:6; `parallel=true`, smc:237); here one process per GPU owns a
 * contiguous range of positions and what the threads share through memory is exchanged by collectives the LIBRARY issues, over one
 * of two transports:
 *   RCCL over xGMI on the context's own stream (comm_unique_id + comm_init).  librccl is opened lazily by the first of these calls:
 *     the library itself loads on hosts without RCCL.
 *   a HOST-SUPPLIED transport (comm_init_host): the library stages this rank's piece through page-locked host memory, waits for its
 *     stream, calls the host's all-gather on HOST memory and copies the other ranks' pieces back -- the same in-place semantics at
 *     the same points of the same entry points.  For hosts that exchange through MPI (`MPI.Allgather!`), for ranks sharing one GPU
 *     (RCCL refuses that), for tests.
 *
 *   comm_unique_id: rank 0 obtains the 128-byte rendezvous id; the host carries it to the other ranks (MPI.bcast, a file, torch's store).
 *   comm_init:      every rank, once per context: joins the RCCL communicator of `world` ranks on the context's device.
 *   comm_init_host: every rank, once per context: `allgather(user, buf, piece_bytes)` works IN PLACE on host memory -- on entry this
 *                   rank's piece lies at buf + rank * piece_bytes, on return buf holds all `world` pieces in rank order -- and returns
 *                   0 or an error code; `allreduce(user, buf, n, dtype, op)` reduces n 8-byte elements in place (dtype / op as in
 *                   comm_allreduce) and may be NULL: the library then all-gathers the words and reduces them in rank order itself
 *                   (the same bits on every rank).  The callbacks are called on the calling thread, inside the library call that
 *                   needs the exchange, and must be collective over the same `world` processes.
 *   comm_rank:      this context's (rank, world); returns 1 (and 0 of 1) when the context has no communicator.
 *   comm_kind:      0 = none, 1 = RCCL, 2 = host transport.
 *   comm_allgather: in place over `world` pieces of piece_bytes each, rank r's piece at buf + r * piece_bytes; enqueued on the
 *                   context's stream (ordered with the kernels before and after it; RCCL: does not wait).  Used for the rows /
 *                   log-priors / distances of the initial population (init.jl:2-22) and of abcdemc's sweeps (mc:140-155).
 *   comm_allreduce: in place over n elements; dtype 0 = int64, 1 = float64, 2 = uint64; op 0 = sum, 1 = min, 2 = max.
 *   smc_sweeps_sharded: the sweeps of ONE generation (smc:336-353) on a population sharded by position -- the counterpart of
 *                   abcdez_smc_sweeps_packed for world > 1, one call per generation (RCCL: one host synchronisation): k_max times
 *                   { sweep of the own chunk [rank chunk, (rank + 1) chunk) of the prefix -> all-gather of the flag bytes (1 B per
 *                   position) -> replay of the other ranks' accepted proposals + the device-side test of smc:352 }, read-back, and the
 *                   all-gather of the distance chunks behind it (8 B per position, once per generation).  chunk: a multiple of 64
 *                   with world * chunk >= n_alive; `flags` (u8) and `delta` need room for world * chunk entries.  Counters as in
 *                   abcdez_smc_sweeps_packed; every replica counts the same flags, so all ranks return the same k_done.
 * abcdez_smc_sweeps_sharded and abcdez_mc_generation_sharded_async are COLLECTIVE calls: every rank makes them with the same
 * arguments.  What can fail on one rank alone is done before the call's first collective; a failure after it aborts the
 * communicator (ncclCommAbort; a host transport is marked broken) so that the peers fail instead of waiting, and every later
 * collective of the context returns an error until abcdez_comm_destroy + a new initialisation. */
#define ABCDEZ_COMM_ID_BYTES 128
typedef int (*abcdez_host_allgather_fn)(void* user, void* buf, int64_t piece_bytes);
typedef int (*abcdez_host_allreduce_fn)(void* user, void* buf, int64_t n, int32_t dtype, int32_t op);
ABCDEZ_API int abcdez_comm_unique_id(void* id_out, size_t bytes);
ABCDEZ_API int abcdez_comm_init(abcdez_ctx* ctx, const void* unique_id, size_t bytes, int rank, int world);
ABCDEZ_API int abcdez_comm_init_host(abcdez_ctx* ctx, int rank, int world, abcdez_host_allgather_fn allgather,
                                     abcdez_host_allreduce_fn allreduce, void* user);
ABCDEZ_API int abcdez_comm_destroy(abcdez_ctx* ctx);
ABCDEZ_API int abcdez_comm_rank(abcdez_ctx* ctx, int32_t* rank, int32_t* world);
ABCDEZ_API int abcdez_comm_kind(abcdez_ctx* ctx, int32_t* kind);
ABCDEZ_API int abcdez_comm_allgather(abcdez_ctx* ctx, void* buf, int64_t piece_bytes);
ABCDEZ_API int abcdez_comm_allreduce(abcdez_ctx* ctx, void* buf, int64_t n, int dtype, int op);
ABCDEZ_API int abcdez_smc_sweeps_sharded(abcdez_ctx* ctx, uint32_t* bits_a, uint32_t* bits_b, int64_t n_alive, int64_t chunk,
                                         double* slot0, double* slot1, double* logpi, double* delta, uint8_t* flags, double eps,
                                         double gamma0, double gamma_sigma, uint32_t sweep0, int32_t k_max, double kmcmc_min,
                                         int64_t* nacc, int64_t* nsim, int32_t* k_done);
ABCDEZ_API int abcdez_smc_resample_gather_packed(abcdez_ctx* ctx, const uint32_t* inds, int64_t N, uint32_t* bits,
                                                 uint32_t* bits_other, double* slot0, double* slot1, const double* logpi,
                                                 const double* delta, double* nlogpi, double* ndelta, double* wns,
                                                 uint8_t* alive);
ABCDEZ_API int abcdez_packed_gather(abcdez_ctx* ctx, const uint32_t* bits, int64_t N, const double* slot0,
                                    const double* slot1, double* out);

/* S5+S6  abcdesmc_update_ws!(ws, alive, Ds, eps_k, eps_k_new, nparticles) src/abcdez_smc.jl:59-83
 *        followed by the driver's wprod/wnorm/Wns/alive lines :308-311 and get_ess :8,:323. */
ABCDEZ_API int abcdez_smc_reweight(abcdez_ctx* ctx, const double* delta, double* wns, uint8_t* alive, int64_t N,
                        double eps_old, double eps_new, double* wnorm, double* ess, int64_t* n_alive);
ABCDEZ_API int abcdez_get_ess(abcdez_ctx* ctx, const double* wns, int64_t N, double* ess);

/* S7  wsample_stratified!(rng, weights, inds)  src/abcdez_smc.jl:15-56 (0-based indices).
 *     draw = resampling number (RNG epoch).                                           */
ABCDEZ_API int abcdez_wsample_stratified(abcdez_ctx* ctx, const double* wns, int64_t N, uint32_t draw, uint32_t* inds);

/* S9  quantile(Ds[alive], alpha)  src/abcdez_smc.jl:301 (Julia default, type 7).
 *     Also returns the two order statistics it interpolates.                         */
ABCDEZ_API int abcdez_quantile_alive(abcdez_ctx* ctx, const double* delta, const uint8_t* alive, int64_t N,
                          int64_t n_alive_hint /* sum(alive) if known, else -1 */, double p,
                          double* q, double* xj, double* xj1);

/* S10  extrema(Ds) src/abcdez_smc.jl:286,364, src/abcdez_mc.jl:146,163;
 *      sum(Ds .> eps_target) src/abcdez_mc.jl:133,156.                               */
ABCDEZ_API int abcdez_extrema(abcdez_ctx* ctx, const double* delta, int64_t N, double* lo, double* hi);
ABCDEZ_API int abcdez_count_gt(abcdez_ctx* ctx, const double* delta, int64_t N, double thr, int64_t* count);

/* S4  abcdemc_swarm!(prior, dist!, varexternal, thetas, logpi, Ds, nthetas, nlogpi, nDs,
 *     eps_pop, eps_target, gamma0, gamma_sigma, nparticles, nsims, rng, ex, nblobs)
 *     src/abcdez_mc.jl:5-61 plus the copies of :140-143.
 *     rank_prepare builds the enumeration the "better particle" draw of mc:23 indexes into -- the particles with
 *     Ds <= eps_pop in index order, then the others sorted by (Ds, index); sorted_delta[p] = max(Ds[order[p]], eps_pop);
 *     cnt[i] = #{j : Ds[j] <= Ds[i]} = upper_bound(sorted_delta, Ds[i]) for every particle i with Ds[i] > eps_pop (the
 *     size of its candidate set; undefined for the others, which never draw, mc:19-20)
 *     -- with hand-written kernels (bucket ids over the window (eps_pop, dmax_hint], stable LSD radix passes, fix-up
 *     of shared buckets).  dmax_hint: maximum(Ds) as the driver knows it from mc:146; it only shapes the binning,
 *     any value gives the same result.  Asynchronous (no host synchronisation).
 *     By rank or by rejection: mc:23 is a uniform draw from {j : Ds[j] <= Ds[i]}; order = cnt = NULL makes mc_swarm draw it
 *     by rejection instead (uniform j over all particles until Ds[j] <= Ds[i]; include/abcdez_spec.h,
 *     abz_mc_better_by_rejection) -- no rank pass at all.  The spec's rule: a generation draws by rejection iff at least 1 / 16
 *     of the particles it reads lie at or below eps_target (abz_mc_draws_by_rejection; then every candidate set holds at
 *     least N / 16 particles).  mc_swarm takes the caller's word for it (and returns an error if a particle runs out of trials,
 *     which under the rule does not happen); abcdez_mc_generation and abcdez_mc_generation_async apply the rule themselves.
 *     mc_swarm also returns the reductions the driver takes of the generation it leaves behind, over the
 *     particles [i0, i0+n_local): n_above_target = sum(nDs .> eps_target) (mc:156), (dmin, dmax) = extrema(nDs)
 *     (mc:146,163) -- any of the three pointers may be NULL.                                                    */
ABCDEZ_API int abcdez_mc_rank_prepare(abcdez_ctx* ctx, const double* delta, int64_t N, double eps_pop, double dmax_hint,
                           uint32_t* order, double* sorted_delta, uint32_t* cnt);
ABCDEZ_API int abcdez_mc_swarm(abcdez_ctx* ctx, const uint32_t* order, const uint32_t* cnt, int64_t N,
                    const double* theta, const double* logpi, const double* delta,
                    double* ntheta, double* nlogpi, double* ndelta,
                    double eps_pop, double eps_target, double gamma0, double gamma_sigma,
                    int64_t i0, int64_t n_local, uint32_t sweep, int64_t* nsim, int64_t* n_above_target,
                    double* dmin, double* dmax);

/* One generation of abcdemc!'s loop body (src/abcdez_mc.jl:140-156) on one GPU in one call: rank_prepare when
 * dmax > eps_target (the host passes the extrema it got from the previous generation, mc:146) and the generation draws by
 * rank, then mc_swarm over all N particles.  n_above = #(Ds > eps_target) of the distances read (mc:133, or
 * *n_above_target of the generation before, mc:156), -1 = count them here.  One host synchronisation per generation. */
ABCDEZ_API int abcdez_mc_generation(abcdez_ctx* ctx, int64_t N, const double* theta, const double* logpi, const double* delta,
                         double* ntheta, double* nlogpi, double* ndelta, uint32_t* order, double* sorted_delta, uint32_t* cnt,
                         double eps_pop, double eps_target, double dmax, int64_t n_above, double gamma0, double gamma_sigma,
                         uint32_t sweep, int64_t* nsim, int64_t* n_above_target, double* dmin, double* dmax_out);
/* The same loop body WITHOUT a host synchronisation -- abcdemc!'s loop (mc:134-161) has no data-dependent exit, so the
 * host may run ahead of the device.  The extrema of mc:146 are taken from the sweep before, on the device (lo_hi = NULL),
 * or from lo_hi[0..1] (first generation of a run / after anything else changed the population); eps_pop =
 * max(eps_target, lo + alpha (hi - lo)) (mc:147) is evaluated on the device with the host driver's operations; do_rank = 0
 * skips the rank pass (legal once a redeemed generation reported max Ds <= eps_target: converged populations stay
 * converged and nobody draws from mc:23's sets).  By rank or by rejection: decided on the device from #(Ds > eps_target),
 * counted when a chain of generations starts and carried from sweep to sweep; once a redeemed generation has shown the
 * switch to rejection (it is final, include/abcdez_spec.h) the library launches no rank pass any more -- abcdez_mc_draw_stats.
 * *ticket numbers the generation; at most 8 may be in flight.
 * abcdez_mc_generation_wait redeems the tickets in issue order and waits for THAT generation only: nsim, #(new Ds >
 * eps_target) (mc:156), extrema of the new distances (mc:146, :163) and the eps_pop the generation ran with.          */
ABCDEZ_API int abcdez_mc_generation_async(abcdez_ctx* ctx, int64_t N, const double* theta, const double* logpi,
                                          const double* delta, double* ntheta, double* nlogpi, double* ndelta,
                                          uint32_t* order, double* sorted_delta, uint32_t* cnt, double alpha,
                                          double eps_target, const double* lo_hi, int32_t do_rank, double gamma0,
                                          double gamma_sigma, uint32_t sweep, int64_t* ticket);
ABCDEZ_API int abcdez_mc_generation_wait(abcdez_ctx* ctx, int64_t ticket, int64_t* nsim, int64_t* n_above_target,
                                         double* dmin, double* dmax, double* eps_pop);
/* The same generation on a population SHARDED over the ranks of the context's communicator (abcdez_comm_init; N divisible by the
 * number of ranks): rank r runs abcdemc_swarm! (src/abcdez_mc.jl:5-61) for the particles [r N / world, (r + 1) N / world) of the full
 * arrays every rank holds; behind the sweep, still without a host synchronisation, the ranks exchange the new rows / log-priors /
 * distances (/ blob stamps) in place and the generation's reductions (mc:146,156,163) in one group of RCCL collectives on the
 * context's stream; rank pass, eps_pop (mc:147) and the rank-or-rejection rule run replicated on the exchanged population, so every
 * rank takes the same decisions.  Tickets are redeemed with abcdez_mc_generation_wait, which returns the GLOBAL reductions on every
 * rank.  The first generation of a chain needs lo_hi (a rank's own sweep only knows its own particles' extrema). */
ABCDEZ_API int abcdez_mc_generation_sharded_async(abcdez_ctx* ctx, int64_t N, const double* theta, const double* logpi,
                                                  const double* delta, double* ntheta, double* nlogpi, double* ndelta,
                                                  uint32_t* order, double* sorted_delta, uint32_t* cnt, double alpha,
                                                  double eps_target, const double* lo_hi, int32_t do_rank, double gamma0,
                                                  double gamma_sigma, uint32_t sweep, int64_t* ticket);

/* T2  push_p over the population (src/abcdez_types.jl:20-23; result P, smc:382, mc:166). */
ABCDEZ_API int abcdez_push_p(abcdez_ctx* ctx, const double* theta, int64_t N, double* out);

/* Test hooks: evaluate the spec arithmetic on the device (fn: 0 log, 1 exp, 2 sincos2pi,
 * 3 rint, 4 floor, 5 sqrt, 6 x/y2, 7 table log, 8 table sincos2pi, 9 sqrt_pn, 10 lgamma,
 * 11 log-density of the context's prior factor (int)y2[i] at x[i], src/abcdez_priors.jl:41-45). */
ABCDEZ_API int abcdez_math_eval(abcdez_ctx* ctx, int fn, const double* x, double* y, double* y2, int64_t n);
ABCDEZ_API int abcdez_tree_sum(abcdez_ctx* ctx, const double* x, int64_t n, double* out);
/* Test hook: the per-particle scalar draws of one sweep -- donor ranks (src/abcdez_smc.jl:119-126), gamma =
 * gamma0 (1 + randn gamma_sigma) (smc:128), log(rand) of the accept test (smc:145) -- of particles [i0, i0+n) with
 * alive rank = index in a pool of n_pool, evaluated by the sweep kernels' own routine for a lane-group width of
 * `lanes` (1, 2, 4, 8, 16).  The results must not depend on `lanes`.                                      */
ABCDEZ_API int abcdez_draws_eval(abcdez_ctx* ctx, int lanes, int64_t i0, int64_t n, int64_t n_pool, uint32_t sweep,
                                 double gamma0, double gamma_sigma, uint32_t* ra, uint32_t* rb, double* gamma,
                                 double* log_u);

#ifdef __cplusplus
}
#endif
#endif /* ABCDEZ_HIP_H */
