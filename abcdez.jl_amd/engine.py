"""Population engine: device-resident particle state + one call per reference function.

``PopulationEngine`` owns the arrays the reference driver owns
(``θs, logπ, Δs, Wns, alive`` and their ``n*`` doubles, src/abcdez_smc.jl:242-275)
as torch tensors, shards the particles over the ranks of a ``torch.distributed``
process group, and forwards every population-sized step to an *ops* backend:

* :class:`HipOps` -- the product: ctypes calls into ``libabcdez_hip.so`` with raw
  device pointers (torch is plumbing: memory, streams, collectives);
* the test suite injects an oracle-backed ops object to exercise this file's host
  logic (sharding, collectives, buffer swapping) on CPU over ``gloo``.

Layout in HBM (per rank, every array FULL length N so donors can be any particle):
``theta[2][N][ld]`` f64 row-major ping-pong, ``logpi[2][N]``, ``delta[2][N]`` f64,
``wns[N]`` f64, ``alive[N]`` u8, ``alive_idx[N]``, ``arank[N]``, ``inds[N]`` u32,
and for abcdemc ``order[N]`` u32 + ``sorted_delta[N]`` f64.

Multi-GPU: rank r updates the contiguous index range
[r N/G, (r+1) N/G) and every rank keeps the whole population, so the next sweep's
donors come from the global population.  RNG counters are keyed by the global
particle index, so results do not depend on G.  The cheap per-generation passes
(quantile, reweight, compaction, resampling indices) run replicated on the full arrays.

* ``storage="classic"`` (abcdemc; abcdesmc on request): after every sweep the new
  rows / logπ / Δ of all ranks are exchanged with one in-place all-gather each.
* ``storage="rows"`` (abcdesmc default): the replicas exchange ONE BYTE per particle
  and sweep -- the accept flag -- and rebuild the accepted proposals themselves
  (``smc_replay_rows``: the proposal is a function of replicated rows and of
  counter-based random numbers).  Distances are all-gathered once per generation
  (quantile / reweight need them), log-priors only before a resampling, and the
  resampling gathers run replicated.  xGMI carries 1 + 8/K bytes per particle and
  sweep instead of 8 ld + 16.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .model import ModelSpec

PACKED_ALIGN = 64        # sub-ranges of the packed prefix start / end at multiples of this (include/abcdez_hip.h)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class HipOps:
    """ctypes -> HIP kernels.  Tensors must live on this ops' CUDA(HIP) device."""

    name = "hip"

    def __init__(self, spec: ModelSpec, device_index: Optional[int] = None, lanes: int = 0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.AbcdezError("no HIP device visible: the population loop runs only on the GPU (no CPU fallback)")
        if device_index is None:
            device_index = torch.cuda.current_device()
        self.device = torch.device("cuda", device_index)
        self.spec = spec
        self._data_host = np.ascontiguousarray(spec.data, dtype=np.float64)
        cm = spec.cstruct(self._data_host.ctypes.data if self._data_host.size else None)
        ctx = C.c_void_p()
        source = getattr(spec.sim, "source", None)
        if source is not None:      # user-supplied simulator: compiled by the library with hiprtc
            _lib.check(self.lib, self.lib.abcdez_ctx_create_user(C.byref(cm), source.encode("utf-8"), device_index,
                                                                 C.byref(ctx)))
        else:
            _lib.check(self.lib, self.lib.abcdez_ctx_create(C.byref(cm), device_index, C.byref(ctx)))
        self.ctx = ctx
        if lanes:
            _lib.check(self.lib, self.lib.abcdez_ctx_set_lanes(self.ctx, lanes))
        self.use_current_stream()

    def use_current_stream(self):
        s = torch.cuda.current_stream(self.device)
        _lib.check(self.lib, self.lib.abcdez_ctx_set_stream(self.ctx, C.c_void_p(s.cuda_stream)))

    def reserve(self, n: int):
        """size the library's workspace for n particles up front (otherwise the first resampling grows it mid-run)"""
        _lib.check(self.lib, self.lib.abcdez_ctx_reserve(self.ctx, n))

    def layout(self):
        ld, L, Cc = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.lib, self.lib.abcdez_ctx_get_layout(self.ctx, C.byref(ld), C.byref(L), C.byref(Cc)))
        return ld.value, L.value, Cc.value

    def set_timing(self, on: bool):
        _lib.check(self.lib, self.lib.abcdez_ctx_set_timing(self.ctx, int(on)))

    def get_timing(self):
        """(sweep-kernel ms, launches, particle-updates) accumulated since set_timing(True)"""
        ms, n, u = C.c_double(), C.c_int64(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_ctx_get_timing(self.ctx, C.byref(ms), C.byref(n), C.byref(u)))
        return ms.value, n.value, u.value

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.abcdez_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- one method per C-ABI entry point -------------------------------------------------
    def init(self, theta, logpi, delta, i0, n):
        _lib.check(self.lib, self.lib.abcdez_init(self.ctx, _ptr(theta), _ptr(logpi), _ptr(delta), i0, n))

    def alive_compact(self, alive, alive_idx, arank, n_known=None) -> int:
        if n_known is not None:     # sum(alive) known from the reweight: no host sync
            _lib.check(self.lib, self.lib.abcdez_alive_compact(self.ctx, _ptr(alive), alive.numel(), _ptr(alive_idx),
                                                               _ptr(arank), None))
            return n_known
        n = C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_alive_compact(self.ctx, _ptr(alive), alive.numel(), _ptr(alive_idx),
                                                           _ptr(arank), C.byref(n)))
        return n.value

    def smc_swarm(self, alive_idx, arank, n_alive, r_lo, r_hi, cur, nxt, eps, gamma0, gsig, i0, n_local, copy_dead,
                  sweep, dead_synced=None):
        nacc, nsim = C.c_int64(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_smc_swarm(
            self.ctx, _ptr(alive_idx), _ptr(arank), n_alive, r_lo, r_hi, _ptr(cur[0]), _ptr(cur[1]), _ptr(cur[2]),
            _ptr(nxt[0]), _ptr(nxt[1]), _ptr(nxt[2]), eps, gamma0, gsig, i0, n_local, int(copy_dead),
            _ptr(dead_synced), sweep, C.byref(nacc), C.byref(nsim)))
        return nacc.value, nsim.value

    # ---- row-store mode: include/abcdez_hip.h, abcdez_smc_swarm_rows (+ _shard / replay for sharded runs) ----
    supports_rows = True

    def alive_compact_rows(self, alive, cur_row, alive_row, arank):
        _lib.check(self.lib, self.lib.abcdez_alive_compact_rows(self.ctx, _ptr(alive), alive.numel(), _ptr(cur_row),
                                                                _ptr(alive_row), _ptr(arank), None))

    def smc_swarm_rows(self, alive_row, alive_row_out, n_alive, slot0, slot1, logpi, delta, eps, gamma0, gsig, sweep):
        nacc, nsim = C.c_int64(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_smc_swarm_rows(
            self.ctx, _ptr(alive_row), _ptr(alive_row_out), n_alive, _ptr(slot0), _ptr(slot1), _ptr(logpi), _ptr(delta),
            eps, gamma0, gsig, sweep, C.byref(nacc), C.byref(nsim)))
        return nacc.value, nsim.value

    def smc_swarm_rows_shard(self, alive_row, alive_row_out, n_alive, r_lo, r_hi, slot0, slot1, logpi, delta, accepted,
                             eps, gamma0, gsig, sweep, want_counts=True):
        """want_counts=False: no host synchronisation (the replay reports the other ranks' counters, the flags
        carry this rank's)"""
        nacc, nsim = C.c_int64(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_smc_swarm_rows_shard(
            self.ctx, _ptr(alive_row), _ptr(alive_row_out), n_alive, r_lo, r_hi, _ptr(slot0), _ptr(slot1), _ptr(logpi),
            _ptr(delta), _ptr(accepted), eps, gamma0, gsig, sweep, C.byref(nacc) if want_counts else None,
            C.byref(nsim) if want_counts else None))
        return (nacc.value, nsim.value) if want_counts else None

    def smc_replay_rows(self, alive_row, alive_row_out, n_alive, skip_lo, skip_hi, slot0, slot1, accepted, gamma0, gsig,
                        sweep):
        nacc, nsim = C.c_int64(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_smc_replay_rows(
            self.ctx, _ptr(alive_row), _ptr(alive_row_out), n_alive, skip_lo, skip_hi, _ptr(slot0), _ptr(slot1),
            _ptr(accepted), gamma0, gsig, sweep, C.byref(nacc), C.byref(nsim)))
        return nacc.value, nsim.value

    def rows_commit(self, alive_row, n_alive, cur_row):
        _lib.check(self.lib, self.lib.abcdez_rows_commit(self.ctx, _ptr(alive_row), n_alive, _ptr(cur_row)))

    def smc_resample_gather_rows(self, inds, cur_row, slot0, slot1, logpi, delta, nlogpi, ndelta, wns, alive):
        _lib.check(self.lib, self.lib.abcdez_smc_resample_gather_rows(
            self.ctx, _ptr(inds), inds.numel(), _ptr(cur_row), _ptr(slot0), _ptr(slot1), _ptr(logpi), _ptr(delta),
            _ptr(nlogpi), _ptr(ndelta), _ptr(wns), _ptr(alive)))

    def rows_gather(self, cur_row, slot0, slot1, out):
        _lib.check(self.lib, self.lib.abcdez_rows_gather(self.ctx, _ptr(cur_row), cur_row.numel(), _ptr(slot0),
                                                         _ptr(slot1), _ptr(out)))

    # ---- packed population: include/abcdez_hip.h, abcdez_smc_partition / abcdez_smc_swarm_packed / ... ----
    supports_packed = True

    def smc_partition(self, n_prev, n_new, alive, bits, bits_other, slot0, slot1, logpi, delta, wns):
        _lib.check(self.lib, self.lib.abcdez_smc_partition(self.ctx, _ptr(alive), alive.numel(), n_prev, n_new, _ptr(bits),
                                                           _ptr(bits_other), _ptr(slot0), _ptr(slot1), _ptr(logpi),
                                                           _ptr(delta), _ptr(wns)))

    def smc_prologue_packed(self, delta, wns, alive, n_prev, alpha, eps_prev, eps_target, eps_k, ess_min, bits, bits_other,
                            slot0, slot1, logpi):
        """extrema, eps (smc:301), reweight (smc:305-311), ESS and -- unless the driver is about to resample -- the
        partition, in one call and one host synchronisation -> (eps, wnorm, ess, n_alive, partitioned, lo, hi)"""
        eps, q, wnorm, ess, lo, hi = (C.c_double() for _ in range(6))
        na, part = C.c_int64(), C.c_int32()
        _lib.check(self.lib, self.lib.abcdez_smc_prologue_packed(
            self.ctx, _ptr(delta), _ptr(wns), _ptr(alive), delta.numel(), n_prev, alpha, eps_prev, eps_target, eps_k, ess_min,
            _ptr(bits), _ptr(bits_other), _ptr(slot0), _ptr(slot1), _ptr(logpi), C.byref(eps), C.byref(q), C.byref(wnorm),
            C.byref(ess), C.byref(na), C.byref(part), C.byref(lo), C.byref(hi)))
        return eps.value, wnorm.value, ess.value, na.value, bool(part.value), lo.value, hi.value

    def smc_swarm_packed(self, bits, bits_out, n_alive, r_lo, r_hi, slot0, slot1, logpi, delta, flags, eps, gamma0, gsig,
                         sweep, want_counts=True):
        """want_counts=False: no host synchronisation (the replay reports the sweep's global counters)"""
        nacc, nsim = C.c_int64(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_smc_swarm_packed(
            self.ctx, _ptr(bits), _ptr(bits_out), n_alive, r_lo, r_hi, _ptr(slot0), _ptr(slot1), _ptr(logpi), _ptr(delta),
            _ptr(flags), eps, gamma0, gsig, sweep, C.byref(nacc) if want_counts else None,
            C.byref(nsim) if want_counts else None))
        return (nacc.value, nsim.value) if want_counts else None

    def smc_replay_packed(self, bits, bits_out, n_alive, skip_lo, skip_hi, slot0, slot1, logpi, flags, gamma0, gsig, sweep):
        nacc, nsim = C.c_int64(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_smc_replay_packed(
            self.ctx, _ptr(bits), _ptr(bits_out), n_alive, skip_lo, skip_hi, _ptr(slot0), _ptr(slot1), _ptr(logpi),
            _ptr(flags), gamma0, gsig, sweep, C.byref(nacc), C.byref(nsim)))
        return nacc.value, nsim.value

    def smc_resample_gather_packed(self, inds, bits, bits_other, slot0, slot1, logpi, delta, nlogpi, ndelta, wns, alive):
        _lib.check(self.lib, self.lib.abcdez_smc_resample_gather_packed(
            self.ctx, _ptr(inds), inds.numel(), _ptr(bits), _ptr(bits_other), _ptr(slot0), _ptr(slot1), _ptr(logpi),
            _ptr(delta), _ptr(nlogpi), _ptr(ndelta), _ptr(wns), _ptr(alive)))

    def packed_gather(self, bits, slot0, slot1, out):
        _lib.check(self.lib, self.lib.abcdez_packed_gather(self.ctx, _ptr(bits), out.shape[0], _ptr(slot0), _ptr(slot1),
                                                           _ptr(out)))

    def smc_reweight(self, delta, wns, alive, eps_old, eps_new):
        wnorm, ess, na = C.c_double(), C.c_double(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_smc_reweight(self.ctx, _ptr(delta), _ptr(wns), _ptr(alive), delta.numel(),
                                                          eps_old, eps_new, C.byref(wnorm), C.byref(ess), C.byref(na)))
        return wnorm.value, ess.value, na.value

    def get_ess(self, wns) -> float:
        ess = C.c_double()
        _lib.check(self.lib, self.lib.abcdez_get_ess(self.ctx, _ptr(wns), wns.numel(), C.byref(ess)))
        return ess.value

    def tree_sum(self, x) -> float:
        out = C.c_double()
        _lib.check(self.lib, self.lib.abcdez_tree_sum(self.ctx, _ptr(x), x.numel(), C.byref(out)))
        return out.value

    def wsample_stratified(self, wns, draw, inds):
        _lib.check(self.lib, self.lib.abcdez_wsample_stratified(self.ctx, _ptr(wns), wns.numel(), draw, _ptr(inds)))

    def smc_resample_gather(self, inds, i0, n_local, cur, nxt, wns, alive):
        _lib.check(self.lib, self.lib.abcdez_smc_resample_gather(
            self.ctx, _ptr(inds), inds.numel(), i0, n_local, _ptr(cur[0]), _ptr(cur[1]), _ptr(cur[2]),
            _ptr(nxt[0]), _ptr(nxt[1]), _ptr(nxt[2]), _ptr(wns), _ptr(alive)))

    def quantile_alive(self, delta, alive, p, n_alive=-1):
        q, a, b = C.c_double(), C.c_double(), C.c_double()
        _lib.check(self.lib, self.lib.abcdez_quantile_alive(self.ctx, _ptr(delta), _ptr(alive), delta.numel(),
                                                            n_alive, p, C.byref(q), C.byref(a), C.byref(b)))
        return q.value, a.value, b.value

    def extrema(self, delta):
        lo, hi = C.c_double(), C.c_double()
        _lib.check(self.lib, self.lib.abcdez_extrema(self.ctx, _ptr(delta), delta.numel(), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def count_gt(self, delta, thr) -> int:
        c = C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_count_gt(self.ctx, _ptr(delta), delta.numel(), thr, C.byref(c)))
        return c.value

    def mc_rank_prepare(self, delta, eps_pop, dmax_hint, order, sorted_delta, cnt):
        """asynchronous: no host synchronisation"""
        _lib.check(self.lib, self.lib.abcdez_mc_rank_prepare(self.ctx, _ptr(delta), delta.numel(), eps_pop, dmax_hint,
                                                             _ptr(order), _ptr(sorted_delta), _ptr(cnt)))

    def mc_swarm(self, order, cnt, cur, nxt, eps_pop, eps_target, gamma0, gsig, i0, n_local, sweep):
        """-> (nsim, #(new Ds > eps_target), min, max of the new Ds) over particles [i0, i0+n_local): ONE host sync"""
        nsim, ngt, lo, hi = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(self.lib, self.lib.abcdez_mc_swarm(
            self.ctx, _ptr(order), _ptr(cnt), cur[1].numel(), _ptr(cur[0]), _ptr(cur[1]), _ptr(cur[2]),
            _ptr(nxt[0]), _ptr(nxt[1]), _ptr(nxt[2]), eps_pop, eps_target, gamma0, gsig, i0, n_local, sweep,
            C.byref(nsim), C.byref(ngt), C.byref(lo), C.byref(hi)))
        return nsim.value, ngt.value, lo.value, hi.value

    def mc_generation(self, cur, nxt, order, sorted_delta, cnt, eps_pop, eps_target, dmax, gamma0, gsig, sweep):
        """rank pass (if dmax > eps_target) + sweep over all particles: one library call, one host sync"""
        nsim, ngt, lo, hi = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(self.lib, self.lib.abcdez_mc_generation(
            self.ctx, cur[1].numel(), _ptr(cur[0]), _ptr(cur[1]), _ptr(cur[2]), _ptr(nxt[0]), _ptr(nxt[1]), _ptr(nxt[2]),
            _ptr(order), _ptr(sorted_delta), _ptr(cnt), eps_pop, eps_target, dmax, gamma0, gsig, sweep, C.byref(nsim),
            C.byref(ngt), C.byref(lo), C.byref(hi)))
        return nsim.value, ngt.value, lo.value, hi.value

    def push_p(self, theta, out):
        _lib.check(self.lib, self.lib.abcdez_push_p(self.ctx, _ptr(theta), theta.shape[0], _ptr(out)))
        _lib.check(self.lib, self.lib.abcdez_sync(self.ctx))

    # ---- blobs: stamps carried with the distances, simulated data rebuilt on demand (include/abcdez_hip.h) ----
    def set_stamps(self, cur, nxt):
        _lib.check(self.lib, self.lib.abcdez_ctx_set_stamps(self.ctx, _ptr(cur), _ptr(nxt)))

    def blob_width(self) -> int:
        w = C.c_int32()
        _lib.check(self.lib, self.lib.abcdez_blob_width(self.ctx, C.byref(w)))
        return w.value

    def blob_eval(self, theta, stamp, blob, delta_out):
        _lib.check(self.lib, self.lib.abcdez_blob_eval(self.ctx, _ptr(theta), _ptr(stamp), theta.shape[0], _ptr(blob),
                                                       _ptr(delta_out)))
        _lib.check(self.lib, self.lib.abcdez_sync(self.ctx))

    def math_eval(self, fn, x, y, y2=None):
        _lib.check(self.lib, self.lib.abcdez_math_eval(self.ctx, fn, _ptr(x), _ptr(y), _ptr(y2), x.numel()))

    def draws_eval(self, lanes, i0, n_pool, sweep, gamma0, gsig, ra, rb, g, log_u):
        """test hook: the sweep kernels' per-particle draws for a lane-group width of ``lanes``"""
        _lib.check(self.lib, self.lib.abcdez_draws_eval(self.ctx, lanes, i0, ra.numel(), n_pool, sweep, gamma0, gsig,
                                                        _ptr(ra), _ptr(rb), _ptr(g), _ptr(log_u)))


class PopulationEngine:
    """Device-resident population + the reference's per-generation functions."""

    def __init__(self, spec: ModelSpec, nparticles: int, process_group=None, ops=None, lanes: int = 0,
                 storage: str = "classic", force_collectives: bool = False):
        """storage = "classic": two full generations' arrays, every sweep writes the next one (the reference's
        thetas / nthetas).  storage = "rows": row store -- two slots per particle, only accepted proposals are
        written (abcdez_smc_swarm_rows); sharded: accept-flag exchange + replay (module docstring); abcdesmc only.
        force_collectives: take the sharded code path (collectives, shard sweep + replay) even in a group of ONE
        rank -- lets a single-GPU box exercise the RCCL calls (tests)."""
        self.spec = spec
        self.N = int(nparticles)
        self.pg = process_group
        if self.pg is not None:
            import torch.distributed as dist

            self.rank = dist.get_rank(self.pg)
            self.world = dist.get_world_size(self.pg)
            self._backend = str(dist.get_backend(self.pg))
        else:
            self.rank, self.world = 0, 1
            self._backend = "none"
        if self.N % self.world:
            raise ValueError(f"nparticles ({self.N}) must be divisible by the number of ranks ({self.world})")
        self.n_local = self.N // self.world
        self.lo = self.rank * self.n_local
        self.hi = self.lo + self.n_local
        self.ops = ops if ops is not None else HipOps(spec, lanes=lanes)
        if hasattr(self.ops, "reserve"):
            self.ops.reserve(self.N)
        dev = self.ops.device
        self.device = dev
        N, ld = self.N, spec.ld
        f64 = dict(dtype=torch.float64, device=dev)
        self.rows_mode = storage == "rows" and getattr(self.ops, "supports_rows", False)
        self.packed = storage == "packed" and getattr(self.ops, "supports_packed", False)
        if storage == "packed" and not self.packed:
            raise ValueError("this ops backend has no packed storage")
        self._collectives = self.world > 1 or (force_collectives and self.pg is not None)
        self.sharded_rows = self.rows_mode and self._collectives
        self.sharded_packed = self.packed and self._collectives
        # (theta, logpi, delta) x 2: generation t and t+1 (smc:337-350); in row-store / packed mode the two theta
        # arrays are the two slots of the store and only (logpi, delta) ping-pong -- at resamplings.
        # Sharded packed runs exchange [r_lo, r_lo + chunk) pieces of the per-position arrays: room for G chunks.
        self._npad = N + (self.world * PACKED_ALIGN if self.sharded_packed else 0)
        self._full = [(torch.zeros(self._npad, **f64), torch.zeros(self._npad, **f64)) for _ in range(2)]
        self.buf = [(torch.zeros((N, ld), **f64), self._full[k][0][:N], self._full[k][1][:N]) for k in range(2)]
        self.cur = 0
        if self.packed:
            nw = (N + 31) // 32
            self.bits = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(2)]   # current slot of every position
            self.bc = 0
            self.n_prev = N                 # length of the alive prefix
            self.flags = torch.zeros(self._npad, dtype=torch.uint8, device=dev) if self.sharded_packed else None
            self.chunk = 0
        if self.rows_mode:
            self.cur_row = torch.arange(N, dtype=torch.int32, device=dev)      # particle | slot << 31
            self.alive_row = [torch.zeros(N, dtype=torch.int32, device=dev) for _ in range(2)]
            self.ar = 0
            self._rows_dirty = False
            self._rows_n = 0
        if self.sharded_rows:
            self.accepted = torch.zeros(N, dtype=torch.uint8, device=dev)   # accept flags of the last sweep, by particle
        # blobs (spec.n_blob > 0): one stamp per particle, ping-ponging with (logpi, delta)
        self.blob_on = getattr(spec, "n_blob", 0) > 0
        self.stamp = [torch.zeros(N, dtype=torch.int64, device=dev) for _ in range(2)] if self.blob_on else None
        self._delta_work = None      # sharded row store: distance all-gather in flight (overlapped with the last replay)
        self._prof = None            # optional event timing of the sharded sweep's phases (enable_phase_timing)
        self._delta_stale = False    # sharded row store: other ranks' distances / log-priors not yet fetched
        self._logpi_stale = False
        self.wns = torch.full((N,), 1.0 / N, **f64)
        self.alive = torch.ones(N, dtype=torch.uint8, device=dev)
        self.alive_idx = torch.zeros(N, dtype=torch.int32, device=dev)
        self.arank = torch.zeros(N, dtype=torch.int32, device=dev)
        self.inds = torch.zeros(N, dtype=torch.int32, device=dev)
        self.order = None
        self.sorted_delta = None
        self.n_alive = N
        self.r_lo, self.r_hi = self.lo, self.hi
        self.row_synced = torch.zeros(N, dtype=torch.uint8, device=dev)   # dead rows already carried to both buffers
        self.sweep = 0          # global sweep number = RNG epoch of the swarm kernels
        self.draw = 0           # resampling number = RNG epoch of the stratified draws
        self._dead_carried = True

    # ------------------------------------------------------------------ helpers
    @property
    def state(self):
        """(theta, logpi, delta) of the current generation.  Row-store mode gathers the current rows into a
        fresh array (tests / results only; the hot path never calls this)."""
        if self.packed:
            self._sync_delta()
            th = torch.empty_like(self.buf[0][0])
            self.ops.packed_gather(self.bits[self.bc], self.buf[0][0], self.buf[1][0], th)
            return (th, self.buf[self.cur][1], self.buf[self.cur][2])
        if not self.rows_mode:
            return self.buf[self.cur]
        self._rows_commit()
        self._sync_delta()
        self._sync_logpi()
        th = torch.empty_like(self.buf[0][0])
        self.ops.rows_gather(self.cur_row, self.buf[0][0], self.buf[1][0], th)
        return (th, self.buf[self.cur][1], self.buf[self.cur][2])

    def _rows_commit(self):
        if self.rows_mode and self._rows_dirty:
            # the list was built for the alive set of the LAST compaction (a reweight may have shrunk n_alive since)
            self.ops.rows_commit(self.alive_row[self.ar], self._rows_n, self.cur_row)
            self._rows_dirty = False

    @property
    def delta(self):
        self._sync_delta()
        return self.buf[self.cur][2]

    def _sync_delta(self):
        """sharded row store: fetch the other ranks' distances (once per generation, before the first consumer)"""
        if self._delta_work is not None:          # started behind the generation's last replay: just join it
            self._mark("delta_allgather_wait", 0)
            self._delta_work.wait()
            self._delta_work = None
            self._mark("delta_allgather_wait", 1)
            self._delta_stale = False
        if self._delta_stale:
            self._mark("delta_allgather", 0)
            if self.sharded_packed:
                self._allgather_chunks(self._full[self.cur][1])
            else:
                self._allgather_state((self.buf[self.cur][2],))
            self._mark("delta_allgather", 1)
            self._delta_stale = False

    def _start_delta_allgather(self):
        """sharded row store, last sweep of a generation: the owners' distances are final once the shard sweep has
        run, so their all-gather goes out on the collective stream while this rank replays the other shards."""
        import torch.distributed as dist

        if self._backend == "nccl" or self.device.type == "cpu":
            if self.sharded_packed:
                t = self._full[self.cur][1]
                self._delta_work = dist.all_gather_into_tensor(t[:self.world * self.chunk],
                                                               t[self.rank * self.chunk:(self.rank + 1) * self.chunk],
                                                               group=self.pg, async_op=True)
                return
            t = self.buf[self.cur][2]
            self._delta_work = dist.all_gather_into_tensor(t, t[self.lo:self.hi], group=self.pg, async_op=True)

    # ---- optional phase timing of the sharded row-store sweep (bench.py's multi-GPU breakdown) ----
    def enable_phase_timing(self):
        """record device events around the phases of every sharded sweep: own shard sweep, flag all-gather, replay,
        and the per-generation distance all-gather (device tensors only)"""
        self._prof = {} if self.device.type == "cuda" else None

    def _mark(self, phase, end):
        if self._prof is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._prof.setdefault(phase, []).append(ev)

    def phase_timing(self) -> dict:
        """{phase: (count, total ms)} since enable_phase_timing()"""
        if not self._prof:
            return {}
        torch.cuda.synchronize()
        out = {}
        for phase, evs in self._prof.items():
            pairs = list(zip(evs[0::2], evs[1::2]))
            out[phase] = (len(pairs), sum(a.elapsed_time(b) for a, b in pairs))
        return out

    def _sync_logpi(self):
        """sharded row store: fetch the other ranks' log-priors and blob stamps (read by other ranks only when
        resampling, smc:97,99)"""
        if self._logpi_stale:
            self._allgather_state((self.buf[self.cur][1],) + ((self.stamp[self.cur],) if self.blob_on else ()))
            self._logpi_stale = False

    def _bind_stamps(self):
        """blobs: tell the ops which stamp arrays belong to the current / next generation"""
        if self.blob_on:
            self.ops.set_stamps(self.stamp[self.cur], self.stamp[1 - self.cur])
        else:
            self.ops.set_stamps(None, None)

    def alive_indices(self) -> torch.Tensor:
        """particle indices of the alive list of the last compaction (int64)"""
        if self.packed:
            return torch.arange(self.n_prev, dtype=torch.int64, device=self.device)
        if self.rows_mode:
            return self.alive_row[self.ar][:self._rows_n].to(torch.int64) & 0x7FFFFFFF
        return self.alive_idx[:self.n_alive].to(torch.int64)

    @property
    def other(self):
        return self.buf[1 - self.cur]

    def _swap(self):
        self.cur = 1 - self.cur

    def _allgather_state(self, bufs):
        """exchange entries [lo, hi) of per-particle arrays (theta, logpi, delta, ...) -- one all-gather per array.

        RCCL ("nccl" backend): in place, straight between the device buffers over xGMI.
        Any other backend (gloo in the CPU tests): CPU tensors in place; device tensors are
        staged through host memory, so the path is deterministic on every rank."""
        if not self._collectives:
            return
        import torch.distributed as dist

        direct = self._backend == "nccl" or self.device.type == "cpu"
        for t in bufs:
            if direct:
                dist.all_gather_into_tensor(t, t[self.lo:self.hi], group=self.pg)
            else:
                host = torch.empty(t.shape, dtype=t.dtype)
                dist.all_gather_into_tensor(host, t[self.lo:self.hi].cpu(), group=self.pg)
                t.copy_(host)

    def _allgather_chunks(self, t):
        """sharded packed runs: rank r owns the positions [r chunk, (r + 1) chunk) of the prefix (chunk = a multiple of
        PACKED_ALIGN covering n_alive / G); exchange those pieces of a per-position array of length >= G chunk"""
        import torch.distributed as dist

        c, g = self.chunk, self.world
        if self._backend == "nccl" or self.device.type == "cpu":
            dist.all_gather_into_tensor(t[:g * c], t[self.rank * c:(self.rank + 1) * c], group=self.pg)
        else:
            host = torch.empty(g * c, dtype=t.dtype)
            dist.all_gather_into_tensor(host, t[self.rank * c:(self.rank + 1) * c].cpu(), group=self.pg)
            t[:g * c].copy_(host)

    def _allreduce_counts(self, *vals):
        if not self._collectives:
            return vals
        import torch.distributed as dist

        t = torch.tensor(vals, dtype=torch.int64, device=self.device if self._backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return tuple(int(v) for v in t.tolist())

    # ------------------------------------------------------------------ S1
    def init_population(self):
        if self._delta_work is not None:
            self._delta_work.wait()
            self._delta_work = None
        if self.rows_mode:       # a fresh population lives in slot `cur` of every particle (also when an engine is re-used)
            self.cur_row.copy_(torch.arange(self.N, dtype=torch.int32, device=self.device))
            if self.cur:
                self.cur_row.bitwise_or_(-(1 << 31))
            self._rows_dirty = False
            self._rows_n = 0
        if self.packed:          # a fresh population lives in slot `cur` of every position
            for b in self.bits:
                b.fill_(-1 if self.cur else 0)
            self.n_prev = self.N
        self._delta_stale = self._logpi_stale = False
        th, lp, dl = self.buf[self.cur]
        self._bind_stamps()
        self.ops.init(th, lp, dl, self.lo, self.n_local)
        self._allgather_state(self.buf[self.cur] + ((self.stamp[self.cur],) if self.blob_on else ()))

    def reset_weights(self):  # smc:266-270
        self.wns.fill_(1.0 / self.N)
        self.alive.fill_(1)
        self.n_alive = self.N
        if self.packed:
            self.n_prev = self.N
        self._dead_carried = True
        self.row_synced.zero_()

    # ------------------------------------------------------------------ S9, S10
    def quantile_alive(self, alpha: float) -> float:
        if self.packed:      # the alive particles are the prefix [0, n_prev): the dead tail is not even read
            n = self.n_prev
            return self.ops.quantile_alive(self.delta[:n], self.alive[:n], alpha, self.n_alive)[0]
        return self.ops.quantile_alive(self.delta, self.alive, alpha, self.n_alive)[0]

    def extrema(self):
        return self.ops.extrema(self.delta)

    def count_gt(self, thr: float) -> int:
        return self.ops.count_gt(self.delta, thr)

    # ------------------------------------------------------------------ S5, S6
    def smc_reweight(self, eps_old: float, eps_new: float):
        if self.packed:
            # prefix only: the dead tail has weight +0.0 and contributes exact zeros to the fixed summation trees
            n = self.n_prev
            wnorm, ess, n_alive = self.ops.smc_reweight(self.delta[:n], self.wns[:n], self.alive[:n], eps_old, eps_new)
            self.n_alive = n_alive
            return wnorm, ess, n_alive
        wnorm, ess, n_alive = self.ops.smc_reweight(self.delta, self.wns, self.alive, eps_old, eps_new)
        self.n_alive = n_alive
        self._dead_carried = False
        return wnorm, ess, n_alive

    def get_ess(self) -> float:
        return self.ops.get_ess(self.wns)

    def smc_prologue(self, alpha: float, eps_prev: float, eps_target: float, eps_k: float, ess_min: float):
        """Everything the driver does between two groups of sweeps except the resampling itself: extrema(Ds) of the
        generation that just ended (smc:364), eps = max(min(quantile(Ds[alive], alpha), eps_prev), eps_target)
        (smc:301), reweight + ESS (smc:305-311, :323) -> (eps, wnorm, ess, n_alive, (lo, hi)).  On the packed HIP
        population this is ONE library call with ONE host synchronisation (which also partitions the population
        unless ESS < ess_min announces a resampling); elsewhere the three separate calls."""
        if self.packed and hasattr(self.ops, "smc_prologue_packed"):
            self._sync_delta()
            self._bind_stamps()
            cur = self.buf[self.cur]
            eps, wnorm, ess, n_alive, part, lo, hi = self.ops.smc_prologue_packed(
                cur[2], self.wns, self.alive, self.n_prev, alpha, eps_prev, eps_target, eps_k, ess_min, self.bits[self.bc],
                self.bits[1 - self.bc], self.buf[0][0], self.buf[1][0], cur[1])
            self.n_alive = n_alive
            if part:
                self.n_prev = n_alive
            return eps, wnorm, ess, n_alive, (lo, hi)
        rng = self.extrema()
        eps = max(min(self.quantile_alive(alpha), eps_prev), eps_target)
        wnorm, ess, n_alive = self.smc_reweight(eps_k, eps)
        return eps, wnorm, ess, n_alive, rng

    # ------------------------------------------------------------------ S7, S8
    def smc_resample(self):
        self.ops.wsample_stratified(self.wns, self.draw, self.inds)
        self.draw += 1
        self._bind_stamps()
        if self.packed:
            self._sync_delta()
            cur, oth = self.buf[self.cur], self.buf[1 - self.cur]
            self.ops.smc_resample_gather_packed(self.inds, self.bits[self.bc], self.bits[1 - self.bc], self.buf[0][0],
                                                self.buf[1][0], cur[1], cur[2], oth[1], oth[2], self.wns, self.alive)
            self._swap()
            self.n_alive = self.n_prev = self.N
            return
        if self.rows_mode:
            self._rows_commit()
            self._sync_delta()
            self._sync_logpi()
            cur, oth = self.buf[self.cur], self.buf[1 - self.cur]
            self.ops.smc_resample_gather_rows(self.inds, self.cur_row, self.buf[0][0], self.buf[1][0], cur[1], cur[2],
                                              oth[1], oth[2], self.wns, self.alive)
            self._swap()
            self.n_alive = self.N
            return
        self.ops.smc_resample_gather(self.inds, self.lo, self.n_local, self.state, self.other, self.wns, self.alive)
        if self._collectives:
            self.wns.fill_(1.0 / self.N)   # the other ranks' ranges (smc:102-103)
            self.alive.fill_(1)
        self._allgather_state(self.other + ((self.stamp[1 - self.cur],) if self.blob_on else ()))
        self._swap()
        self.n_alive = self.N
        self._dead_carried = True
        self.row_synced.zero_()

    # ------------------------------------------------------------------ alive list + S2, S3
    def alive_compact(self) -> int:
        if self.packed:
            # no alive list: the population is partitioned so that the alive particles are the positions [0, n_alive)
            if self.n_alive < self.n_prev:
                self._sync_delta()
                cur = self.buf[self.cur]
                self._bind_stamps()
                self.ops.smc_partition(self.n_prev, self.n_alive, self.alive, self.bits[self.bc], self.bits[1 - self.bc],
                                       self.buf[0][0], self.buf[1][0], cur[1], cur[2], self.wns)
                self.n_prev = self.n_alive
            if self.sharded_packed:     # positions [r_lo, r_hi) of the prefix are this rank's for the coming sweeps
                per = -(-self.n_alive // self.world)
                self.chunk = -(-per // PACKED_ALIGN) * PACKED_ALIGN
                self.r_lo = min(self.rank * self.chunk, self.n_alive)
                self.r_hi = min(self.r_lo + self.chunk, self.n_alive)
            else:
                self.r_lo, self.r_hi = 0, self.n_alive
            return self.n_alive
        if self.rows_mode:
            self._rows_commit()
            self.ops.alive_compact_rows(self.alive, self.cur_row, self.alive_row[self.ar], self.arank)
            self._rows_n = self.n_alive
            if self.sharded_rows:   # alive ranks of this rank's particles: #alive below lo, #alive in [lo, hi)
                below = self.alive[:self.lo].sum(dtype=torch.int64)
                mine = self.alive[self.lo:self.hi].sum(dtype=torch.int64)
                b, m = (int(v) for v in torch.stack((below, mine)).tolist())
                self.r_lo, self.r_hi = b, b + m
            return self.n_alive
        n = self.ops.alive_compact(self.alive, self.alive_idx, self.arank, self.n_alive)
        self.n_alive = n
        if self.world == 1:
            self.r_lo, self.r_hi = 0, n
        else:
            # alive ranks owned by this rank's index range: first alive index >= lo / >= hi
            idx = self.alive_idx[:n]
            bounds = torch.searchsorted(idx, torch.tensor([self.lo, self.hi], dtype=torch.int32, device=self.device))
            self.r_lo, self.r_hi = (int(v) for v in bounds.tolist())
        return n

    def smc_swarm(self, eps: float, gamma0: float, gsig: float, last: bool = False):
        """last: no further sweep follows in this generation (the driver's i == Kmcmc) -- lets a sharded run start
        the per-generation distance exchange early; has no effect on results"""
        self._bind_stamps()
        if self.packed:
            cur = self.buf[self.cur]
            b_in, b_out = self.bits[self.bc], self.bits[1 - self.bc]
            if not self.sharded_packed:
                counts = self.ops.smc_swarm_packed(b_in, b_out, self.n_alive, 0, self.n_alive, self.buf[0][0],
                                                   self.buf[1][0], cur[1], cur[2], None, eps, gamma0, gsig, self.sweep)
                self.sweep += 1
                self.bc = 1 - self.bc
                return counts
            if self._delta_work is not None:      # a caller swept again after announcing the last sweep
                self._delta_work.wait()
                self._delta_work = None
            self._mark("own_sweep", 0)
            self.ops.smc_swarm_packed(b_in, b_out, self.n_alive, self.r_lo, self.r_hi, self.buf[0][0], self.buf[1][0],
                                      cur[1], cur[2], self.flags, eps, gamma0, gsig, self.sweep, want_counts=False)
            self._mark("own_sweep", 1)
            self._mark("flag_allgather", 0)
            self._allgather_chunks(self.flags)               # 1 byte per position: accepted | simulated << 1
            self._mark("flag_allgather", 1)
            if last:
                self._start_delta_allgather()
            self._mark("replay", 0)
            counts = self.ops.smc_replay_packed(b_in, b_out, self.n_alive, self.r_lo, self.r_hi, self.buf[0][0],
                                                self.buf[1][0], cur[1], self.flags, gamma0, gsig, self.sweep)
            self._mark("replay", 1)
            self.sweep += 1
            self.bc = 1 - self.bc
            self._delta_stale = True
            return counts                                    # global (nacc, nsim), counted from the flags
        if self.sharded_rows:
            if self._delta_work is not None:      # a caller swept again after announcing the last sweep: that
                self._delta_work.wait()           # exchange is obsolete (the stale flag stays set)
                self._delta_work = None
            cur = self.buf[self.cur]
            a_in, a_out = self.alive_row[self.ar], self.alive_row[1 - self.ar]
            self._mark("own_sweep", 0)
            self.ops.smc_swarm_rows_shard(a_in, a_out, self.n_alive, self.r_lo, self.r_hi, self.buf[0][0], self.buf[1][0],
                                          cur[1], cur[2], self.accepted, eps, gamma0, gsig, self.sweep, want_counts=False)
            self._mark("own_sweep", 1)
            self._mark("flag_allgather", 0)
            self._allgather_state((self.accepted,))          # 1 byte per particle: accepted | simulated << 1
            self._mark("flag_allgather", 1)
            if last:
                self._start_delta_allgather()
            self._mark("replay", 0)
            counts = self.ops.smc_replay_rows(a_in, a_out, self.n_alive, self.r_lo, self.r_hi, self.buf[0][0],
                                              self.buf[1][0], self.accepted, gamma0, gsig, self.sweep)
            self._mark("replay", 1)
            self.sweep += 1
            self.ar = 1 - self.ar
            self._rows_dirty = True
            self._delta_stale = self._logpi_stale = True
            return counts                                    # global (nacc, nsim), counted from the flags
        if self.rows_mode:
            cur = self.buf[self.cur]
            nacc, nsim = self.ops.smc_swarm_rows(self.alive_row[self.ar], self.alive_row[1 - self.ar], self.n_alive,
                                                 self.buf[0][0], self.buf[1][0], cur[1], cur[2], eps, gamma0, gsig,
                                                 self.sweep)
            self.sweep += 1
            self.ar = 1 - self.ar
            self._rows_dirty = True
            return nacc, nsim
        copy_dead = (not self._dead_carried) and self.n_alive < self.N
        nacc, nsim = self.ops.smc_swarm(self.alive_idx, self.arank, self.n_alive, self.r_lo, self.r_hi, self.state,
                                        self.other, eps, gamma0, gsig, self.lo, self.n_local, copy_dead, self.sweep,
                                        self.row_synced)
        self.sweep += 1
        self._dead_carried = True
        self._allgather_state(self.other + ((self.stamp[1 - self.cur],) if self.blob_on else ()))
        self._swap()
        return self._allreduce_counts(nacc, nsim)

    # ------------------------------------------------------------------ S4
    def _mc_arrays(self):
        """the enumeration of mc:23: order (positions -> particles), sorted_delta, and every particle's candidate count"""
        if self.order is None:
            self.order = torch.zeros(self.N, dtype=torch.int32, device=self.device)
            self.sorted_delta = torch.zeros(self.N, dtype=torch.float64, device=self.device)
            self.rank_cnt = torch.zeros(self.N, dtype=torch.int32, device=self.device)

    def mc_rank_prepare(self, eps_pop: float = None, dmax_hint: float = None):
        """the enumeration of mc:23 for the current distances: particles with Ds <= eps_pop in index order, then the
        others by (Ds, index).  eps_pop / dmax_hint default to the values of mc:146-147 with eps_target = 0 (one
        extrema pass); the driver passes what it already knows.  No host synchronisation on the HIP path."""
        if self.rows_mode or self.packed:
            raise RuntimeError("abcdemc needs storage='classic'")
        self._mc_arrays()
        if eps_pop is None or dmax_hint is None:
            lo, hi = self.extrema()
            eps_pop = lo if eps_pop is None else eps_pop
            dmax_hint = hi if dmax_hint is None else dmax_hint
        self.ops.mc_rank_prepare(self.state[2], eps_pop, dmax_hint, self.order, self.sorted_delta, self.rank_cnt)

    def mc_swarm(self, eps_pop: float, eps_target: float, gamma0: float, gsig: float):
        """one sweep of abcdemc_swarm! -> (nsim, #(Ds > eps_target), min Ds, max Ds) of the generation it leaves
        (mc:149,156,146,163), global over all ranks"""
        self._mc_arrays()        # (a converged population never consults them)
        self._bind_stamps()
        nsim, ngt, lo, hi = self.ops.mc_swarm(self.order, self.rank_cnt, self.state, self.other, eps_pop, eps_target,
                                              gamma0, gsig, self.lo, self.n_local, self.sweep)
        self.sweep += 1
        self._allgather_state(self.other + ((self.stamp[1 - self.cur],) if self.blob_on else ()))
        self._swap()
        nsim, ngt = self._allreduce_counts(nsim, ngt)
        if self._collectives:
            import torch.distributed as dist

            t = torch.tensor([lo, -hi], dtype=torch.float64, device=self.device if self._backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.pg)
            lo, hi = float(t[0]), -float(t[1])
        return nsim, ngt, lo, hi

    def mc_generation(self, eps_pop: float, eps_target: float, dmax: float, gamma0: float, gsig: float):
        """the body of abcdemc!'s loop (mc:140-156): rank pass while some Ds > eps_target, one sweep -> (nsim,
        #(Ds > eps_target), min Ds, max Ds) of the new generation.  One library call on a single GPU."""
        if not self._collectives and hasattr(self.ops, "mc_generation") and not self.rows_mode and not self.packed:
            self._mc_arrays()
            self._bind_stamps()
            out = self.ops.mc_generation(self.state, self.other, self.order, self.sorted_delta, self.rank_cnt, eps_pop,
                                         eps_target, dmax, gamma0, gsig, self.sweep)
            self.sweep += 1
            self._swap()
            return out
        if dmax > eps_target:
            self.mc_rank_prepare(eps_pop, dmax)
        return self.mc_swarm(eps_pop, eps_target, gamma0, gsig)

    # ------------------------------------------------------------------ checkpoint / resume
    def download_state(self) -> dict:
        """Everything a later run needs to continue this one bit for bit: the population arrays the reference
        driver owns (thetas, logpi, Ds, Wns, alive; smc:242-270) and the two RNG epochs.  Host numpy arrays."""
        th, lp, dl = self.state
        extra = {"stamp": self.stamp[self.cur].cpu().numpy().copy()} if self.blob_on else {}
        return {
            **extra,
            "theta": th.cpu().numpy().copy(), "logpi": lp.cpu().numpy().copy(), "delta": dl.cpu().numpy().copy(),
            "wns": self.wns.cpu().numpy().copy(), "alive": self.alive.cpu().numpy().copy(),
            "sweep": int(self.sweep), "draw": int(self.draw), "N": self.N, "ld": int(th.shape[1]),
            "seed": int(self.spec.seed),
        }

    def upload_state(self, st: dict):
        """Inverse of :meth:`download_state` (every rank uploads the full arrays)."""
        if self._delta_work is not None:
            self._delta_work.wait()
            self._delta_work = None
        th = torch.as_tensor(np.ascontiguousarray(st["theta"], dtype=np.float64))
        if tuple(th.shape) != tuple(self.buf[0][0].shape):
            raise ValueError(f"checkpoint holds a population of shape {tuple(th.shape)}, "
                             f"the engine expects {tuple(self.buf[0][0].shape)}")
        if int(st.get("seed", self.spec.seed)) != int(self.spec.seed):
            raise ValueError("checkpoint was written with a different seed: the run would not continue its own stream")
        self.cur = 0
        cur = self.buf[0]
        cur[0].copy_(th)
        cur[1].copy_(torch.as_tensor(np.ascontiguousarray(st["logpi"], dtype=np.float64)))
        cur[2].copy_(torch.as_tensor(np.ascontiguousarray(st["delta"], dtype=np.float64)))
        self.wns.copy_(torch.as_tensor(np.ascontiguousarray(st["wns"], dtype=np.float64)))
        self.alive.copy_(torch.as_tensor(np.ascontiguousarray(st["alive"], dtype=np.uint8)))
        self.n_alive = int(self.alive.sum().item())
        self.sweep, self.draw = int(st["sweep"]), int(st["draw"])
        if self.blob_on:
            if "stamp" not in st:
                raise ValueError("checkpoint was written without blobs")
            self.stamp[0].copy_(torch.as_tensor(np.ascontiguousarray(st["stamp"], dtype=np.int64)))
        if self.packed:                          # every position's current row is slot 0 again
            for b in self.bits:
                b.zero_()
            self.n_prev = self.n_alive
            if not bool(self.alive[:self.n_alive].all()):
                raise ValueError("checkpoint of a packed population must have its alive particles in front")
        if self.rows_mode:                       # every particle's current row is slot 0 again
            self.cur_row.copy_(torch.arange(self.N, dtype=torch.int32, device=self.device))
            self._rows_dirty = False
            self._rows_n = 0
        self._delta_stale = self._logpi_stale = False
        self._dead_carried = False
        self.row_synced.zero_()

    # ------------------------------------------------------------------ results (smc:382-393, mc:166-171)
    def result(self):
        th, lp, dl = self.state
        pushed = torch.empty_like(th)
        self.ops.push_p(th, pushed)
        d = self.spec.d
        P = pushed[:, :d].cpu().numpy()
        if d == 1 and not hasattr(self.spec.prior, "p"):
            P = P[:, 0]                      # univariate prior -> vector of scalars
        blobs = None
        if self.blob_on:
            # rebuild the simulated data behind every particle's distance from its stamp; the re-run's distance
            # must be the stored one, bit for bit
            nb = self.spec.n_blob
            out = torch.zeros((self.N, self.ops.blob_width()), dtype=torch.float64, device=self.device)
            redo = torch.empty_like(dl)
            self.ops.blob_eval(th, self.stamp[self.cur], out, redo)
            if not torch.equal(redo.view(torch.int64), dl.view(torch.int64)):
                raise RuntimeError("blobs: a re-run simulation does not reproduce the stored distance")
            blobs = out[:, :nb].cpu().numpy()
            if nb == 1:
                blobs = blobs[:, 0]
        return {
            "blobs": blobs,
            "P": P,
            "theta": th[:, :d].cpu().numpy(),  # internal (unrounded) state, mc:216-220
            "logpi": lp.cpu().numpy(),
            "C": dl.cpu().numpy(),
            "Wns": self.wns.cpu().numpy(),
            "alive": self.alive.cpu().numpy().astype(bool),
        }


def HipEngine(spec: ModelSpec, nparticles: int, process_group=None, lanes: int = 0,
              storage: str = "packed", force_collectives: bool = False) -> PopulationEngine:
    """The product engine: HIP kernels on the current CUDA(HIP) device.  Row-store sweeps (only accepted
    proposals are written); sharded runs keep one replica per GPU and exchange accept flags (module docstring)."""
    return PopulationEngine(spec, nparticles, process_group, ops=None, lanes=lanes, storage=storage,
                            force_collectives=force_collectives)
