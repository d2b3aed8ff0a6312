"""Population engine: device-resident particle state + one call per reference function.

``PopulationEngine`` owns the arrays the reference driver owns
(``θs, logπ, Δs, Wns, alive`` and their ``n*`` doubles, src/abcdez_smc.jl:242-275)
as torch tensors, shards the work over the ranks of a ``torch.distributed``
process group, and forwards every population-sized step to an *ops* backend:

* :class:`HipOps` -- the product: ctypes calls into ``libabcdez_hip.so`` with raw
  device pointers (torch is plumbing: memory, streams, collectives);
* the test suite injects an oracle-backed ops object to exercise this file's host
  logic (sharding, collectives, buffer swapping) on CPU over ``gloo``.

Two storages, one per driver (every array has FULL length N on every rank: a donor can be any particle):

* ``storage="packed"`` -- ``abcdesmc``.  Two row slots per position (``buf[0][0]``, ``buf[1][0]``, each
  ``f64[N][ld]`` row-major), one bit per position naming the current slot (``bits``, ping-pong), ``logpi`` /
  ``delta`` updated in place (they ping-pong only at resamplings), ``wns``, ``alive``.  After every reweight the
  population is PARTITIONED so that the alive particles are the positions ``[0, n_alive)`` (include/abcdez_hip.h):
  no alive list, donors addressed directly, reweight / quantile touch the prefix only.
  Multi-GPU: every rank keeps the whole population; rank r sweeps the chunk
  ``[r c, (r+1) c)`` of the prefix (c = a multiple of 64 covering n_alive / G), the ranks exchange ONE BYTE per
  position and sweep -- the accept flag -- and rebuild the other ranks' accepted proposals and their log-priors
  themselves (``smc_replay_packed``: a proposal is a function of replicated rows and of counter-based random
  numbers).  Distances are all-gathered once per generation, overlapped with the last replay.  Random numbers
  are keyed by position, so results do not depend on G.
* ``storage="classic"`` -- ``abcdemc``.  ``buf[t]`` / ``buf[1-t]`` are generations t and t+1 (the reference's
  ``θs`` / ``nθs``, src/abcdez_mc.jl:140-155); rank r updates the particles ``[r N/G, (r+1) N/G)`` and the new rows /
  logπ / Δ are exchanged with one in-place all-gather each after every sweep.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._trace import rng as _rng
from .model import ModelSpec

PACKED_ALIGN = 64        # sub-ranges of the packed prefix start / end at multiples of this (include/abcdez_hip.h)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def pg_host_transport(pg, with_allreduce: bool = False):
    """The callbacks of abcdez_comm_init_host over a torch.distributed process group whose backend moves HOST memory (gloo; an
    MPI host would pass MPI.Allgather! here): ``allgather(address, piece_bytes)`` in place on the library's page-locked block,
    and -- optionally -- ``allreduce(address, n, dtype, op)`` on 8-byte words (dtype 0 int64, 1 float64, 2 uint64; op 0 sum,
    1 min, 2 max).  Without the second the library reduces in rank order itself (include/abcdez_hip.h)."""
    import torch.distributed as dist

    rank, world = dist.get_rank(pg), dist.get_world_size(pg)

    def view(address, nbytes, dtype):
        return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(address))).view(dtype)

    def allgather(address, piece):
        t = view(address, piece * world, torch.uint8)
        dist.all_gather_into_tensor(t, t[rank * piece:(rank + 1) * piece], group=pg)

    def allreduce(address, n, dtype, op):
        rop = (dist.ReduceOp.SUM, dist.ReduceOp.MIN, dist.ReduceOp.MAX)[op]
        if dtype == 1:
            dist.all_reduce(view(address, 8 * n, torch.float64), op=rop, group=pg)
            return
        t = view(address, 8 * n, torch.int64)
        if dtype == 2 and op != 0:          # unsigned order through signed words: flip the top bit, reduce, flip it back
            t ^= -0x8000000000000000
            dist.all_reduce(t, op=rop, group=pg)
            t ^= -0x8000000000000000
        else:                               # (a sum wraps the same way in both)
            dist.all_reduce(t, op=rop, group=pg)

    return allgather, (allreduce if with_allreduce else None)


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

    def set_uniform_weights(self, on: bool):
        """the alive particles' weights are 1 / n_alive (the host wrote them so): with an indicator kernel the prologue uses the
        closed forms of the reweight (include/abcdez_hip.h); the library keeps the flag itself afterwards"""
        _lib.check(self.lib, self.lib.abcdez_ctx_set_uniform_weights(self.ctx, 1 if on else 0))

    def get_uniform_weights(self) -> bool:
        on = C.c_int32()
        _lib.check(self.lib, self.lib.abcdez_ctx_get_uniform_weights(self.ctx, C.byref(on), None))
        return bool(on.value)

    def fast_prologues(self) -> int:
        on, n = C.c_int32(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_ctx_get_uniform_weights(self.ctx, C.byref(on), C.byref(n)))
        return n.value

    def set_graphs(self, on: bool):
        """replay abcdemc generations as HIP graphs (default off: measured slower than stream launches; results do not depend on it)"""
        _lib.check(self.lib, self.lib.abcdez_ctx_set_graphs(self.ctx, 1 if on else 0))

    def graph_stats(self):
        """(generations replayed from a graph, graphs captured, generations enqueued launch by launch)"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_graph_stats(self.ctx, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_timing(self, on):
        """True / 1: every sweep launch; 2: the first sweep of every grouped call; False: off"""
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

    def smc_generation_packed(self, n_prev, bits_cur, bits_oth, slot0, slot1, cur, oth, wns, alive, inds, alpha, eps_prev, eps_target,
                              eps_k, ess_min, gamma0, gsig, sweep0, draw, k_max, kmcmc_min, select_ahead):
        """one generation of smc:301-353 in ONE library call (abcdez_smc_generation_packed): prologue, the resampling when
        ESS < ess_min, the sweeps.  cur / oth = the (logpi, delta) pairs -> dict of everything the driver needs"""
        eps, wnorm, ess, essr, lo, hi = (C.c_double() for _ in range(6))
        na, ns = C.c_int64(), C.c_int64()
        part, res, done = C.c_int32(), C.c_int32(), C.c_int32()
        nacc, nsim = (C.c_int64 * k_max)(), (C.c_int64 * k_max)()
        _lib.check(self.lib, self.lib.abcdez_smc_generation_packed(
            self.ctx, wns.numel(), n_prev, _ptr(bits_cur), _ptr(bits_oth), _ptr(slot0), _ptr(slot1), _ptr(cur[0]), _ptr(cur[1]),
            _ptr(oth[0]), _ptr(oth[1]), _ptr(wns), _ptr(alive), _ptr(inds), alpha, eps_prev, eps_target, eps_k, ess_min, gamma0, gsig,
            sweep0, draw, k_max, kmcmc_min, 1 if select_ahead else 0, C.byref(eps), C.byref(wnorm), C.byref(ess), C.byref(na),
            C.byref(part), C.byref(res), C.byref(essr), C.byref(ns), nacc, nsim, C.byref(done), C.byref(lo), C.byref(hi)))
        return dict(eps=eps.value, wnorm=wnorm.value, ess=ess.value, n_alive=na.value, partitioned=bool(part.value),
                    resampled=bool(res.value), ess_resampled=essr.value, n_swept=ns.value, naccs=list(nacc[:done.value]),
                    nsims=list(nsim[:done.value]), Ki=done.value, range=(lo.value, hi.value))

    def smc_swarm_packed(self, bits, bits_out, n_alive, r_lo, r_hi, slot0, slot1, logpi, delta, flags, eps, gamma0, gsig,
                         sweep, want_counts=True):
        """want_counts=False: no host synchronisation (the replay reports the sweep's global counters)"""
        nacc, nsim = C.c_int64(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_smc_swarm_packed(
            self.ctx, _ptr(bits), _ptr(bits_out), n_alive, r_lo, r_hi, _ptr(slot0), _ptr(slot1), _ptr(logpi), _ptr(delta),
            _ptr(flags), eps, gamma0, gsig, sweep, C.byref(nacc) if want_counts else None,
            C.byref(nsim) if want_counts else None))
        return (nacc.value, nsim.value) if want_counts else None

    SWEEPS_MAX = 16

    def smc_sweeps_packed(self, bits_a, bits_b, n_alive, slot0, slot1, logpi, delta, eps, gamma0, gsig, sweep0, k_max, kmcmc_min):
        """up to k_max sweeps behind the device-side test of smc:352 -> (naccs per sweep, nsims per sweep, Ki): ONE host sync"""
        nacc, nsim, done = (C.c_int64 * k_max)(), (C.c_int64 * k_max)(), C.c_int32()
        _lib.check(self.lib, self.lib.abcdez_smc_sweeps_packed(
            self.ctx, _ptr(bits_a), _ptr(bits_b), n_alive, _ptr(slot0), _ptr(slot1), _ptr(logpi), _ptr(delta), eps, gamma0,
            gsig, sweep0, k_max, kmcmc_min, nacc, nsim, C.byref(done)))
        return list(nacc[:done.value]), list(nsim[:done.value]), done.value

    def smc_select_ahead(self, delta, alive, alpha, eps_target):
        """arm the next grouped sweeps: they also start the next generation's select (include/abcdez_hip.h)"""
        _lib.check(self.lib, self.lib.abcdez_smc_select_ahead(self.ctx, _ptr(delta), _ptr(alive), delta.numel(), alpha, eps_target))

    # the sweeps of a generation on a sharded population, one host synchronisation (include/abcdez_hip.h)
    def smc_group_begin(self, n_alive, kmcmc_min):
        _lib.check(self.lib, self.lib.abcdez_smc_group_begin(self.ctx, n_alive, kmcmc_min))

    def smc_group_replay(self, bits, bits_out, skip_lo, skip_hi, slot0, slot1, logpi, flags, gamma0, gsig, sweep):
        _lib.check(self.lib, self.lib.abcdez_smc_group_replay(self.ctx, _ptr(bits), _ptr(bits_out), skip_lo, skip_hi, _ptr(slot0),
                                                              _ptr(slot1), _ptr(logpi), _ptr(flags), gamma0, gsig, sweep))

    def smc_group_publish(self):
        _lib.check(self.lib, self.lib.abcdez_smc_group_publish(self.ctx))

    def smc_group_abort(self):
        """abandon an open group after a failure between begin and end (no-op when none is open)"""
        _lib.check(self.lib, self.lib.abcdez_smc_group_abort(self.ctx))

    def stream_version(self):
        """(library version, Philox rounds): what every random number of a run depends on -- stored in checkpoints"""
        return int(self.lib.abcdez_version()), int(self.lib.abcdez_rng_rounds())

    def smc_group_end(self, k_max):
        nacc, nsim, done = (C.c_int64 * k_max)(), (C.c_int64 * k_max)(), C.c_int32()
        _lib.check(self.lib, self.lib.abcdez_smc_group_end(self.ctx, nacc, nsim, C.byref(done)))
        return list(nacc[:done.value]), list(nsim[:done.value]), done.value

    # ---- multi-GPU: RCCL behind the C ABI (include/abcdez_hip.h, csrc/abz_comm.hip) ----
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        _lib.check(self.lib, self.lib.abcdez_comm_unique_id(buf, 128))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        _lib.check(self.lib, self.lib.abcdez_comm_init(self.ctx, unique_id, len(unique_id), rank, world))

    def comm_init_host(self, rank: int, world: int, allgather, allreduce=None):
        """the host-supplied transport (abcdez_comm_init_host): ``allgather(address, piece_bytes)`` gathers in place on HOST memory
        (this rank's piece at address + rank * piece_bytes), ``allreduce(address, n, dtype, op)`` reduces n 8-byte words in place
        (optional: without it the library all-gathers the words and reduces them in rank order).  Python callables; an
        exception inside one is reported as a failed collective."""
        def guard(fn, what):
            def call(user, *args):
                try:
                    fn(*args)
                    return 0
                except BaseException as e:          # nothing may propagate through the C frames
                    import sys
                    print(f"[abcdez host transport] {what} failed: {e!r}", file=sys.stderr)
                    return 1
            return call
        # the library keeps the function pointers for the life of the communicator: so must this object
        self._host_cb = (_lib.HOST_ALLGATHER_FN(guard(allgather, "all-gather")),
                         _lib.HOST_ALLREDUCE_FN(guard(allreduce, "all-reduce")) if allreduce is not None else None)
        _lib.check(self.lib, self.lib.abcdez_comm_init_host(self.ctx, rank, world, C.cast(self._host_cb[0], C.c_void_p),
                                                            C.cast(self._host_cb[1], C.c_void_p) if self._host_cb[1] else None, None))

    def comm_kind(self) -> int:
        """0: no communicator, 1: RCCL, 2: host transport"""
        k = C.c_int32()
        _lib.check(self.lib, self.lib.abcdez_comm_kind(self.ctx, C.byref(k)))
        return k.value

    def comm_destroy(self):
        _lib.check(self.lib, self.lib.abcdez_comm_destroy(self.ctx))
        self._host_cb = None

    def comm_rank(self):
        r, w = C.c_int32(), C.c_int32()
        rc = self.lib.abcdez_comm_rank(self.ctx, C.byref(r), C.byref(w))
        if rc < 0:
            _lib.check(self.lib, rc)
        return r.value, w.value, rc == 0

    def comm_allgather(self, t, piece_elems: int):
        """in place over `world` pieces of piece_elems elements of t (rank r's piece at t[r * piece_elems]); on the library's stream"""
        _lib.check(self.lib, self.lib.abcdez_comm_allgather(self.ctx, _ptr(t), piece_elems * t.element_size()))

    def comm_allreduce(self, t, op: str = "sum"):
        dt = {torch.int64: 0, torch.float64: 1, torch.uint64: 2}[t.dtype]
        _lib.check(self.lib, self.lib.abcdez_comm_allreduce(self.ctx, _ptr(t), t.numel(), dt, {"sum": 0, "min": 1, "max": 2}[op]))

    def smc_sweeps_sharded(self, bits_a, bits_b, n_alive, chunk, slot0, slot1, logpi, delta, flags, eps, gamma0, gsig, sweep0,
                           k_max, kmcmc_min):
        """the sweeps of one generation on a sharded population, collectives included: ONE call, ONE host sync"""
        nacc, nsim, done = (C.c_int64 * k_max)(), (C.c_int64 * k_max)(), C.c_int32()
        _lib.check(self.lib, self.lib.abcdez_smc_sweeps_sharded(
            self.ctx, _ptr(bits_a), _ptr(bits_b), n_alive, chunk, _ptr(slot0), _ptr(slot1), _ptr(logpi), _ptr(delta), _ptr(flags),
            eps, gamma0, gsig, sweep0, k_max, kmcmc_min, nacc, nsim, C.byref(done)))
        return list(nacc[:done.value]), list(nsim[:done.value]), done.value

    def smc_select_discard(self):
        """forget a select armed / enqueued ahead (the population was written by other means, or the run ends)"""
        _lib.check(self.lib, self.lib.abcdez_smc_select_discard(self.ctx))

    def mc_rank_stats(self):
        """rank passes that launched (both sorts, only the LDS sort, only the radix sort)"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_mc_rank_stats(self.ctx, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def smc_select_stats(self):
        """(prologues that found their select enqueued ahead, prologues that ran it themselves)"""
        a, b = C.c_int64(), C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_smc_select_stats(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

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

    def mc_generation(self, cur, nxt, order, sorted_delta, cnt, eps_pop, eps_target, dmax, n_above, gamma0, gsig, sweep):
        """rank pass (if dmax > eps_target and fewer than 1 / 16 of the particles lie at or below eps_target: n_above above it, or -1 = count
        them here) + sweep over all particles: one library call, one host sync"""
        nsim, ngt, lo, hi = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(self.lib, self.lib.abcdez_mc_generation(
            self.ctx, cur[1].numel(), _ptr(cur[0]), _ptr(cur[1]), _ptr(cur[2]), _ptr(nxt[0]), _ptr(nxt[1]), _ptr(nxt[2]),
            _ptr(order), _ptr(sorted_delta), _ptr(cnt), eps_pop, eps_target, dmax, n_above, gamma0, gsig, sweep, C.byref(nsim),
            C.byref(ngt), C.byref(lo), C.byref(hi)))
        return nsim.value, ngt.value, lo.value, hi.value

    def mc_draws_by_rejection(self, n_above: int, N: int) -> bool:
        r = self.lib.abcdez_mc_draws_by_rejection(n_above, N)
        if r < 0:
            raise ValueError("mc_draws_by_rejection: need 0 <= n_above <= N, N >= 1")
        return bool(r)

    def mc_draw_stats(self) -> int:
        """asynchronous generations that needed no rank pass because they draw their better particles by rejection"""
        n = C.c_int64()
        _lib.check(self.lib, self.lib.abcdez_mc_draw_stats(self.ctx, C.byref(n)))
        return n.value

    MC_IN_FLIGHT = 8

    def mc_generation_async(self, cur, nxt, order, sorted_delta, cnt, alpha, eps_target, lo_hi, do_rank, gamma0, gsig, sweep):
        """the same without a host synchronisation (extrema / eps_pop of mc:146-147 stay on the device) -> ticket"""
        t = C.c_int64()
        lh = (C.c_double * 2)(*lo_hi) if lo_hi is not None else None
        _lib.check(self.lib, self.lib.abcdez_mc_generation_async(
            self.ctx, cur[1].numel(), _ptr(cur[0]), _ptr(cur[1]), _ptr(cur[2]), _ptr(nxt[0]), _ptr(nxt[1]), _ptr(nxt[2]),
            _ptr(order), _ptr(sorted_delta), _ptr(cnt), alpha, eps_target, lh, 1 if do_rank else 0, gamma0, gsig, sweep,
            C.byref(t)))
        return t.value

    def mc_generation_sharded_async(self, cur, nxt, order, sorted_delta, cnt, alpha, eps_target, lo_hi, do_rank, gamma0, gsig, sweep):
        """the same on a population sharded over the ranks of the library's communicator: own range swept, rows / reductions
        exchanged by the library, no host synchronisation -> ticket (global results from mc_generation_wait)"""
        t = C.c_int64()
        lh = (C.c_double * 2)(*lo_hi) if lo_hi is not None else None
        _lib.check(self.lib, self.lib.abcdez_mc_generation_sharded_async(
            self.ctx, cur[1].numel(), _ptr(cur[0]), _ptr(cur[1]), _ptr(cur[2]), _ptr(nxt[0]), _ptr(nxt[1]), _ptr(nxt[2]),
            _ptr(order), _ptr(sorted_delta), _ptr(cnt), alpha, eps_target, lh, 1 if do_rank else 0, gamma0, gsig, sweep,
            C.byref(t)))
        return t.value

    def mc_generation_wait(self, ticket):
        """-> (nsim, #(Ds > eps_target), min Ds, max Ds, eps_pop) of generation `ticket`; waits for that generation only"""
        nsim, ngt, lo, hi, ep = C.c_int64(), C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        _lib.check(self.lib, self.lib.abcdez_mc_generation_wait(self.ctx, ticket, C.byref(nsim), C.byref(ngt), C.byref(lo),
                                                                C.byref(hi), C.byref(ep)))
        return nsim.value, ngt.value, lo.value, hi.value, ep.value

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
                 storage: str = "packed", force_collectives: bool = False):
        """storage = "packed": abcdesmc (module docstring); storage = "classic": the double buffer of abcdemc.
        force_collectives: take the sharded code path (collectives, chunk sweep + replay) even in a group of ONE
        rank -- lets a single-GPU box exercise the RCCL calls (tests)."""
        if storage not in ("packed", "classic"):
            raise ValueError("storage must be 'packed' (abcdesmc) or 'classic' (abcdemc)")
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
        # Collectives of the HIP ops are issued BY THE LIBRARY (abcdez_comm_*, csrc/abz_comm.hip): RCCL on the library's own stream
        # when the group's backend is RCCL (the rendezvous id travels through the process group once); for any other backend (gloo:
        # ranks that share one GPU, hosts without RCCL) the library's host transport with this group's all-gather underneath
        # (abcdez_comm_init_host; ABZ_HOST_ALLREDUCE=1 also hands it the group's all-reduce instead of the library's rank-order
        # reduction).  torch.distributed's collectives are called by this file only for ops without a communicator of their own
        # (the CPU oracle of the gloo tests).
        self._native_comm = False
        if self.pg is not None and hasattr(self.ops, "comm_init") and (self.world > 1 or force_collectives):
            import torch.distributed as dist

            if self._backend == "nccl":
                idt = torch.zeros(128, dtype=torch.uint8, device=self.ops.device)
                if self.rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(self.ops.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(idt, src=dist.get_global_rank(self.pg, 0), group=self.pg)
                self.ops.comm_init(bytes(idt.cpu().numpy().tobytes()), self.rank, self.world)
            else:
                ag, ar = pg_host_transport(self.pg, with_allreduce=os.environ.get("ABZ_HOST_ALLREDUCE", "0") == "1")
                self.ops.comm_init_host(self.rank, self.world, ag, ar)
            self._native_comm = True
        dev = self.ops.device
        self.device = dev
        N, ld = self.N, spec.ld
        f64 = dict(dtype=torch.float64, device=dev)
        self.packed = storage == "packed"
        self._collectives = self.world > 1 or (force_collectives and self.pg is not None)
        self.sharded_packed = self.packed and self._collectives
        # (theta, logpi, delta) x 2.  classic: generations t and t+1.  packed: the theta arrays are the two slots of the
        # store, (logpi, delta) ping-pong at resamplings only.  Sharded packed runs exchange [r c, (r+1) c) pieces of
        # the per-position arrays: room for G chunks.
        self._npad = N + (self.world * PACKED_ALIGN if self.sharded_packed else 0)
        rows = [torch.zeros((N, ld), **f64) for _ in range(2)]
        self._full = [(torch.zeros(self._npad, **f64), torch.zeros(self._npad, **f64)) for _ in range(2)]
        self.buf = [(rows[k], self._full[k][0][:N], self._full[k][1][:N]) for k in range(2)]
        self.cur = 0
        if self.packed:
            nw = (N + 31) // 32
            self.bits = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(2)]   # current slot of every position
            self.bc = 0
            self.n_prev = N                 # length of the alive prefix
            self.flags = torch.zeros(self._npad, dtype=torch.uint8, device=dev) if self.sharded_packed else None
            self.chunk = 0
        # blobs (spec.n_blob > 0): one stamp per particle, travelling with (logpi, delta)
        self.blob_on = getattr(spec, "n_blob", 0) > 0
        self.stamp = [torch.zeros(N, dtype=torch.int64, device=dev) for _ in range(2)] if self.blob_on else None
        self._delta_work = None      # sharded packed: distance all-gather in flight (overlapped with the last replay)
        self._prof = None            # optional event timing of the sharded sweep's phases (enable_phase_timing)
        self._delta_stale = False    # sharded packed: the other ranks' distances not yet fetched
        self.wns = torch.full((N,), 1.0 / N, **f64)
        self.alive = torch.ones(N, dtype=torch.uint8, device=dev)
        self.inds = torch.zeros(N, dtype=torch.int32, device=dev)
        self.order = None
        self.sorted_delta = None
        self.rank_cnt = None
        self.n_alive = N
        self.r_lo, self.r_hi = 0, N
        self.sweep = 0          # global sweep number = RNG epoch of the swarm kernels
        self.draw = 0           # resampling number = RNG epoch of the stratified draws

    def run_scope(self):
        """Context manager for a whole run: work on a side stream of this engine when the caller's current stream is the
        legacy default stream (which cannot be captured into a graph and serialises with every other blocking stream); the
        side stream first waits for the caller's stream and the caller's stream waits for it at exit, so code around the
        run sees ordinary stream order.  A caller that already runs on a stream of its own keeps it."""
        import contextlib

        if self.device.type != "cuda":
            return contextlib.nullcontext()
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != 0:
            return contextlib.nullcontext()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(self.device)

        @contextlib.contextmanager
        def scope():
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                try:
                    yield
                finally:
                    cur.wait_stream(self._side)
        return scope()

    def _stream(self):
        """the library launches on the stream that is current NOW: torch ops and collectives issued by this engine
        between library calls go to torch's current stream, so the two must be the same one at every call"""
        if hasattr(self.ops, "use_current_stream"):
            self.ops.use_current_stream()

    # ------------------------------------------------------------------ helpers
    @property
    def state(self):
        """(theta, logpi, delta) of the current generation.  Packed storage gathers the current rows into a fresh
        dense array (tests / results only; the hot path never calls this)."""
        if not self.packed:
            return self.buf[self.cur]
        self._stream()
        self._sync_delta()
        th = torch.empty_like(self.buf[0][0])
        self.ops.packed_gather(self.bits[self.bc], self.buf[0][0], self.buf[1][0], th)
        return (th, self.buf[self.cur][1], self.buf[self.cur][2])

    @property
    def delta(self):
        self._sync_delta()
        return self.buf[self.cur][2]

    def _sync_delta(self):
        """sharded packed: fetch the other ranks' distances (once per generation, before the first consumer)"""
        if self._delta_work is not None:          # started behind the generation's last replay: just join it
            self._mark("delta_allgather_wait", 0)
            with _rng("delta_allgather_wait"):
                self._delta_work.wait()
            self._delta_work = None
            self._mark("delta_allgather_wait", 1)
            self._delta_stale = False
        if self._delta_stale:
            self._mark("delta_allgather", 0)
            with _rng("delta_allgather"):
                self._allgather_chunks(self._full[self.cur][1])
            self._mark("delta_allgather", 1)
            self._delta_stale = False

    def _start_delta_allgather(self):
        """sharded packed, last sweep of a generation: the owners' distances are final once the chunk sweep has run,
        so their all-gather goes out on the collective stream while this rank replays the other chunks."""
        import torch.distributed as dist

        if self._native_comm:          # same stream as the kernels: ordered behind the sweep that made the distances final
            self.ops.comm_allgather(self._full[self.cur][1], self.chunk)
            self._delta_stale = False
            return True
        t = self._full[self.cur][1]    # (CPU tensors of an ops without a communicator: the oracle under gloo)
        self._delta_work = dist.all_gather_into_tensor(t[:self.world * self.chunk],
                                                       t[self.rank * self.chunk:(self.rank + 1) * self.chunk],
                                                       group=self.pg, async_op=True)

    # ---- optional phase timing of the sharded packed sweep (bench.py's multi-GPU breakdown) ----
    def enable_phase_timing(self):
        """record device events around the phases of every sharded sweep: own chunk sweep, flag all-gather, replay,
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

    def _bind_stamps(self):
        """blobs: tell the ops which stamp arrays belong to the current / next generation"""
        if self.blob_on:
            self.ops.set_stamps(self.stamp[self.cur], self.stamp[1 - self.cur])
        else:
            self.ops.set_stamps(None, None)

    def alive_indices(self) -> torch.Tensor:
        """positions of the alive particles (int64): the prefix of the packed population"""
        return torch.arange(self.n_prev, dtype=torch.int64, device=self.device)

    @property
    def other(self):
        return self.buf[1 - self.cur]

    def _swap(self):
        self.cur = 1 - self.cur

    def _allgather_state(self, bufs):
        """classic storage: exchange entries [lo, hi) of per-particle arrays (theta, logpi, delta, ...) -- one in-place
        all-gather per array: by the library for the HIP ops (RCCL over xGMI, or its host transport), by torch.distributed on
        the CPU tensors of an ops without a communicator (the oracle under gloo)."""
        if not self._collectives:
            return
        import torch.distributed as dist

        for t in bufs:
            if self._native_comm:
                self.ops.comm_allgather(t, t.numel() // self.world)
            else:
                dist.all_gather_into_tensor(t, t[self.lo:self.hi], group=self.pg)

    def _allgather_chunks(self, t):
        """sharded packed runs: rank r owns the positions [r chunk, (r + 1) chunk) of the prefix (chunk = a multiple of
        PACKED_ALIGN covering n_alive / G); exchange those pieces of a per-position array of length >= G chunk"""
        import torch.distributed as dist

        c, g = self.chunk, self.world
        if self._native_comm:
            self.ops.comm_allgather(t, c)
        else:
            dist.all_gather_into_tensor(t[:g * c], t[self.rank * c:(self.rank + 1) * c], group=self.pg)

    def _allreduce_counts(self, *vals):
        if not self._collectives:
            return vals
        import torch.distributed as dist

        t = torch.tensor(vals, dtype=torch.int64, device=self.device if self._native_comm else "cpu")
        if self._native_comm:
            self._stream()
            self.ops.comm_allreduce(t, "sum")
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return tuple(int(v) for v in t.tolist())

    def _need_packed(self, what):
        if not self.packed:
            raise RuntimeError(f"{what} needs storage='packed' (abcdesmc); this engine holds abcdemc's double buffer")

    # ------------------------------------------------------------------ S1
    def discard_select_ahead(self):
        """A select enqueued ahead of its prologue describes the distances / flags as the library last wrote them: every
        write from the host side (torch copies below) and the end of a run invalidate it."""
        if hasattr(self.ops, "smc_select_discard"):
            self.ops.smc_select_discard()

    def init_population(self):
        self._stream()
        self.discard_select_ahead()
        self._mc_sharded_chain = False
        if self._delta_work is not None:
            self._delta_work.wait()
            self._delta_work = None
        if self.packed:          # a fresh population lives in slot `cur` of every position (also when an engine is re-used)
            for b in self.bits:
                b.fill_(-1 if self.cur else 0)
            self.n_prev = self.N
        self._delta_stale = False
        th, lp, dl = self.buf[self.cur]
        self._bind_stamps()
        with _rng("init"):
            self.ops.init(th, lp, dl, self.lo, self.n_local)
        with _rng("init_allgather"):
            self._allgather_state(self.buf[self.cur] + ((self.stamp[self.cur],) if self.blob_on else ()))

    def reset_weights(self):  # smc:266-270
        self.discard_select_ahead()
        self.wns.fill_(1.0 / self.N)
        self.alive.fill_(1)
        if hasattr(self.ops, "set_uniform_weights"):
            self.ops.set_uniform_weights(True)       # Wns = 1/N on every particle: indicator kernels may use the closed forms
        self.n_alive = self.N
        if self.packed:
            self.n_prev = self.N

    # ------------------------------------------------------------------ S9, S10
    def quantile_alive(self, alpha: float) -> float:
        self._stream()
        n = self.n_prev if self.packed else self.N      # packed: the alive particles are the prefix, the tail is not read
        return self.ops.quantile_alive(self.delta[:n], self.alive[:n], alpha, self.n_alive)[0]

    def extrema(self):
        self._stream()
        return self.ops.extrema(self.delta)

    def count_gt(self, thr: float) -> int:
        self._stream()
        return self.ops.count_gt(self.delta, thr)

    # ------------------------------------------------------------------ S5, S6
    def smc_reweight(self, eps_old: float, eps_new: float):
        self._need_packed("smc_reweight")
        self._stream()
        # prefix only: the dead tail has weight +0.0 and contributes exact zeros to the fixed summation trees
        n = self.n_prev
        wnorm, ess, n_alive = self.ops.smc_reweight(self.delta[:n], self.wns[:n], self.alive[:n], eps_old, eps_new)
        self.n_alive = n_alive
        return wnorm, ess, n_alive

    def get_ess(self) -> float:
        self._stream()
        return self.ops.get_ess(self.wns)

    def smc_prologue(self, alpha: float, eps_prev: float, eps_target: float, eps_k: float, ess_min: float):
        """Everything the driver does between two groups of sweeps except the resampling itself: extrema(Ds) of the
        generation that just ended (smc:364), eps = max(min(quantile(Ds[alive], alpha), eps_prev), eps_target)
        (smc:301), reweight + ESS (smc:305-311, :323) -> (eps, wnorm, ess, n_alive, (lo, hi)).  ONE library call
        with ONE host synchronisation, which also partitions the population unless ESS < ess_min announces a
        resampling."""
        self._need_packed("smc_prologue")
        self._stream()
        self._sync_delta()
        self._bind_stamps()
        cur = self.buf[self.cur]
        with _rng("prologue"):
            eps, wnorm, ess, n_alive, part, lo, hi = self.ops.smc_prologue_packed(
                cur[2], self.wns, self.alive, self.n_prev, alpha, eps_prev, eps_target, eps_k, ess_min, self.bits[self.bc],
                self.bits[1 - self.bc], self.buf[0][0], self.buf[1][0], cur[1])
        self.n_alive = n_alive
        if part:
            self.n_prev = n_alive
        return eps, wnorm, ess, n_alive, (lo, hi)

    # ------------------------------------------------------------------ S7, S8
    def smc_resample(self):
        self._need_packed("smc_resample")
        self._stream()
        with _rng("resample"):
            self.ops.wsample_stratified(self.wns, self.draw, self.inds)
            self.draw += 1
            self._bind_stamps()
            self._sync_delta()
            cur, oth = self.buf[self.cur], self.buf[1 - self.cur]
            self.ops.smc_resample_gather_packed(self.inds, self.bits[self.bc], self.bits[1 - self.bc], self.buf[0][0],
                                                self.buf[1][0], cur[1], cur[2], oth[1], oth[2], self.wns, self.alive)
        self._swap()
        self.n_alive = self.n_prev = self.N

    # ------------------------------------------------------------------ partition + S2, S3
    def alive_compact(self) -> int:
        """the reference finds its donors by scanning `alive` (smc:121,125); here the population is partitioned so that
        the alive particles are the positions [0, n_alive) (a no-op when smc_prologue has done it already)"""
        self._need_packed("alive_compact")
        self._stream()
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

    def smc_swarm(self, eps: float, gamma0: float, gsig: float, last: bool = False):
        """last: no further sweep follows in this generation (the driver's i == Kmcmc) -- lets a sharded run start
        the per-generation distance exchange early; has no effect on results"""
        self._need_packed("smc_swarm")
        self._stream()
        self._bind_stamps()
        cur = self.buf[self.cur]
        b_in, b_out = self.bits[self.bc], self.bits[1 - self.bc]
        if not self.sharded_packed:
            counts = self.ops.smc_swarm_packed(b_in, b_out, self.n_alive, 0, self.n_alive, self.buf[0][0],
                                               self.buf[1][0], cur[1], cur[2], None, eps, gamma0, gsig, self.sweep)
            self.sweep += 1
            self.bc = 1 - self.bc
            return counts
        if self._delta_work is not None:      # a caller swept again after announcing the last sweep: that exchange
            self._delta_work.wait()           # is obsolete (the stale flag stays set)
            self._delta_work = None
        self._mark("own_sweep", 0)
        self.ops.smc_swarm_packed(b_in, b_out, self.n_alive, self.r_lo, self.r_hi, self.buf[0][0], self.buf[1][0],
                                  cur[1], cur[2], self.flags, eps, gamma0, gsig, self.sweep, want_counts=False)
        self._mark("own_sweep", 1)
        self._mark("flag_allgather", 0)
        self._allgather_chunks(self.flags)               # 1 byte per position: accepted | simulated << 1
        self._mark("flag_allgather", 1)
        exchanged = bool(self._start_delta_allgather()) if last else False
        self._mark("replay", 0)
        counts = self.ops.smc_replay_packed(b_in, b_out, self.n_alive, self.r_lo, self.r_hi, self.buf[0][0],
                                            self.buf[1][0], cur[1], self.flags, gamma0, gsig, self.sweep)
        self._mark("replay", 1)
        self.sweep += 1
        self.bc = 1 - self.bc
        self._delta_stale = not exchanged       # (the library's own all-gather is already in the stream, in order)
        return counts                                    # global (nacc, nsim), counted from the flags

    def smc_sweeps(self, eps: float, gamma0: float, gsig: float, Kmcmc: int, Kmcmc_min: float, next_prologue=None):
        """The sweeps of one generation, `for i in 1:Kmcmc ... (sum(naccs) / n_alive >= Kmcmc_min) && break`
        (smc:336-353) -> (naccs per sweep, nsims per sweep, Ki).  On an unsharded HIP population the whole group is one
        library call with the test of smc:352 evaluated on the device between the sweeps (one host synchronisation);
        otherwise one call per sweep with the test here."""
        self._need_packed("smc_sweeps")
        if (not self.sharded_packed and hasattr(self.ops, "smc_sweeps_packed") and Kmcmc <= self.ops.SWEEPS_MAX
                and Kmcmc_min >= 0.0):
            self._stream()
            self._bind_stamps()
            cur = self.buf[self.cur]
            if next_prologue is not None and hasattr(self.ops, "smc_select_ahead"):
                # (alpha, eps_target) of the next smc_prologue: its select is enqueued behind these sweeps (same results;
                # the device works on it while the host reads the counters and applies its stop rules)
                self.ops.smc_select_ahead(cur[2], self.alive, next_prologue[0], next_prologue[1])
            with _rng("sweeps_group"):
                naccs, nsims, Ki = self.ops.smc_sweeps_packed(self.bits[self.bc], self.bits[1 - self.bc], self.n_alive,
                                                              self.buf[0][0], self.buf[1][0], cur[1], cur[2], eps, gamma0, gsig,
                                                              self.sweep, Kmcmc, Kmcmc_min)
            self.sweep += Ki
            if Ki & 1:
                self.bc = 1 - self.bc
            return naccs, nsims, Ki
        if self.sharded_packed and hasattr(self.ops, "smc_group_begin") and Kmcmc <= self.ops.SWEEPS_MAX and Kmcmc_min >= 0.0:
            return self._smc_sweeps_sharded_group(eps, gamma0, gsig, Kmcmc, Kmcmc_min)
        naccs, nsims = [], []
        for i in range(1, Kmcmc + 1):
            nacc, nsim = self.smc_swarm(eps, gamma0, gsig, last=(i == Kmcmc))
            naccs.append(nacc)
            nsims.append(nsim)
            if sum(naccs) / self.n_alive >= Kmcmc_min:   # smc:352
                break
        return naccs, nsims, len(naccs)

    def smc_generation(self, alpha: float, eps_prev: float, eps_target: float, eps_k: float, ess_min: float, gamma0: float, gsig: float,
                       Kmcmc: int, Kmcmc_min: float, select_ahead: bool = True):
        """The loop body of abcdesmc! between `iters += 1` and the acceptance fraction (smc:301-353): new eps, weights, evidence
        increment, ESS; the resampling when ESS < ess_min (smc:323-326); the Kmcmc sweeps with their early exit (smc:336-353).
        -> dict(eps, wnorm, ess (after a resampling: of the resampled weights, smc:325), n_alive (the count the sweeps ran on), naccs,
        nsims, Ki, range = extrema(Ds) of the generation before, resampled).
        On an unsharded HIP population ONE library call (abcdez_smc_generation_packed): the two decisions between prologue,
        resampling and sweeps are taken inside the library instead of in this interpreter, which otherwise walks from call to call
        while the device queue stands empty.  Otherwise (sharded runs, the CPU oracle) the same steps through the methods above."""
        self._need_packed("smc_generation")
        fused = (not self.sharded_packed and hasattr(self.ops, "smc_generation_packed") and Kmcmc <= self.ops.SWEEPS_MAX
                 and Kmcmc_min >= 0.0 and os.environ.get("ABZ_GENERATION_STEPWISE", "0") != "1")
        if fused:
            self._stream()
            self._sync_delta()
            self._bind_stamps()
            cur, oth = self.buf[self.cur], self.buf[1 - self.cur]
            with _rng("generation"):
                g = self.ops.smc_generation_packed(self.n_prev, self.bits[self.bc], self.bits[1 - self.bc], self.buf[0][0], self.buf[1][0],
                                                   (cur[1], cur[2]), (oth[1], oth[2]), self.wns, self.alive, self.inds, alpha, eps_prev,
                                                   eps_target, eps_k, ess_min, gamma0, gsig, self.sweep, self.draw, Kmcmc, Kmcmc_min,
                                                   select_ahead)
            self.n_alive = g["n_alive"]
            if g["partitioned"]:
                self.n_prev = g["n_alive"]
            if g["resampled"]:                     # the other (logpi, delta, stamp) arrays are the current ones now
                self.draw += 1
                self._swap()
                self.n_alive = self.n_prev = self.N
                g["ess"] = g["ess_resampled"]
            self.r_lo, self.r_hi = 0, self.n_alive
            self.sweep += g["Ki"]
            if g["Ki"] & 1:
                self.bc = 1 - self.bc
            g["n_alive"] = g["n_swept"]
            return g
        eps, wnorm, ess, n_alive, rng_prev = self.smc_prologue(alpha, eps_prev, eps_target, eps_k, ess_min)
        resampled = n_alive > 0 and ess < ess_min
        if resampled:                              # smc:323-326
            self.smc_resample()
            ess = self.get_ess()
            n_alive = self.N
        naccs, nsims, Ki = [], [], 0
        if n_alive >= 3:                           # donor draws need three alive particles (smc:119-126)
            self.alive_compact()
            naccs, nsims, Ki = self.smc_sweeps(eps, gamma0, gsig, Kmcmc, Kmcmc_min,
                                               next_prologue=(alpha, eps_target) if select_ahead and eps > eps_target else None)
        return dict(eps=eps, wnorm=wnorm, ess=ess, n_alive=n_alive, naccs=list(naccs), nsims=list(nsims), Ki=Ki, range=rng_prev,
                    resampled=resampled)

    def _smc_sweeps_sharded_group(self, eps, gamma0, gsig, Kmcmc, Kmcmc_min):
        """sharded population: all Kmcmc sweeps enqueued back to back -- own range, flag all-gather, replay + the device-side
        test of smc:352 -- and ONE read-back; every replica counts the same flags, so all ranks stop after the same sweep."""
        self._stream()
        self._bind_stamps()
        if self._delta_work is not None:
            self._delta_work.wait()
            self._delta_work = None
        cur = self.buf[self.cur]
        if self._native_comm and self._prof is None:
            # the whole group -- own chunk sweeps, flag all-gathers, replays, the device-side test of smc:352, the read-back and the
            # distance all-gather -- is ONE library call on the library's stream (abcdez_smc_sweeps_sharded)
            with _rng("sweeps_group_sharded"):
                naccs, nsims, Ki = self.ops.smc_sweeps_sharded(self.bits[self.bc], self.bits[1 - self.bc], self.n_alive, self.chunk,
                                                               self.buf[0][0], self.buf[1][0], cur[1], self._full[self.cur][1],
                                                               self.flags, eps, gamma0, gsig, self.sweep, Kmcmc, Kmcmc_min)
            self.sweep += Ki
            if Ki & 1:
                self.bc = 1 - self.bc
            self._delta_stale = False
            return naccs, nsims, Ki
        self.ops.smc_group_begin(self.n_alive, Kmcmc_min)
        try:
            bc = self.bc
            for k in range(Kmcmc):
                b_in, b_out = self.bits[bc], self.bits[1 - bc]
                self._mark("own_sweep", 0)
                with _rng("own_sweep"):
                    self.ops.smc_swarm_packed(b_in, b_out, self.n_alive, self.r_lo, self.r_hi, self.buf[0][0], self.buf[1][0],
                                              cur[1], cur[2], self.flags, eps, gamma0, gsig, self.sweep + k, want_counts=False)
                self._mark("own_sweep", 1)
                self._mark("flag_allgather", 0)
                with _rng("flag_allgather"):
                    self._allgather_chunks(self.flags)           # executed by every rank whether or not the sweep ran
                self._mark("flag_allgather", 1)
                self._mark("replay", 0)
                with _rng("replay"):
                    self.ops.smc_group_replay(b_in, b_out, self.r_lo, self.r_hi, self.buf[0][0], self.buf[1][0], cur[1], self.flags,
                                              gamma0, gsig, self.sweep + k)
                self._mark("replay", 1)
                bc = 1 - bc
            self.ops.smc_group_publish()
            self._delta_stale = True
            with _rng("delta_allgather_start"):
                self._start_delta_allgather()                    # behind the read-back: travels while the host applies its rules
            with _rng("group_readback"):
                naccs, nsims, Ki = self.ops.smc_group_end(Kmcmc)
        except BaseException:
            # a collective or a launch failed half way: close the group, or every later call of this context is refused
            if hasattr(self.ops, "smc_group_abort"):
                try:
                    self.ops.smc_group_abort()
                except Exception:
                    pass
            raise
        self.sweep += Ki
        if Ki & 1:
            self.bc = 1 - self.bc
        return naccs, nsims, Ki

    # ------------------------------------------------------------------ S4
    def _mc_arrays(self):
        """the enumeration of mc:23: order (positions -> particles), sorted_delta, and every particle's candidate count"""
        if self.packed:
            raise RuntimeError("abcdemc needs storage='classic'")
        if self.order is None:
            self.order = torch.zeros(self.N, dtype=torch.int32, device=self.device)
            self.sorted_delta = torch.zeros(self.N, dtype=torch.float64, device=self.device)
            self.rank_cnt = torch.zeros(self.N, dtype=torch.int32, device=self.device)

    def mc_rank_prepare(self, eps_pop: float = None, dmax_hint: float = None):
        """the enumeration of mc:23 for the current distances: particles with Ds <= eps_pop in index order, then the
        others by (Ds, index).  eps_pop / dmax_hint default to the values of mc:146-147 with eps_target = 0 (one
        extrema pass); the driver passes what it already knows.  No host synchronisation on the HIP path."""
        self._mc_arrays()
        self._stream()
        if eps_pop is None or dmax_hint is None:
            lo, hi = self.extrema()
            eps_pop = lo if eps_pop is None else eps_pop
            dmax_hint = hi if dmax_hint is None else dmax_hint
        self.ops.mc_rank_prepare(self.state[2], eps_pop, dmax_hint, self.order, self.sorted_delta, self.rank_cnt)

    def mc_draws_by_rejection(self, n_above: int) -> bool:
        """the rule of include/abcdez_spec.h (abz_mc_draws_by_rejection): the better particle of mc:23 is drawn by rejection,
        without a rank pass, once at least 1 / 16 of the particles lie at or below eps_target"""
        return bool(self.ops.mc_draws_by_rejection(int(n_above), self.N))

    def mc_swarm(self, eps_pop: float, eps_target: float, gamma0: float, gsig: float, reject: bool = False):
        """one sweep of abcdemc_swarm! -> (nsim, #(Ds > eps_target), min Ds, max Ds) of the generation it leaves
        (mc:149,156,146,163), global over all ranks.  reject: the better particle of mc:23 by rejection (no enumeration)"""
        self._mc_arrays()        # (a converged population never consults them)
        self._stream()
        self._bind_stamps()
        nsim, ngt, lo, hi = self.ops.mc_swarm(None if reject else self.order, None if reject else self.rank_cnt, self.state,
                                              self.other, eps_pop, eps_target, gamma0, gsig, self.lo, self.n_local, self.sweep)
        self.sweep += 1
        with _rng("mc_rows_allgather"):
            self._allgather_state(self.other + ((self.stamp[1 - self.cur],) if self.blob_on else ()))
        self._swap()
        nsim, ngt = self._allreduce_counts(nsim, ngt)
        if self._collectives:
            import torch.distributed as dist

            t = torch.tensor([lo, -hi], dtype=torch.float64, device=self.device if self._native_comm else "cpu")
            if self._native_comm:
                self.ops.comm_allreduce(t, "min")
            else:
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.pg)
            lo, hi = float(t[0]), -float(t[1])
        return nsim, ngt, lo, hi

    def mc_generation(self, eps_pop: float, eps_target: float, dmax: float, gamma0: float, gsig: float, n_above: int = None):
        """the body of abcdemc!'s loop (mc:140-156): rank pass while some Ds > eps_target and the better particle is drawn by
        rank, one sweep -> (nsim, #(Ds > eps_target), min Ds, max Ds) of the new generation.  One library call on a single
        GPU.  n_above = #(Ds > eps_target) of the current distances when the caller has it (mc:133 / mc:156 of the generation
        before); counted here otherwise."""
        if not self._collectives and hasattr(self.ops, "mc_generation"):
            self._mc_arrays()
            self._stream()
            self._bind_stamps()
            out = self.ops.mc_generation(self.state, self.other, self.order, self.sorted_delta, self.rank_cnt, eps_pop,
                                         eps_target, dmax, -1 if n_above is None else int(n_above), gamma0, gsig, self.sweep)
            self.sweep += 1
            self._swap()
            return out
        if n_above is None:
            n_above = self.count_gt(eps_target)
        reject = self.mc_draws_by_rejection(n_above)
        if dmax > eps_target and not reject:
            self.mc_rank_prepare(eps_pop, dmax)
        return self.mc_swarm(eps_pop, eps_target, gamma0, gsig, reject=reject)

    # abcdemc!'s loop has no data-dependent exit (mc:134): generations can be ENQUEUED ahead of their results
    def mc_generation_issue(self, alpha: float, eps_target: float, gamma0: float, gsig: float, lo_hi=None, do_rank: bool = True):
        """Enqueue one generation (mc:146-149): eps_pop = max(eps_target, lo + alpha (hi - lo)) from the extrema of the
        generation before -- on an unsharded HIP population they never leave the device and this call does not wait;
        `lo_hi` = the extrema when the host has them (first generation).  Results: :meth:`mc_generation_collect`, in
        issue order.  do_rank=False skips the rank pass (only once a collected generation reported max Ds <= eps_target)."""
        if not hasattr(self, "_mc_pending"):
            self._mc_pending, self._mc_last = [], None
        if not self._collectives and hasattr(self.ops, "mc_generation_async"):
            self._mc_arrays()
            self._stream()
            self._bind_stamps()
            with _rng("mc_generation"):
                t = self.ops.mc_generation_async(self.state, self.other, self.order, self.sorted_delta, self.rank_cnt, alpha,
                                                 eps_target, lo_hi, do_rank, gamma0, gsig, self.sweep)
            self.sweep += 1
            self._swap()
            self._mc_pending.append(("ticket", t))
            return
        if self._native_comm and hasattr(self.ops, "mc_generation_sharded_async"):
            # sharded over the library's communicator: nothing comes back to the host between the sweep, the exchange of the new
            # rows and the reductions -- generations are issued ahead exactly as on one GPU
            self._mc_arrays()
            self._stream()
            self._bind_stamps()
            if lo_hi is None and not getattr(self, "_mc_sharded_chain", False):
                lo_hi = self._mc_last if self._mc_last is not None else self.extrema()
            with _rng("mc_generation_sharded"):
                t = self.ops.mc_generation_sharded_async(self.state, self.other, self.order, self.sorted_delta, self.rank_cnt, alpha,
                                                         eps_target, lo_hi, do_rank, gamma0, gsig, self.sweep)
            self._mc_sharded_chain = True
            self.sweep += 1
            self._swap()
            self._mc_pending.append(("ticket", t))
            return
        lo, hi = lo_hi if lo_hi is not None else self._mc_last
        v = lo + alpha * (hi - lo)
        eps_pop = max(eps_target, v)                                                       # mc:147
        # #(Ds > eps_target) of the distances this generation reads: mc:156 of the generation before (for the same eps_target)
        known = getattr(self, "_mc_above", None)
        n_above = known[1] if lo_hi is None and known is not None and known[0] == eps_target else None
        nsim, ngt, nlo, nhi = self.mc_generation(eps_pop, eps_target, hi if do_rank else -math.inf, gamma0, gsig, n_above=n_above)
        self._mc_last = (nlo, nhi)
        self._mc_above = (eps_target, ngt)
        self._mc_pending.append(("done", (nsim, ngt, nlo, nhi, eps_pop)))

    def mc_generations_in_flight(self) -> int:
        return len(getattr(self, "_mc_pending", []))

    def mc_generation_collect(self):
        """-> (nsim, #(Ds > eps_target), min Ds, max Ds, eps_pop) of the oldest generation issued and not yet collected"""
        kind, v = self._mc_pending.pop(0)
        if kind == "done":
            return v
        out = self.ops.mc_generation_wait(v)
        self._mc_last = (out[2], out[3])
        return out

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
            # every random number after the resume depends on the library's stream: (ABI version, Philox rounds)
            "stream_version": list(self.ops.stream_version()) if hasattr(self.ops, "stream_version") else None,
            # which path the next reweight of an indicator kernel takes (closed forms on uniform weights, or the general sums)
            "w_uniform": bool(self.ops.get_uniform_weights()) if hasattr(self.ops, "get_uniform_weights") else False,
        }

    def upload_state(self, st: dict):
        """Inverse of :meth:`download_state` (every rank uploads the full arrays)."""
        self.discard_select_ahead()
        if self._delta_work is not None:
            self._delta_work.wait()
            self._delta_work = None
        th = torch.as_tensor(np.ascontiguousarray(st["theta"], dtype=np.float64))
        if tuple(th.shape) != tuple(self.buf[0][0].shape):
            raise ValueError(f"checkpoint holds a population of shape {tuple(th.shape)}, "
                             f"the engine expects {tuple(self.buf[0][0].shape)}")
        if int(st.get("seed", self.spec.seed)) != int(self.spec.seed):
            raise ValueError("checkpoint was written with a different seed: the run would not continue its own stream")
        mine = list(self.ops.stream_version()) if hasattr(self.ops, "stream_version") else None
        theirs = st.get("stream_version")
        if mine is not None and theirs is not None and list(theirs)[1:] != mine[1:]:
            raise ValueError(f"checkpoint was written by a build with Philox4x32-{list(theirs)[1]}, this library runs "
                             f"Philox4x32-{mine[1]}: the run would not continue its own stream")
        if mine is not None and theirs is not None and mine[0] != 0 and list(theirs)[0] not in (0, mine[0]):
            import warnings

            warnings.warn(f"checkpoint was written by library version {list(theirs)[0]}, this is {mine[0]}: a version step may change "
                          "which random numbers a generation consumes (401: abcdemc's better particle by rejection) -- the run "
                          "continues, but not necessarily bit for bit", stacklevel=2)
        if mine is not None and theirs is None:
            import warnings
            warnings.warn("checkpoint carries no stream_version (written before round 4): it resumes bit for bit only if it "
                          f"was written by a Philox4x32-{mine[1]} build", stacklevel=2)
        self.cur = 0
        cur = self.buf[0]
        cur[0].copy_(th)
        cur[1].copy_(torch.as_tensor(np.ascontiguousarray(st["logpi"], dtype=np.float64)))
        cur[2].copy_(torch.as_tensor(np.ascontiguousarray(st["delta"], dtype=np.float64)))
        self.wns.copy_(torch.as_tensor(np.ascontiguousarray(st["wns"], dtype=np.float64)))
        self.alive.copy_(torch.as_tensor(np.ascontiguousarray(st["alive"], dtype=np.uint8)))
        self.n_alive = int(self.alive.sum().item())
        if hasattr(self.ops, "set_uniform_weights"):
            self.ops.set_uniform_weights(bool(st.get("w_uniform", False)))
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
        self._delta_stale = False

    # ------------------------------------------------------------------ results (smc:382-393, mc:166-171)
    def posterior_mean(self) -> np.ndarray:
        """Weighted mean of the population, sum(Wns theta) / sum(Wns) over the d parameters -- for the indicator kernels the
        mean of the alive particles, what the reference's tests extract as mean(r.P[r.Wns .> 0.0]) (test/runtests.jl:159-162);
        abcdemc (no weights): the plain mean.  A result / measurement helper (one torch reduction), not part of the loop."""
        self._stream()
        th = self.state[0][:, :self.spec.d]
        if not self.packed:
            return th.mean(dim=0).cpu().numpy()
        w = self.wns
        return ((w[:, None] * th).sum(dim=0) / w.sum()).cpu().numpy()

    def _to_host(self, tensors):
        """device tensors -> numpy arrays.  On the GPU: contiguous copies into PINNED host memory, all enqueued back to back on
        the engine's stream, one wait -- the copy engine runs at the link's rate (a pageable destination goes through the
        runtime's bounce buffers at a quarter of it).  The arrays returned view the pinned blocks."""
        if self.device.type != "cuda":
            return [t.cpu().numpy() for t in tensors]
        host = []
        for t in tensors:
            t = t if t.is_contiguous() else t.contiguous()         # (a column range of the rows: compacted on the device first)
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            host.append(h)
        torch.cuda.current_stream(self.device).synchronize()
        return [h.numpy() for h in host]

    def result(self):
        """the population as the drivers return it (smc:382-393, mc:166-171): P (push_p-cast), the internal rows, logpi, C, Wns,
        alive and -- with blobs on -- the simulated data behind every distance.  Only the d real columns of the rows travel, and
        when no dimension is discrete push_p is the identity (types.jl:20-23): P and theta are then ONE array, moved once."""
        self._stream()
        th, lp, dl = self.state
        d = self.spec.d
        discrete = any(getattr(self.spec, "discrete", (True,)))       # ModelSpec.discrete: push_p rule per dimension
        if discrete:
            pushed = torch.empty_like(th)
            self.ops.push_p(th, pushed)
        blobs = None
        blob_dev = None
        if self.blob_on:
            # rebuild the simulated data behind every particle's distance from its stamp; the re-run's distance
            # must be the stored one, bit for bit
            nb = self.spec.n_blob
            out = torch.zeros((self.N, self.ops.blob_width()), dtype=torch.float64, device=self.device)
            redo = torch.empty_like(dl)
            self.ops.blob_eval(th, self.stamp[self.cur], out, redo)
            if not torch.equal(redo.view(torch.int64), dl.view(torch.int64)):
                raise RuntimeError("blobs: a re-run simulation does not reproduce the stored distance")
            blob_dev = out[:, :nb]
        send = [th[:, :d], lp, dl, self.wns, self.alive] + ([pushed[:, :d]] if discrete else []) + ([blob_dev] if blob_dev is not None else [])
        got = self._to_host(send)
        theta, logpi, Cc, Wns, alive = got[:5]
        P = got[5] if discrete else theta
        if blob_dev is not None:
            blobs = got[-1]
            if blobs.shape[1] == 1:
                blobs = blobs[:, 0]
        if d == 1 and not hasattr(self.spec.prior, "p"):
            P = P[:, 0]                      # univariate prior -> vector of scalars
        return {
            "blobs": blobs,
            "P": P,
            "theta": theta,                  # internal (unrounded) state, mc:216-220
            "logpi": logpi,
            "C": Cc,
            "Wns": Wns,
            "alive": alive.astype(bool),
        }


def HipEngine(spec: ModelSpec, nparticles: int, process_group=None, lanes: int = 0,
              storage: str = "packed", force_collectives: bool = False) -> PopulationEngine:
    """The product engine: HIP kernels on the current CUDA(HIP) device.  storage="packed" for abcdesmc (only accepted
    proposals are written; sharded runs keep one replica per GPU and exchange accept flags), "classic" for abcdemc."""
    return PopulationEngine(spec, nparticles, process_group, ops=None, lanes=lanes, storage=storage,
                            force_collectives=force_collectives)
