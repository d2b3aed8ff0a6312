"""ctypes binding of the CPU oracle (oracle/abcdez_oracle*.c) + an ops backend for
``PopulationEngine``.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

_vp, _i64, _u32, _i32, _f64, _u64 = C.c_void_p, C.c_int64, C.c_uint32, C.c_int32, C.c_double, C.c_uint64
_pi64, _pf64 = C.POINTER(C.c_int64), C.POINTER(C.c_double)


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = line.split()
                    return "fma" in fl and "avx2" in fl
    except OSError:
        pass
    return False


def build(force: bool = False) -> None:
    if force or not os.path.exists(os.path.join(HERE, "liboracle.so")):
        subprocess.check_call(["make", "-C", HERE, "-s"])


class SmcRun(C.Structure):
    _fields_ = [
        ("nparticles", _i64), ("eps_target", _f64), ("alpha", _f64), ("delta_ess", _f64), ("nsims_max", _i64),
        ("Kmcmc", _i32), ("max_iters", _i32), ("Kmcmc_min", _f64), ("facc_stop", _f64), ("facc_min", _f64),
        ("facc_tune", _f64), ("eps", _f64), ("logZ", _f64), ("iters", _i64), ("nsims_total", _i64),
        ("updates_total", _i64), ("n_hist", _i32), ("no_alive", _i32), ("packed", _i32), ("reserved", _i32),
    ]


class McRun(C.Structure):
    _fields_ = [
        ("nparticles", _i64), ("generations", _i32), ("reserved", _i32), ("eps_target", _f64),
        ("nsims_total", _i64), ("reached_eps", _i32), ("reserved2", _i32), ("complete", _f64),
    ]


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    build()
    name = "liboracle_fma.so" if _cpu_has_fma() and os.path.exists(os.path.join(HERE, "liboracle_fma.so")) \
        else "liboracle.so"
    L = C.CDLL(os.path.join(HERE, name))
    L.orc_kernel_pdf.restype = _f64
    L.orc_kernel_pdf.argtypes = [C.c_int, _f64, _f64]
    L.orc_kernel_logpdf.restype = _f64
    L.orc_kernel_logpdf.argtypes = [C.c_int, _f64, _f64]
    L.orc_philox.argtypes = [_vp, _vp, _vp]
    L.orc_philox_r.argtypes = [C.c_int, _vp, _vp, _vp]
    L.orc_philox_rounds.restype = C.c_int
    L.orc_donor_ranks.argtypes = [_u64, _u64, _u32, _u32, _vp, _vp]
    L.orc_weight_fix.restype = _u64
    L.orc_weight_fix.argtypes = [_f64, _u32]
    L.orc_u01.restype = _f64
    L.orc_u01.argtypes = [_u64, C.c_int]
    L.orc_randint.restype = _u32
    L.orc_randint.argtypes = [_u64, _u32]
    L.orc_math_eval.argtypes = [C.c_int, _vp, _vp, _vp, _i64]
    L.orc_particle_draws.argtypes = [_u64, _i64, _i64, _i64, _u32, _f64, _f64, _vp, _vp, _vp, _vp]
    L.orc_mc_better_by_rejection.argtypes = [_u64, _vp, _i64, _u32, _u32, _i64, _vp, _vp]
    L.orc_mc_better_by_rejection.restype = None
    L.orc_mc_draws_by_rejection.argtypes = [_i64, _i64]
    L.orc_mc_draws_by_rejection.restype = C.c_int
    L.orc_rng_words.argtypes = [_u64, _u32, _u32, _u32, _u32, _vp]
    L.orc_normal_pairs.argtypes = [_u64, _u32, _i64, _vp]
    L.orc_push_p.argtypes = [_vp, _vp, _i64, _vp]
    L.orc_logprior.argtypes = [_vp, _vp, _i64, C.c_int, _vp]
    L.orc_sim_dist.restype = _f64
    L.orc_sim_dist.argtypes = [_vp, _vp, _u32, _u32, _u32]
    L.orc_init.argtypes = [_vp, _vp, _vp, _vp, _i64, _i64]
    L.orc_alive_compact.restype = _i64
    L.orc_alive_compact.argtypes = [_vp, _i64, _vp, _vp]
    L.orc_smc_swarm.argtypes = [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _i64, _i64, _u32,
                                _pi64, _pi64]
    L.ref_smc_swarm.argtypes = [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _u32, _pi64, _pi64]
    L.orc_tree_sum.restype = _f64
    L.orc_tree_sum.argtypes = [_vp, _i64]
    L.orc_smc_reweight.argtypes = [C.c_int, _vp, _vp, _vp, _i64, _f64, _f64, _pf64, _pf64, _pi64]
    L.orc_smc_reweight_uniform.argtypes = [C.c_int, _vp, _vp, _vp, _i64, _f64, _pf64, _pf64, _pi64]
    L.orc_get_ess.restype = _f64
    L.orc_get_ess.argtypes = [_vp, _i64]
    L.ref_get_ess.restype = _f64
    L.ref_get_ess.argtypes = [_vp, _i64]
    L.ref_smc_reweight.argtypes = [C.c_int, _vp, _vp, _vp, _vp, _i64, _f64, _f64, _pf64]
    L.ref_wsample_stratified.argtypes = [_vp, _i64, _vp, _vp]
    L.orc_wsample_stratified.argtypes = [_u64, _vp, _i64, _u32, _vp]
    L.orc_stratum_uniforms.argtypes = [_u64, _i64, _u32, _vp]
    L.orc_smc_resample_gather.argtypes = [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.orc_smc_partition.restype = _i64
    L.orc_smc_partition.argtypes = [_vp, _i64, _vp, _vp, _vp, _vp, _vp]
    L.orc_packed_partition.restype = _i64
    L.orc_packed_partition.argtypes = [_vp, _i64, _i64] + [_vp] * 8
    L.orc_packed_gather.argtypes = [_vp, _i64, C.c_int, _vp, _vp, _vp]
    L.orc_smc_swarm_packed.argtypes = [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _u32,
                                       _pi64, _pi64]
    L.orc_smc_replay_packed.argtypes = [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _f64, _f64, _u32, _pi64, _pi64]
    L.orc_smc_resample_gather_packed.argtypes = [_vp, _vp, _i64] + [_vp] * 10
    L.orc_set_stamps.argtypes = [_vp, _vp]
    L.orc_blob_eval.argtypes = [_vp, _vp, _vp, _i64, _vp, C.c_int, _vp]
    L.orc_quantile_alive.restype = _f64
    L.orc_quantile_alive.argtypes = [_vp, _vp, _i64, _f64, _pf64, _pf64]
    L.orc_extrema.argtypes = [_vp, _i64, _pf64, _pf64]
    L.orc_count_gt.restype = _i64
    L.orc_count_gt.argtypes = [_vp, _i64, _f64]
    L.orc_mc_rank_prepare.argtypes = [_vp, _i64, _f64, _vp, _vp, _vp]
    L.orc_mc_swarm.argtypes = [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _f64, _i64, _i64,
                               _u32, _pi64]
    L.orc_mc_draws.argtypes = [_vp, _vp, _vp, _i64, _vp, _f64, _f64, _i64, _u32, _vp, _vp, _vp, _vp]
    L.ref_mc_draws.argtypes = [_vp, _i64, _vp, _f64, _f64, _i64, _u32, _vp, _vp, _vp, _vp]
    L.ref_mc_swarm.argtypes = [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _f64, _u32, _pi64]
    L.orc_abcdesmc.argtypes = [_vp, C.POINTER(SmcRun)] + [_vp] * 13
    L.orc_abcdemc.argtypes = [_vp, C.POINTER(McRun), _vp, _vp, _vp]
    _LIB = L
    return L


def _p(a):
    """pointer of a numpy array or CPU torch tensor"""
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        assert a.device.type == "cpu" and a.is_contiguous()
        return a.data_ptr()
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


class OracleModel:
    """keeps the C model struct (and the host data it points to) alive"""

    def __init__(self, spec):
        self.spec = spec
        self._data = np.ascontiguousarray(spec.data, dtype=np.float64)
        self.c = spec.cstruct(self._data.ctypes.data if self._data.size else None)
        self.ref = C.byref(self.c)
        self.ptr = C.addressof(self.c)


class OracleOps:
    """Same interface as abcdez_amd.engine.HipOps, on CPU tensors, through the oracle."""

    name = "oracle"

    def __init__(self, spec, device_index=None, lanes: int = 0):
        self.L = lib()
        self.spec = spec
        self.m = OracleModel(spec)
        self.device = torch.device("cpu")

    def init(self, theta, logpi, delta, i0, n):
        rc = self.L.orc_init(self.m.ptr, _p(theta), _p(logpi), _p(delta), i0, n)
        if rc:
            raise RuntimeError("oracle init: no finite (log-prior, distance) within the retry limit")

    # ---- packed store (checker for the abcdez_*_packed entry points) ----
    supports_packed = True

    def smc_partition(self, n_prev, n_new, alive, bits, bits_other, slot0, slot1, logpi, delta, wns):
        got = self.L.orc_packed_partition(self.m.ptr, alive.numel(), n_prev, _p(alive), _p(bits), _p(bits_other), _p(slot0),
                                          _p(slot1), _p(logpi), _p(delta), _p(wns))
        assert got == n_new, (got, n_new)

    def smc_prologue_packed(self, delta, wns, alive, n_prev, alpha, eps_prev, eps_target, eps_k, ess_min, bits, bits_other,
                            slot0, slot1, logpi):
        """the checker of abcdez_smc_prologue_packed: the same statements, one oracle call each"""
        lo, hi = self.extrema(delta)
        q = self.quantile_alive(delta[:n_prev], alive[:n_prev], alpha)[0]
        eps = max(min(q, eps_prev), eps_target)
        if getattr(self, "w_uniform", False) and self.spec.abck in (0, 1):
            # indicator kernel + uniform weights: the closed forms (abcdez_oracle.c, orc_smc_reweight_uniform) -- the library
            # takes this path under the same two conditions (abcdez_ctx_set_uniform_weights); the weights stay uniform
            wnorm, ess, na = C.c_double(), C.c_double(), _i64()
            self.L.orc_smc_reweight_uniform(self.spec.abck, _p(delta[:n_prev]), _p(wns[:n_prev]), _p(alive[:n_prev]), n_prev, eps,
                                            C.byref(wnorm), C.byref(ess), C.byref(na))
            wnorm, ess, n_alive = wnorm.value, ess.value, na.value
        else:
            wnorm, ess, n_alive = self.smc_reweight(delta[:n_prev], wns[:n_prev], alive[:n_prev], eps_k, eps)
        part = not (n_alive > 0 and ess < ess_min)
        if part:
            self.smc_partition(n_prev, n_alive, alive, bits, bits_other, slot0, slot1, logpi, delta, wns)
        return eps, wnorm, ess, n_alive, part, lo, hi

    def smc_swarm_packed(self, bits, bits_out, n_alive, r_lo, r_hi, slot0, slot1, logpi, delta, flags, eps, gamma0, gsig,
                         sweep, want_counts=True):
        if getattr(self, "_grp", None) and self._grp["stop"]:
            return None                               # inside a group whose early-exit test has held: this sweep does not run
        nacc, nsim = _i64(), _i64()
        self.L.orc_smc_swarm_packed(self.m.ptr, _p(bits), _p(bits_out), n_alive, r_lo, r_hi, _p(slot0), _p(slot1),
                                    _p(logpi), _p(delta), _p(flags), eps, gamma0, gsig, sweep, C.byref(nacc), C.byref(nsim))
        return nacc.value, nsim.value

    def smc_replay_packed(self, bits, bits_out, n_alive, skip_lo, skip_hi, slot0, slot1, logpi, flags, gamma0, gsig, sweep):
        nacc, nsim = _i64(), _i64()
        self.L.orc_smc_replay_packed(self.m.ptr, _p(bits), _p(bits_out), n_alive, skip_lo, skip_hi, _p(slot0), _p(slot1),
                                     _p(logpi), _p(flags), gamma0, gsig, sweep, C.byref(nacc), C.byref(nsim))
        return nacc.value, nsim.value

    # the grouped sweeps of a sharded population (include/abcdez_hip.h: abcdez_smc_group_*), restated synchronously: a sweep /
    # replay behind a held test of smc:352 does nothing, exactly like the launches that return at once on the device
    SWEEPS_MAX = 16

    def smc_group_begin(self, n_alive, kmcmc_min):
        self._grp = dict(n_alive=n_alive, kmin=kmcmc_min, nacc=[], nsim=[], stop=False)

    def smc_group_replay(self, bits, bits_out, skip_lo, skip_hi, slot0, slot1, logpi, flags, gamma0, gsig, sweep):
        g = self._grp
        if g["stop"]:
            return
        nacc, nsim = self.smc_replay_packed(bits, bits_out, g["n_alive"], skip_lo, skip_hi, slot0, slot1, logpi, flags, gamma0,
                                            gsig, sweep)
        g["nacc"].append(nacc)
        g["nsim"].append(nsim)
        g["stop"] = sum(g["nacc"]) / g["n_alive"] >= g["kmin"]          # smc:352

    def smc_group_publish(self):
        pass

    def smc_group_abort(self):
        self._grp = None

    def stream_version(self):
        """(0 = the oracle, Philox rounds of the shared spec header)"""
        return 0, int(self.L.orc_philox_rounds())

    def smc_group_end(self, k_max):
        g, self._grp = self._grp, None
        return g["nacc"], g["nsim"], len(g["nacc"])

    # uniform-weights state of the population (include/abcdez_hip.h, abcdez_ctx_set_uniform_weights): the same transitions as
    # the library's -- set by the host after it wrote 1/N, kept by the indicator fast path, set by a resampling, cleared by a
    # general reweight
    w_uniform = False

    def set_uniform_weights(self, on: bool):
        self.w_uniform = bool(on)

    def get_uniform_weights(self) -> bool:
        return bool(self.w_uniform)

    def smc_resample_gather_packed(self, inds, bits, bits_other, slot0, slot1, logpi, delta, nlogpi, ndelta, wns, alive):
        self.w_uniform = True                          # Wns .= 1/N (smc:102)
        self.L.orc_smc_resample_gather_packed(self.m.ptr, _p(inds), inds.numel(), _p(bits), _p(bits_other), _p(slot0),
                                              _p(slot1), _p(logpi), _p(delta), _p(nlogpi), _p(ndelta), _p(wns), _p(alive))

    def packed_gather(self, bits, slot0, slot1, out):
        self.L.orc_packed_gather(_p(bits), out.shape[0], self.spec.ld, _p(slot0), _p(slot1), _p(out))

    def smc_reweight(self, delta, wns, alive, eps_old, eps_new):
        self.w_uniform = False                         # the general path: weights as the floating sums leave them
        wnorm, ess, na = _f64(), _f64(), _i64()
        self.L.orc_smc_reweight(self.spec.abck, _p(delta), _p(wns), _p(alive), delta.numel(), eps_old, eps_new,
                                C.byref(wnorm), C.byref(ess), C.byref(na))
        return wnorm.value, ess.value, na.value

    def get_ess(self, wns) -> float:
        return self.L.orc_get_ess(_p(wns), wns.numel())

    def tree_sum(self, x) -> float:
        return self.L.orc_tree_sum(_p(x), x.numel())

    def wsample_stratified(self, wns, draw, inds):
        self.L.orc_wsample_stratified(self.spec.seed, _p(wns), wns.numel(), draw, _p(inds))

    def quantile_alive(self, delta, alive, p, n_alive=-1):
        a, b = _f64(), _f64()
        q = self.L.orc_quantile_alive(_p(delta), _p(alive), delta.numel(), p, C.byref(a), C.byref(b))
        return q, a.value, b.value

    def extrema(self, delta):
        lo, hi = _f64(), _f64()
        self.L.orc_extrema(_p(delta), delta.numel(), C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def count_gt(self, delta, thr) -> int:
        return self.L.orc_count_gt(_p(delta), delta.numel(), thr)

    def mc_draws_by_rejection(self, n_above: int, N: int) -> bool:
        return bool(self.L.orc_mc_draws_by_rejection(n_above, N))

    def mc_rank_prepare(self, delta, eps_pop, dmax_hint, order, sorted_delta, cnt):
        self.L.orc_mc_rank_prepare(_p(delta), delta.numel(), eps_pop, _p(order), _p(sorted_delta), _p(cnt))

    def mc_swarm(self, order, cnt, cur, nxt, eps_pop, eps_target, gamma0, gsig, i0, n_local, sweep):
        nsim = _i64()
        self.L.orc_mc_swarm(self.m.ptr, _p(order), _p(cnt), cur[1].numel(), _p(cur[0]), _p(cur[1]),
                            _p(cur[2]), _p(nxt[0]), _p(nxt[1]), _p(nxt[2]), eps_pop, eps_target, gamma0, gsig, i0,
                            n_local, sweep, C.byref(nsim))
        nd = nxt[2][i0:i0 + n_local]            # the driver reductions of the new generation, over this call's particles
        if n_local == 0:
            return nsim.value, 0, float("inf"), float("-inf")
        return nsim.value, int((nd > eps_target).sum()), float(nd.min()), float(nd.max())

    def push_p(self, theta, out):
        self.L.orc_push_p(self.m.ptr, _p(theta), theta.shape[0], _p(out))

    # ---- blobs ----
    def set_stamps(self, cur, nxt):
        self.L.orc_set_stamps(_p(cur), _p(nxt))

    def blob_width(self) -> int:
        return self.spec.ld if self.spec.sim.sim_id == 1 else self.spec.n_blob      # ABZ_SIM_MVN: laid out like a row

    def blob_eval(self, theta, stamp, blob, delta_out):
        self.L.orc_blob_eval(self.m.ptr, _p(theta), _p(stamp), theta.shape[0], _p(blob), blob.shape[1], _p(delta_out))

    def math_eval(self, fn, x, y, y2=None):
        self.L.orc_math_eval(fn, _p(x), _p(y), _p(y2), x.numel())

    def draws_eval(self, lanes, i0, n_pool, sweep, gamma0, gsig, ra, rb, g, log_u):
        self.L.orc_particle_draws(self.spec.seed, i0, ra.numel(), n_pool, sweep, gamma0, gsig, _p(ra), _p(rb), _p(g),
                                  _p(log_u))


def oracle_engine(spec, nparticles, process_group=None, storage="packed"):
    """PopulationEngine (the product's host logic) driven by the oracle instead of the GPU."""
    import abcdez_amd.engine as E

    return E.PopulationEngine(spec, nparticles, process_group, ops=OracleOps(spec), storage=storage)


def run_abcdesmc(spec, nparticles, eps_target, alpha=0.95, delta_ess=0.5, nsims_max=10 ** 7, Kmcmc=3, Kmcmc_min=1.0,
                 facc_stop=0.0, facc_min=0.0, facc_tune=0.975, max_iters=100000, packed=True):
    """The C restatement of the whole driver (oracle/abcdez_oracle_driver.c) on dense arrays.  packed (default, the
    spec): the population is partitioned after every reweight so that the alive particles form a prefix
    (orc_smc_partition); packed=False: particles keep their index for life as in the reference -- the same algorithm
    under another labelling of the particles, used by the tests that show the two agree in law."""
    L = lib()
    L.orc_set_stamps(None, None)      # the C drivers carry no blobs: unbind stamp arrays an earlier engine left behind
    m = OracleModel(spec)
    N, ld = nparticles, spec.ld
    R = SmcRun(nparticles=N, eps_target=eps_target, alpha=alpha, delta_ess=delta_ess, nsims_max=nsims_max,
               Kmcmc=Kmcmc, max_iters=max_iters, Kmcmc_min=Kmcmc_min, facc_stop=facc_stop, facc_min=facc_min,
               facc_tune=facc_tune, packed=int(packed))
    theta = np.zeros((N, ld)); logpi = np.zeros(N); delta = np.zeros(N); wns = np.zeros(N)
    alive = np.zeros(N, dtype=np.uint8)
    H = max_iters + 2
    h = [np.zeros(H) for _ in range(7)]
    hK = np.zeros(H, dtype=np.int32)
    rc = L.orc_abcdesmc(m.ptr, C.byref(R), _p(theta), _p(logpi), _p(delta), _p(wns), _p(alive),
                        *[_p(a) for a in h], _p(hK))
    if rc:
        raise RuntimeError("oracle abcdesmc failed")
    n = R.n_hist
    return dict(theta=theta[:, :spec.d], logpi=logpi, C=delta, Wns=wns, alive=alive.astype(bool), eps=R.eps,
                logZ=R.logZ, iters=R.iters, nsims=R.nsims_total, updates=R.updates_total, no_alive=bool(R.no_alive),
                eps_hist=h[0][:n], lo_hist=h[1][:n], hi_hist=h[2][:n], logZ_hist=h[3][:n], ess_hist=h[4][:n],
                facc_hist=h[5][:n], gamma0_hist=h[6][:n], K_hist=hK[:n])


def run_abcdemc(spec, nparticles, eps_target, generations):
    L = lib()
    L.orc_set_stamps(None, None)      # as in run_abcdesmc
    m = OracleModel(spec)
    N, ld = nparticles, spec.ld
    R = McRun(nparticles=N, generations=generations, eps_target=eps_target)
    theta = np.zeros((N, ld)); logpi = np.zeros(N); delta = np.zeros(N)
    rc = L.orc_abcdemc(m.ptr, C.byref(R), _p(theta), _p(logpi), _p(delta))
    if rc:
        raise RuntimeError("oracle abcdemc failed")
    return dict(theta=theta[:, :spec.d], logpi=logpi, C=delta, reached_eps=bool(R.reached_eps), nsims=R.nsims_total,
                complete=R.complete)
