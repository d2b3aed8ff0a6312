"""The call sequence of julia/ABCdeZHIP.jl, transliterated to raw ctypes: NO torch tensors, no engine.py -- device
memory from abcdez_dev_alloc, transfers with abcdez_memcpy_*, one C call per reference function in the order the
shim issues them (packed population: abcdez_smc_prologue_packed / _swarm_packed / _resample_gather_packed /
abcdez_packed_gather; abcdemc on the double buffer: abcdez_mc_rank_prepare / abcdez_mc_swarm; blobs; a user-supplied simulator).  Julia is not available to run the
shim itself; this pins what it relies on -- argument order and types of every entry point it binds, library-side
defaults, the need to initialise Wns / alive -- against the oracle's complete drivers, bit for bit."""
import ctypes as C
import math

import numpy as np
import pytest

import abcdez_amd as A
from abcdez_amd import _lib

pytestmark = pytest.mark.gpu


class ShimEngine:
    """mutable struct Engine of julia/ABCdeZHIP.jl (1-based Julia indices become 0-based here)"""

    def __init__(self, prior, sim, ABCk, seed, N):
        self.lib = _lib.load()
        self.spec = A.ModelSpec(prior, sim, ABCk, seed=seed)
        lay = (C.c_int32 * 32)()                                    # check_abi()
        assert self.lib.abcdez_abi_layout(lay, 32) == 23
        self._data = np.ascontiguousarray(self.spec.data, dtype=np.float64)
        m = self.spec.cstruct(self._data.ctypes.data if self._data.size else None)
        ctx = C.c_void_p()
        src = getattr(sim, "source", None)
        if src is not None:
            self.ck(self.lib.abcdez_ctx_create_user(C.byref(m), src.encode(), 0, C.byref(ctx)))
        else:
            self.ck(self.lib.abcdez_ctx_create(C.byref(m), 0, C.byref(ctx)))
        self.ck(self.lib.abcdez_ctx_reserve(ctx, N))
        self.ctx, self.N, self.ld, self.d, self.nb = ctx, N, self.spec.ld, self.spec.d, self.spec.n_blob
        al = self.devalloc
        nw = (N + 31) // 32
        self.slot = [al(8 * N * self.ld) for _ in range(2)]
        self.logpi = [al(8 * N) for _ in range(2)]
        self.delta = [al(8 * N) for _ in range(2)]
        self.bits = [al(4 * nw) for _ in range(2)]
        self.stamp = [al(8 * N) for _ in range(2)] if self.nb > 0 else []
        self.wns, self.alive, self.inds = al(8 * N), al(N), al(4 * N)
        self.order, self.sorted, self.cnt = al(4 * N), al(8 * N), al(4 * N)
        self.cur, self.bc, self.sweep, self.draw, self.n_alive, self.n_prev = 0, 0, 0, 0, N, N
        z = np.zeros(nw, dtype=np.uint32)
        for b in self.bits:
            self.ck(self.lib.abcdez_memcpy_h2d(self.ctx, b, z.ctypes.data, 4 * nw))

    def ck(self, rc):
        assert rc == 0, self.lib.abcdez_last_error()

    def devalloc(self, nbytes):
        p = C.c_void_p()
        self.ck(self.lib.abcdez_dev_alloc(nbytes, C.byref(p)))
        return p

    @property
    def other(self):
        return 1 - self.cur

    def bind_stamps(self):
        if self.nb > 0:
            self.ck(self.lib.abcdez_ctx_set_stamps(self.ctx, self.stamp[self.cur], self.stamp[self.other]))

    def init(self):
        self.bind_stamps()
        self.ck(self.lib.abcdez_init(self.ctx, self.slot[0], self.logpi[self.cur], self.delta[self.cur], 0, self.N))

    def reset_weights(self):
        w = np.full(self.N, 1.0 / self.N)
        a = np.ones(self.N, dtype=np.uint8)
        self.ck(self.lib.abcdez_memcpy_h2d(self.ctx, self.wns, w.ctypes.data, 8 * self.N))
        self.ck(self.lib.abcdez_memcpy_h2d(self.ctx, self.alive, a.ctypes.data, self.N))
        self.ck(self.lib.abcdez_ctx_set_uniform_weights(self.ctx, 1))      # Wns = 1/N: indicator kernels take the closed forms
        self.n_alive = self.n_prev = self.N

    def extrema(self):
        lo, hi = C.c_double(), C.c_double()
        self.ck(self.lib.abcdez_extrema(self.ctx, self.delta[self.cur], self.N, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def prologue(self, alpha, eps, eps_target, eps_k, ess_min):
        self.bind_stamps()
        en, q, wn, es, lo, hi = (C.c_double() for _ in range(6))
        na, part = C.c_int64(), C.c_int32()
        self.ck(self.lib.abcdez_smc_prologue_packed(
            self.ctx, self.delta[self.cur], self.wns, self.alive, self.N, self.n_prev, alpha, eps, eps_target, eps_k, ess_min,
            self.bits[self.bc], self.bits[1 - self.bc], self.slot[0], self.slot[1], self.logpi[self.cur], C.byref(en),
            C.byref(q), C.byref(wn), C.byref(es), C.byref(na), C.byref(part), C.byref(lo), C.byref(hi)))
        self.n_alive = na.value
        if part.value:
            self.n_prev = self.n_alive
        return en.value, wn.value, es.value, na.value, (lo.value, hi.value)

    def get_ess(self):
        e = C.c_double()
        self.ck(self.lib.abcdez_get_ess(self.ctx, self.wns, self.N, C.byref(e)))
        return e.value

    def resample(self):
        self.ck(self.lib.abcdez_wsample_stratified(self.ctx, self.wns, self.N, self.draw, self.inds))
        self.draw += 1
        o = self.other
        self.bind_stamps()
        self.ck(self.lib.abcdez_smc_resample_gather_packed(
            self.ctx, self.inds, self.N, self.bits[self.bc], self.bits[1 - self.bc], self.slot[0], self.slot[1],
            self.logpi[self.cur], self.delta[self.cur], self.logpi[o], self.delta[o], self.wns, self.alive))
        self.cur, self.n_alive, self.n_prev = o, self.N, self.N

    def smc_swarm(self, eps, g0, gs):
        nacc, nsim = C.c_int64(), C.c_int64()
        self.bind_stamps()
        self.ck(self.lib.abcdez_smc_swarm_packed(
            self.ctx, self.bits[self.bc], self.bits[1 - self.bc], self.n_alive, 0, self.n_alive, self.slot[0], self.slot[1],
            self.logpi[self.cur], self.delta[self.cur], None, eps, g0, gs, self.sweep, C.byref(nacc), C.byref(nsim)))
        self.sweep += 1
        self.bc = 1 - self.bc
        return nacc.value, nsim.value

    def smc_sweeps(self, eps, g0, gs, Kmcmc, Kmcmc_min, next_prologue=None):
        nacc, nsim, done = (C.c_int64 * Kmcmc)(), (C.c_int64 * Kmcmc)(), C.c_int32()
        self.bind_stamps()
        if next_prologue is not None:
            self.ck(self.lib.abcdez_smc_select_ahead(self.ctx, self.delta[self.cur], self.alive, self.N, next_prologue[0], next_prologue[1]))
        self.ck(self.lib.abcdez_smc_sweeps_packed(
            self.ctx, self.bits[self.bc], self.bits[1 - self.bc], self.n_alive, self.slot[0], self.slot[1],
            self.logpi[self.cur], self.delta[self.cur], eps, g0, gs, self.sweep, Kmcmc, Kmcmc_min, nacc, nsim, C.byref(done)))
        self.sweep += done.value
        if done.value & 1:
            self.bc = 1 - self.bc
        return sum(nacc), sum(nsim), done.value

    def mc_generation_issue(self, alpha, eps_target, g0, gs, lo_hi, do_rank):
        t = C.c_int64()
        o = self.other
        self.bind_stamps()
        lh = (C.c_double * 2)(*lo_hi) if lo_hi is not None else None
        self.ck(self.lib.abcdez_mc_generation_async(
            self.ctx, self.N, self.slot[self.cur], self.logpi[self.cur], self.delta[self.cur], self.slot[o], self.logpi[o],
            self.delta[o], self.order, self.sorted, self.cnt, alpha, eps_target, lh, 1 if do_rank else 0, g0, gs, self.sweep,
            C.byref(t)))
        self.sweep += 1
        self.cur = o
        return t.value

    def mc_generation_collect(self, ticket):
        nsim, ngt, lo, hi, ep = C.c_int64(), C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        self.ck(self.lib.abcdez_mc_generation_wait(self.ctx, ticket, C.byref(nsim), C.byref(ngt), C.byref(lo), C.byref(hi),
                                                   C.byref(ep)))
        return nsim.value, ngt.value, lo.value, hi.value

    def count_gt(self, thr):
        c = C.c_int64()
        self.ck(self.lib.abcdez_count_gt(self.ctx, self.delta[self.cur], self.N, thr, C.byref(c)))
        return c.value

    def rank_prepare(self, eps_pop, dmax):
        self.ck(self.lib.abcdez_mc_rank_prepare(self.ctx, self.delta[self.cur], self.N, eps_pop, dmax, self.order,
                                                self.sorted, self.cnt))

    def mc_swarm(self, eps_pop, eps_target, g0, gs):
        nsim, ngt, lo, hi = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        o = self.other
        self.bind_stamps()
        self.ck(self.lib.abcdez_mc_swarm(self.ctx, self.order, self.cnt, self.N, self.slot[self.cur], self.logpi[self.cur],
                                         self.delta[self.cur], self.slot[o], self.logpi[o], self.delta[o], eps_pop,
                                         eps_target, g0, gs, 0, self.N, self.sweep, C.byref(nsim), C.byref(ngt),
                                         C.byref(lo), C.byref(hi)))
        self.sweep += 1
        self.cur = o
        return nsim.value, ngt.value, lo.value, hi.value

    def download(self, packed):
        th = np.empty((self.N, self.ld))
        rows, pushed = self.devalloc(8 * self.N * self.ld), self.devalloc(8 * self.N * self.ld)
        if packed:
            self.ck(self.lib.abcdez_packed_gather(self.ctx, self.bits[self.bc], self.N, self.slot[0], self.slot[1], rows))
        src = rows if packed else self.slot[self.cur]
        self.ck(self.lib.abcdez_push_p(self.ctx, src, self.N, pushed))
        dl, w = np.empty(self.N), np.empty(self.N)
        # the shim's download: page-locked staging memory, three copies enqueued back to back, one wait (julia/ABCdeZHIP.jl)
        nrow = th.nbytes
        stage = C.c_void_p()
        self.ck(self.lib.abcdez_host_alloc(nrow + 16 * self.N, C.byref(stage)))
        try:
            self.ck(self.lib.abcdez_memcpy_d2h_async(self.ctx, stage.value, pushed, nrow))
            self.ck(self.lib.abcdez_memcpy_d2h_async(self.ctx, stage.value + nrow, self.delta[self.cur], 8 * self.N))
            self.ck(self.lib.abcdez_memcpy_d2h_async(self.ctx, stage.value + nrow + 8 * self.N, self.wns, 8 * self.N))
            self.ck(self.lib.abcdez_sync(self.ctx))
            C.memmove(th.ctypes.data, stage.value, nrow)
            C.memmove(dl.ctypes.data, stage.value + nrow, 8 * self.N)
            C.memmove(w.ctypes.data, stage.value + nrow + 8 * self.N, 8 * self.N)
        finally:
            self.lib.abcdez_host_free(stage)
        blobs = None
        if self.nb > 0:
            wd = C.c_int32()
            self.ck(self.lib.abcdez_blob_width(self.ctx, C.byref(wd)))
            bl, redo = self.devalloc(8 * self.N * wd.value), self.devalloc(8 * self.N)
            self.ck(self.lib.abcdez_blob_eval(self.ctx, src, self.stamp[self.cur], self.N, bl, redo))
            self.ck(self.lib.abcdez_sync(self.ctx))
            B, R = np.empty((self.N, wd.value)), np.empty(self.N)
            self.ck(self.lib.abcdez_memcpy_d2h(self.ctx, B.ctypes.data, bl, B.nbytes))
            self.ck(self.lib.abcdez_memcpy_d2h(self.ctx, R.ctypes.data, redo, 8 * self.N))
            assert np.array_equal(R.view(np.int64), dl.view(np.int64))     # the re-run distance is the stored one
            blobs = B[:, :self.nb]
            self.lib.abcdez_dev_free(bl); self.lib.abcdez_dev_free(redo)
        self.lib.abcdez_dev_free(rows); self.lib.abcdez_dev_free(pushed)
        return th[:, :self.d], w, dl, blobs

    def close(self):                                                # free!(e)
        for p in self.slot + self.logpi + self.delta + self.bits + self.stamp + [self.wns, self.alive, self.inds, self.order,
                                                                                 self.sorted, self.cnt]:
            self.ck(self.lib.abcdez_dev_free(p))
        self.lib.abcdez_ctx_destroy(self.ctx)


def shim_abcdesmc(prior, sim, eps_target, N, seed, ABCk=A.IndicatorStrict0toϵ, alpha=0.95, dess=0.5, Kmcmc=3, Kmcmc_min=1.0,
                  nsims_max=10 ** 7):
    """function abcdesmc!(prior, dist!::DeviceSimulator, ...) of the shim"""
    e = ShimEngine(prior, sim, ABCk, seed, N)
    e.init()
    e.reset_weights()
    eps = eps_k = math.inf
    logZ, nsims, facc, Ki, iters = 0.0, 0, 1.0, Kmcmc, 0
    g0, gs = 2.38 / math.sqrt(2 * e.d), 1e-5
    eps_hist, ranges = [eps], [e.extrema()]
    while True:
        iters += 1
        eps, wnorm, ess, n_alive, range_prev = e.prologue(alpha, eps, eps_target, eps_k, N * dess)
        if iters > 1:
            ranges.append(range_prev)
        logZ += math.log(wnorm)
        naccs, Ki = 0, Kmcmc
        if n_alive > 0 and ess < N * dess:
            e.resample()
            ess = e.get_ess()
            n_alive = N
        if n_alive >= 3 and Kmcmc <= 16:
            naccs, nsim, Ki = e.smc_sweeps(eps, g0, gs, Kmcmc, Kmcmc_min, (alpha, eps_target) if eps > eps_target else None)
            nsims += nsim
        elif n_alive >= 3:
            for i in range(1, Kmcmc + 1):
                nacc, nsim = e.smc_swarm(eps, g0, gs)
                naccs += nacc
                nsims += nsim
                if naccs / n_alive >= Kmcmc_min:
                    Ki = i
                    break
        facc = naccs / (n_alive * Ki)
        eps_k = eps
        eps_hist.append(eps)
        if n_alive < 3 or eps <= eps_target or nsims >= nsims_max:
            break
    e.ck(e.lib.abcdez_smc_select_discard(e.ctx))                 # as the shim does when the run ends
    ranges.append(e.extrema())
    P, W, D, blobs = e.download(packed=True)
    e.close()
    return dict(P=P, Wns=W, C=D, logZ=logZ, iters=iters, nsims=nsims, eps_hist=eps_hist, ranges=ranges, blobs=blobs)


def shim_abcdemc(prior, sim, eps_target, N, seed, generations, ahead=4):
    e = ShimEngine(prior, sim, A.IndicatorStrict0toϵ, seed, N)
    e.init()
    nsims, g0, gs = 0, 2.38 / math.sqrt(2 * e.d), 1e-5
    lo, hi = e.extrema()
    tickets, converged = [], False
    for it in range(generations):
        tickets.append(e.mc_generation_issue(0.0, eps_target, g0, gs, (lo, hi) if it == 0 else None, not converged))
        while len(tickets) > (ahead if it < generations - 1 else 0):
            nsim, n_above, lo, hi = e.mc_generation_collect(tickets.pop(0))
            nsims += nsim
            converged = converged or hi <= eps_target
            if not tickets:                              # nothing in flight: the arrays are this generation's
                assert n_above == e.count_gt(eps_target) and (lo, hi) == e.extrema()
    conv = hi <= eps_target
    P, _, D, _ = e.download(packed=False)
    e.close()
    return dict(P=P, C=D, nsims=nsims, reached=conv)


CASES = {
    "normal1d": (A.Normal(0.0, math.sqrt(10.0)), A.Normal1D(3.0), 0.3, 3000),
    "mvn8": (A.Factored(*[A.Normal(0.0, 1.0)] * 8), A.MVNormal((1.0,) * 8), 2.5, 2048),
    "mixed": (A.Factored(A.Normal(1, 0.5), A.DiscreteUniform(1, 10)), A.NormalTimesDU(5.5), 0.05, 400),
    "socks": (A.Factored(A.NegativeBinomial(900 / 195, (900 / 195) / (30 + 900 / 195)), A.Beta(15, 2)), A.Socks(0, 11), 0.01, 1500),
    "blobs": (A.Factored(*[A.Normal(0.0, 1.0)] * 3), A.MVNormal((1.0, 0.5, 0.2), blobs=True), 1.0, 1000),
}


@pytest.mark.parametrize("name", list(CASES))
def test_shim_call_sequence_abcdesmc(oracle, name):
    prior, sim, eps, N = CASES[name]
    got = shim_abcdesmc(prior, sim, eps, N, seed=31)
    ref = oracle.run_abcdesmc(A.ModelSpec(prior, sim, seed=31), N, eps)
    assert got["iters"] == ref["iters"] and got["nsims"] == ref["nsims"] and got["logZ"] == ref["logZ"]
    assert np.array_equal(np.array(got["eps_hist"]), ref["eps_hist"])
    assert [r[0] for r in got["ranges"]] == list(ref["lo_hist"]) and [r[1] for r in got["ranges"]] == list(ref["hi_hist"])
    if name == "blobs":            # the product host returns the same blobs (and checks them against the distances)
        r = A.abcdesmc(prior, sim, eps, None, nparticles=N, verbose=False, rng=31)
        assert np.array_equal(r.blobs, got["blobs"]) and np.array_equal(r.C, got["C"])
    assert np.array_equal(got["C"], ref["C"]) and np.array_equal(got["Wns"], ref["Wns"])
    m = oracle.OracleModel(A.ModelSpec(prior, sim, seed=31))
    th = np.ascontiguousarray(np.pad(ref["theta"], ((0, 0), (0, A.ModelSpec(prior, sim).ld - ref["theta"].shape[1]))))
    out = np.empty_like(th)
    oracle.lib().orc_push_p(m.ptr, th.ctypes.data, N, out.ctypes.data)
    assert np.array_equal(got["P"], out[:, :ref["theta"].shape[1]])                    # P is push_p-cast (smc:382)


@pytest.mark.parametrize("name", ["normal1d", "mvn8"])
def test_shim_call_sequence_abcdemc(oracle, name):
    prior, sim, eps, N = CASES[name]
    got = shim_abcdemc(prior, sim, eps, N, seed=32, generations=30)
    ref = oracle.run_abcdemc(A.ModelSpec(prior, sim, seed=32), N, eps, 30)
    assert got["nsims"] == ref["nsims"] and got["reached"] == ref["reached_eps"]
    assert np.array_equal(got["C"], ref["C"]) and np.array_equal(got["P"], ref["theta"])
