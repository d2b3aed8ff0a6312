"""User-supplied simulators (hiprtc) beyond one thread per row: the COOPERATIVE form on rows of 17 .. 64 parameters and the STAGED
form whose proposals leave the simulation early (include/abcdez_hip.h: abcdez_ctx_create_user; csrc/abz_user_rounds.h).  Each restates
a built-in simulator, so the whole run must equal the built-in's -- hence the CPU oracle's -- bit for bit.  The reference calls any
dist!(theta, ve) for any length(prior): src/abcdez_smc.jl:137,166-173."""
import json
import math
import os

import numpy as np
import pytest
import torch

import abcdez_amd as A
from abcdez_amd import _lib
from abcdez_amd.engine import HipOps, PopulationEngine

from user_sources import USER_LV_ROUNDS, USER_MVN16, USER_MVN_LANES, USER_SEQ16, USER_SEQ16_ROUNDS

pytestmark = pytest.mark.gpu
GOLD_DIR = os.path.join(os.path.dirname(__file__), "golden")


def run_both(prior, user, builtin, N, eps, oracle, seed, generations=20):
    r = A.abcdesmc(prior, user, eps, None, nparticles=N, verbose=False, rng=seed, nsims_max=10 ** 10)
    c = oracle.run_abcdesmc(A.ModelSpec(prior, builtin, seed=seed), N, eps, nsims_max=10 ** 10)
    res = r.engine.result()
    assert r.logZ == c["logZ"] and r.nsims == c["nsims"] and r.iters == c["iters"]
    for k in ("theta", "C", "Wns"):
        assert np.array_equal(res[k], c[k]), k
    m = A.abcdemc(prior, user, eps, None, nparticles=N, generations=generations, verbose=False, rng=seed + 1)
    cm = oracle.run_abcdemc(A.ModelSpec(prior, builtin, seed=seed + 1), N, eps, generations)
    mres = m.engine.result()
    assert np.array_equal(mres["theta"], cm["theta"]) and np.array_equal(mres["C"], cm["C"])
    return r


@pytest.mark.parametrize("d", [17, 20, 32, 33, 48, 64, 100, 128, 256])
def test_cooperative_user_simulator_on_wide_rows_equals_the_builtin(oracle, d):
    """the d-dimensional Normal simulator restated as abz_user_dist_lanes (8 components per lane, 4 lanes at d <= 32, 8 at d <= 64; beyond,
    8 lanes of 16 or 32 components; padding components at d = 17, 20, 33, 48, 100): initial population, the two-phase sweep, partition, resampling and abcdemc through the
    run-time-compiled kernels -- whole runs bit-identical to the built-in simulator's oracle.  d = 32 with Normal(0, 1) priors is the
    headline shape (BASELINE.json configs[2])."""
    y = tuple(1.0 + (0.01 * k if d <= 64 else 0.0) for k in range(d))
    fams = [A.Normal(0, 1)] * d
    if d in (20, 48):             # a few of the further families among the priors: the family dispatch inside the user kernels
        fams = [A.Normal(0, 1)] * (d - 3) + [A.Gamma(2.0, 1.0), A.Uniform(-3, 4), A.Laplace(1.0, 1.0)]
    prior = A.Factored(*fams)
    builtin = A.MVNormal(y, sigma=1.0)
    user = A.UserSimulator(USER_MVN_LANES, params=(1.0,), data=y)
    eps = (0.8 if d <= 64 else 0.92) * math.sqrt(3.0 * d)      # below the distance's prior median sqrt(3 d): a dozen generations and more
    r = run_both(prior, user, builtin, 4096, eps, oracle, seed=41)
    assert r.iters >= 3 and int((r.Wns > 0).sum()) > 100          # a run that got somewhere
    ld, L, Cc = r.engine.ops.layout()
    want_ld = 32 if d <= 32 else (64 if d <= 64 else (128 if d <= 128 else 256))
    assert ld == want_ld and (L, Cc) == ((4, 8) if d <= 32 else (8, want_ld // 8))


def test_cooperative_form_is_what_a_wide_row_needs():
    """a source that only defines the one-thread form cannot serve 32 parameters: the compile error names the missing function"""
    src = "__device__ double abz_user_dist(const double* t, int d, const double* a, int n, const double* p, abz_user_rng& r) { return 0.0; }"
    with pytest.raises(_lib.AbcdezError, match="abz_user_dist_lanes"):
        A.abcdesmc(A.Factored(*[A.Normal(0, 1)] * 32), A.UserSimulator(src), 1.0, None, nparticles=4096, verbose=False, rng=1)


def lv_small():
    obs = (1.0, 0.5, 1.46, 0.43, 1.77, 0.62, 1.52, 1.13, 0.95, 1.31, 0.66, 1.09, 0.61, 0.79, 0.75, 0.6)
    builtin = A.LotkaVolterraRK4(obs, dt=0.05, steps_per_obs=10)
    return A.Factored(*[A.Uniform(0.0, 2.0)] * 4), builtin, (builtin.x0, builtin.y0, builtin.dt, float(builtin.steps_per_obs), builtin.noise), obs


@pytest.mark.parametrize("rounds", [1, 3, 8])
@pytest.mark.parametrize("sweep", ["two launches", "one kernel"])
def test_staged_user_simulator_equals_the_builtin(oracle, rounds, sweep, monkeypatch):
    """Lotka-Volterra restated as abz_user_round (1, 3 or 8 rounds over its 8 observations; state = x, y, the running sum of squared
    errors, whose square root bounds the distance from below): in the two-launch sweep the proposals whose bound has passed eps leave
    early and the survivors are re-packed; ABZ_USER_ONE_KERNEL=1 runs every round back to back through the generated abz_user_dist.
    Either way the whole run -- and abcdemc, the initial population -- equals the built-in simulator's oracle bit for bit."""
    monkeypatch.setenv("ABZ_USER_ONE_KERNEL", "1" if sweep == "one kernel" else "0")
    prior, builtin, params, obs = lv_small()
    user = A.UserSimulator(USER_LV_ROUNDS % {"rounds": rounds}, params=params, data=obs)
    run_both(prior, user, builtin, 6000, 1.2, oracle, seed=17, generations=10)


def test_staged_form_with_epanechnikov_kernel_and_infinite_first_eps(oracle):
    """the certain-rejection rule `bound > eps` under a continuous-weight kernel (types.jl:51-73: zero beyond eps as well) and in the
    first generation, whose eps may be anything up to Inf"""
    prior, builtin, params, obs = lv_small()
    user = A.UserSimulator(USER_LV_ROUNDS % {"rounds": 4}, params=params, data=obs)
    N, eps = 5000, 1.5
    r = A.abcdesmc(prior, user, eps, None, nparticles=N, ABCk=A.Epa0toϵ, verbose=False, rng=23, nsims_max=10 ** 10)
    c = oracle.run_abcdesmc(A.ModelSpec(prior, builtin, A.Epa0toϵ, seed=23), N, eps, nsims_max=10 ** 10)
    res = r.engine.result()
    assert r.logZ == c["logZ"] and r.nsims == c["nsims"] and np.array_equal(res["theta"], c["theta"]) and np.array_equal(res["Wns"], c["Wns"])


def test_staged_form_limits_are_compile_errors():
    prior, builtin, params, obs = lv_small()
    src = USER_LV_ROUNDS % {"rounds": 8}
    with pytest.raises(_lib.AbcdezError, match="ABZ_USER_STATE"):
        A.abcdesmc(prior, A.UserSimulator(src.replace("#define ABZ_USER_STATE 3", "#define ABZ_USER_STATE 9"), params=params, data=obs), 1.2, None,
                   nparticles=1024, verbose=False, rng=1)


# ---- 9 to 16 parameters: one lane per particle on rows of 16 doubles; the sweep in two launches (the second one 128 threads wide) ----
def prior_12():
    return A.Factored(A.Normal(0, 2), A.Gamma(2.0, 1.0), A.Uniform(-2, 3), A.truncated(A.Normal(1.0, 2.0), 0.0, None), A.Normal(0, 1),
                      A.LogNormal(0.0, 0.7), *[A.Normal(0.5, 1.5)] * 6)


Y12 = (1.0, 0.5, 0.8, 1.2, 0.3, 1.5, 0.2, 0.9, 1.1, 0.4, 0.7, 0.6)


@pytest.mark.parametrize("d", [9, 12, 16])
@pytest.mark.parametrize("sweep", ["two launches", "one kernel"])
def test_user_simulator_of_nine_to_sixteen_parameters_equals_the_builtin(oracle, sweep, d, monkeypatch):
    """the d-dimensional Normal simulator restated as one opaque abz_user_dist over a row of 16 doubles (d = 16: no padding
    component) equals the built-in one -- the oracle -- bit for bit: in the two-launch sweep (phase 2 over the dense list of the
    proposals that can still be accepted) and inside the one-kernel body (ABZ_USER_ONE_KERNEL=1); abcdemc and the initial
    population with it"""
    monkeypatch.setenv("ABZ_USER_ONE_KERNEL", "1" if sweep == "one kernel" else "0")
    factors = (prior_12().p + (A.Normal(0, 1),) * 4)[:d]
    y = (Y12 + (0.9, 0.1, 1.3, 0.5))[:d]
    prior = A.Factored(*factors)
    run_both(prior, A.UserSimulator(USER_MVN16, params=(0.8,), data=y), A.MVNormal(y, sigma=0.8), 6000, 0.62 * math.sqrt(d), oracle, seed=29,
             generations=10)


def _whole_run(prior, sim, N, eps, seed, **kw):
    r = A.abcdesmc(prior, sim, eps, None, nparticles=N, verbose=False, rng=seed, nsims_max=10 ** 10, **kw)
    res = r.engine.result()
    m = A.abcdemc(prior, sim, eps, None, nparticles=N, generations=8, verbose=False, rng=seed + 1)
    mres = m.engine.result()
    return (r.logZ, r.nsims, r.iters, res["theta"].copy(), res["C"].copy(), res["Wns"].copy(), mres["theta"].copy(), mres["C"].copy())


@pytest.mark.parametrize("kernel", ["indicator", "epanechnikov"])
def test_staged_form_on_rows_of_sixteen_doubles(kernel, monkeypatch):
    """ONE model of 12 parameters in the opaque and in the staged form (squared errors accumulated pair by pair: 1, 2, 4 or 8 rounds
    over the 8 pairs).  The anchor is the opaque form inside the one-kernel body -- the only path such a model had before the
    two-launch sweep took rows of 16 doubles; the opaque form in two launches and every staged form (proposals leaving after the
    round in which their running sum passes eps, survivors re-packed; 128 proposals per workgroup) must reproduce its whole run:
    evidence, counters, every particle, distance and weight, abcdemc's population."""
    prior = prior_12()
    kw = dict(ABCk=A.Epa0toϵ) if kernel == "epanechnikov" else {}
    N, eps, seed = 6000, 2.3, 31
    monkeypatch.setenv("ABZ_USER_ONE_KERNEL", "1")
    anchor = _whole_run(prior, A.UserSimulator(USER_SEQ16, params=(0.8,), data=Y12), N, eps, seed, **kw)
    assert anchor[2] > 5 and np.isfinite(anchor[0])
    monkeypatch.setenv("ABZ_USER_ONE_KERNEL", "0")
    forms = [USER_SEQ16] + [USER_SEQ16_ROUNDS % {"rounds": r} for r in (1, 2, 4, 8)]
    for src in forms:
        got = _whole_run(prior, A.UserSimulator(src, params=(0.8,), data=Y12), N, eps, seed, **kw)
        assert got[:3] == anchor[:3], (src[:60], got[:3], anchor[:3])
        for a, b in zip(got[3:], anchor[3:]):
            assert np.array_equal(a, b)


def test_user_simulator_example_script():
    """examples/user_simulator.py: an SIR epidemic written by a user in the staged form recovers the parameters its data were generated
    at, and equals the same model written as one opaque call bit for bit"""
    import importlib.util

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "user_simulator.py")
    spec = importlib.util.spec_from_file_location("user_simulator_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r, mean, sd, truth = mod.main(20000, verbose=False)
    o, _, _, _ = mod.main(20000, source=mod.SIR_OPAQUE, verbose=False)
    assert r.ϵ == 0.08 and r.iters > 20
    assert np.all(np.abs(mean - np.array(truth)) < 2.5 * sd), (mean, sd)
    assert sd[0] < 0.2 and sd[1] < 0.06                                  # the data are informative: far narrower than the priors
    assert r.logZ == o.logZ and r.nsims == o.nsims and np.array_equal(np.array(r.P), np.array(o.P)) and np.array_equal(r.Wns, o.Wns)
