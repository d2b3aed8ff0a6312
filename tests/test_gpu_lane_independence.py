"""Results must not depend on the lane-group shape (L lanes x C components per particle).

The sweep kernels evaluate the per-particle scalar draws -- donor ranks (src/abcdez_smc.jl:119-126), gamma
(smc:128), log(rand) of the accept test (smc:145) -- in two ways: spread over three lanes for L >= 4, straight for
L < 4.  Both must be the oracle's values bit for bit, so that an accept decision can never depend on L."""
import math

import numpy as np
import pytest
import torch

import abcdez_amd as A
from abcdez_amd.engine import HipOps, PopulationEngine

pytestmark = pytest.mark.gpu


def _spec(d, seed=5):
    prior = A.Factored(*[A.Normal(0.0, 1.0) for _ in range(d)])
    return A.ModelSpec(prior, A.MVNormal(tuple([1.0] * d)), seed=seed)


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16])
def test_particle_draws_equal_oracle_for_every_lane_width(oracle, lanes):
    """(ra, rb, gamma, log u) of 2^20 + 77 particles (ragged last block) for three sweeps, bit for bit"""
    spec = _spec(32)
    ops = HipOps(spec)
    orc = oracle.OracleOps(spec)
    n = (1 << 20) + 77
    n_pool = n + 1000
    g0 = 2.38 / math.sqrt(64)
    dev = dict(ra=torch.zeros(n, dtype=torch.int32, device="cuda"), rb=torch.zeros(n, dtype=torch.int32, device="cuda"),
               g=torch.zeros(n, dtype=torch.float64, device="cuda"), lu=torch.zeros(n, dtype=torch.float64, device="cuda"))
    host = {k: torch.zeros_like(v, device="cpu") for k, v in dev.items()}
    for sweep in (0, 7, 123456):
        ops.draws_eval(lanes, 500, n_pool, sweep, g0, 1e-5, dev["ra"], dev["rb"], dev["g"], dev["lu"])
        orc.draws_eval(lanes, 500, n_pool, sweep, g0, 1e-5, host["ra"], host["rb"], host["g"], host["lu"])
        assert torch.equal(dev["ra"].cpu(), host["ra"]) and torch.equal(dev["rb"].cpu(), host["rb"])
        assert torch.equal(dev["g"].cpu().view(torch.int64), host["g"].view(torch.int64))
        assert torch.equal(dev["lu"].cpu().view(torch.int64), host["lu"].view(torch.int64))
        a, b = host["ra"].numpy().view(np.uint32), host["rb"].numpy().view(np.uint32)
        own = np.arange(500, 500 + n, dtype=np.uint32)
        assert (a != own).all() and (b != own).all() and (a != b).all()      # distinct donors, smc:119-126
        assert (a < n_pool).all() and (b < n_pool).all()
        assert (host["lu"].numpy() < 0).all()


# (every lane shape of the dispatch table against the oracle population: tests/test_gpu_packed.py::test_packed_lane_shapes)


@pytest.mark.parametrize("name", ["normal1d", "lv"])
def test_single_lane_simulators_over_a_million_particles(oracle, name):
    """L = 1 simulators (whole row in one thread: the straight branch of particle_draws) at N = 2^20: one
    generation's reweight + compaction + sweep, every particle compared with the oracle"""
    if name == "normal1d":
        prior, sim = A.Normal(0.0, math.sqrt(10.0)), A.Normal1D(3.0)
    else:
        prior = A.Factored(*[A.Uniform(0.0, 2.0)] * 4)
        sim = A.LotkaVolterraRK4(tuple(np.linspace(0.5, 1.5, 8)), dt=0.1, steps_per_obs=5)
    spec = A.ModelSpec(prior, sim, seed=9)
    N = 1 << 20
    out = []
    for ops in (HipOps(spec), oracle.OracleOps(spec)):
        e = PopulationEngine(spec, N, ops=ops)
        e.init_population()
        e.reset_weights()
        eps = e.quantile_alive(0.8)
        e.smc_reweight(math.inf, eps)
        e.alive_compact()
        c = [e.smc_swarm(eps, 2.38 / math.sqrt(2 * spec.d), 1e-5) for _ in range(2)]
        out.append((c, [t.cpu() for t in e.state]))
    assert out[0][0] == out[1][0]
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.equal(a.view(torch.int64), b.view(torch.int64))
