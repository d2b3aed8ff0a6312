"""Device simulators: what goes in the ``dist!`` position of ``abcdesmc`` / ``abcdemc``.

The reference calls an arbitrary Julia closure ``dist!(θ, ve) -> (d, blob)``
(src/abcdez_smc.jl:137, src/abcdez_mc.jl:45, src/abcdez_init.jl:10,17).  A closure
cannot run on the GPU, so the drop-in dispatches on a :class:`DeviceSimulator`
marker that selects one of the built-in on-device simulators (ids and exact
arithmetic: csrc/abcdez_spec.h, ``ABZ_SIM_*``).  ``blob`` is always ``None``
(component 20).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence, Tuple

SIM_NORMAL1D, SIM_MVN, SIM_DIRAC, SIM_QUAD2D, SIM_MIXTURE, SIM_NORMDU, SIM_WIENER, SIM_LV, SIM_SOCKS, SIM_USER = range(10)


class DeviceSimulator:
    sim_id = -1
    ndim = None  # required length(prior), None = any

    def params(self) -> Tuple[float, ...]:
        return ()

    def data(self) -> Sequence[float]:
        return ()


@dataclass(frozen=True)
class Normal1D(DeviceSimulator):
    """x ~ N(θ, sigma); dist = |x − data|  (examples/minimal_example.jl:10-24)."""

    data_value: float
    sigma: float = 1.0
    sim_id = SIM_NORMAL1D
    ndim = 1

    def params(self):
        return (float(self.sigma),)

    def data(self):
        return (float(self.data_value),)


@dataclass(frozen=True)
class MVNormal(DeviceSimulator):
    """x = θ + sigma·z, z ~ N(0, I_d); dist = ‖x − y‖₂  (BASELINE.json config 3)."""

    y: Tuple[float, ...]
    sigma: float = 1.0
    sim_id = SIM_MVN

    def __post_init__(self):
        object.__setattr__(self, "y", tuple(float(v) for v in self.y))
        object.__setattr__(self, "ndim", len(self.y))

    def params(self):
        return (float(self.sigma),)

    def data(self):
        return self.y


@dataclass(frozen=True)
class DiracSquare(DeviceSimulator):
    """deterministic dist = |θ² + 1 − target|  (test/runtests.jl:495-497)."""

    target: float = 1.5
    sim_id = SIM_DIRAC
    ndim = 1

    def params(self):
        return (float(self.target),)


@dataclass(frozen=True)
class Quad2D(DeviceSimulator):
    """50(x + 0.01n₁ − y²)² + (y − 1 + 0.01n₂)², +Inf with prob ``p_inf`` (test/runtests.jl:603,614)."""

    p_inf: float = 0.0
    sim_id = SIM_QUAD2D
    ndim = 2

    def params(self):
        return (float(self.p_inf),)


@dataclass(frozen=True)
class Mixture01(DeviceSimulator):
    """x = θ + (coin ? 0.1n₁ : n₂); dist = |x − target|  (test/runtests.jl:582-583)."""

    target: float = 0.0
    sim_id = SIM_MIXTURE
    ndim = 1

    def params(self):
        return (float(self.target),)


@dataclass(frozen=True)
class NormalTimesDU(DeviceSimulator):
    """x = (n² + du)(n + 0.01n₁); dist = |x − target|  (test/runtests.jl:524-525)."""

    target: float = 5.5
    sim_id = SIM_NORMDU
    ndim = 2

    def params(self):
        return (float(self.target),)


@dataclass(frozen=True)
class WienerRMS(DeviceSimulator):
    """rms_t = sqrt(μ²t² + σ²t)(0.95 + 0.1u), t = 0..len(tdata)−1; mean |rms − tdata|  (test/runtests.jl:537-546)."""

    tdata: Tuple[float, ...]
    sim_id = SIM_WIENER
    ndim = 2

    def __post_init__(self):
        object.__setattr__(self, "tdata", tuple(float(v) for v in self.tdata))

    def data(self):
        return self.tdata


@dataclass(frozen=True)
class LotkaVolterraRK4(DeviceSimulator):
    """Lotka–Volterra by classical RK4 on device (BASELINE.json config 4).

    θ = (a, b, c, e); x' = ax − bxy, y' = −cy + exy from (x0, y0); ``steps_per_obs``
    RK4 steps of ``dt`` between observations; observed (x, y) pairs with additive
    N(0, noise²) noise compared to ``obs`` (flat x0,y0,x1,y1,…) by Euclidean distance.
    """

    obs: Tuple[float, ...]
    x0: float = 1.0
    y0: float = 0.5
    dt: float = 0.01
    steps_per_obs: int = 100
    noise: float = 0.1
    sim_id = SIM_LV
    ndim = 4

    def __post_init__(self):
        object.__setattr__(self, "obs", tuple(float(v) for v in self.obs))
        if len(self.obs) % 2 or not self.obs:
            raise ValueError("obs must hold (x, y) pairs")

    def params(self):
        return (float(self.x0), float(self.y0), float(self.dt), float(self.steps_per_obs), float(self.noise))

    def data(self):
        return self.obs


@dataclass(frozen=True)
class Socks(DeviceSimulator):
    """"Tiny data, ABC and the socks of Karl Broman" (test/runtests.jl:427-437): θ = (n_socks, prop_pairs);
    pick ``n_picked`` socks at random, count pairs and odd socks; dist = |pairs − data[0]| + |odds − data[1]|."""

    pairs: float = 0.0
    odds: float = 11.0
    n_picked: int = 11
    sim_id = SIM_SOCKS
    ndim = 2

    def params(self):
        return (float(self.pairs), float(self.odds), float(self.n_picked))


class UserSimulator(DeviceSimulator):
    """A simulator supplied as HIP source text -- the device counterpart of the reference's
    ``dist!(θ, ve)`` closure.  ``source`` must define::

        __device__ double abz_user_dist(const double* theta, int d, const double* data, int n_data,
                                        const double* sim_p, abz_user_rng& rng);

    ``theta`` arrives ``push_p``-cast; ``data`` / ``sim_p`` are the arrays given here (``varexternal``'s role);
    ``rng.uniform()``, ``rng.normal()``, ``rng.normal_pair(z0, z1)``, ``rng.bits()`` draw from the particle's
    counter-based stream.  Compiled with hiprtc when the engine is created (length(prior) <= 16).
    """

    sim_id = SIM_USER
    ndim = None

    def __init__(self, source: str, params: Sequence[float] = (), data: Sequence[float] = ()):
        if "abz_user_dist" not in source:
            raise ValueError("the source must define abz_user_dist")
        if len(params) > 8:
            raise ValueError("at most 8 scalar parameters (sim_p)")
        self.source = str(source)
        self._params = tuple(float(v) for v in params)
        self._data = tuple(float(v) for v in data)

    def params(self):
        return self._params

    def data(self):
        return self._data
