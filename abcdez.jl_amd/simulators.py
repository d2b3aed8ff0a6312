"""Device simulators: what goes in the ``dist!`` position of ``abcdesmc`` / ``abcdemc``.

The reference calls an arbitrary Julia closure ``dist!(θ, ve) -> (d, blob)``
(src/abcdez_smc.jl:137, src/abcdez_mc.jl:45, src/abcdez_init.jl:10,17).  A closure
cannot run on the GPU, so the drop-in dispatches on a :class:`DeviceSimulator`
marker that selects one of the built-in on-device simulators (ids and exact
arithmetic: include/abcdez_spec.h, ``ABZ_SIM_*``).

``blob`` -- the second return value, "arbitrary data stored to each particle, e.g. the actual
simulation output" (docs/src/index.md:298-324) -- is ``None`` by default, as in every reference
test.  ``blobs=True`` on a simulator makes the result's ``blobs`` the simulated data behind each
particle's final distance (one row of ``blob_size`` doubles per particle): the population carries
an 8-byte stamp naming the simulator call that produced the distance and the data are rebuilt
from it when the result is read (include/abcdez_hip.h, ``abcdez_blob_eval``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence, Tuple

SIM_NORMAL1D, SIM_MVN, SIM_DIRAC, SIM_QUAD2D, SIM_MIXTURE, SIM_NORMDU, SIM_WIENER, SIM_LV, SIM_SOCKS, SIM_USER = range(10)


class DeviceSimulator:
    sim_id = -1
    ndim = None  # required length(prior), None = any
    blobs = False

    def blob_size(self, d: int) -> int:
        """doubles per blob when ``blobs`` is on (abz_sim_blob_size, include/abcdez_spec.h)"""
        return 1

    def params(self) -> Tuple[float, ...]:
        return ()

    def data(self) -> Sequence[float]:
        return ()


@dataclass(frozen=True)
class Normal1D(DeviceSimulator):
    """x ~ N(θ, sigma); dist = |x − data|  (examples/minimal_example.jl:10-24)."""

    data_value: float
    sigma: float = 1.0
    blobs: bool = False            # blob = the simulated x
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
    blobs: bool = False            # blob = the simulated vector x
    sim_id = SIM_MVN

    def blob_size(self, d):
        return d

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
    blobs: bool = False            # blob = θ² + 1
    sim_id = SIM_DIRAC
    ndim = 1

    def params(self):
        return (float(self.target),)


@dataclass(frozen=True)
class Quad2D(DeviceSimulator):
    """50(x + 0.01n₁ − y²)² + (y − 1 + 0.01n₂)², +Inf with prob ``p_inf`` (test/runtests.jl:603,614)."""

    p_inf: float = 0.0
    blobs: bool = False            # blob = the two residuals (x + 0.01n₁ − y², y − 1 + 0.01n₂)
    sim_id = SIM_QUAD2D
    ndim = 2

    def blob_size(self, d):
        return 2

    def params(self):
        return (float(self.p_inf),)


@dataclass(frozen=True)
class Mixture01(DeviceSimulator):
    """x = θ + (coin ? 0.1n₁ : n₂); dist = |x − target|  (test/runtests.jl:582-583)."""

    target: float = 0.0
    blobs: bool = False            # blob = the simulated x
    sim_id = SIM_MIXTURE
    ndim = 1

    def params(self):
        return (float(self.target),)


@dataclass(frozen=True)
class NormalTimesDU(DeviceSimulator):
    """x = (n² + du)(n + 0.01n₁); dist = |x − target|  (test/runtests.jl:524-525)."""

    target: float = 5.5
    blobs: bool = False            # blob = the simulated x
    sim_id = SIM_NORMDU
    ndim = 2

    def params(self):
        return (float(self.target),)


@dataclass(frozen=True)
class WienerRMS(DeviceSimulator):
    """rms_t = sqrt(μ²t² + σ²t)(0.95 + 0.1u), t = 0..len(tdata)−1; mean |rms − tdata|  (test/runtests.jl:537-546)."""

    tdata: Tuple[float, ...]
    blobs: bool = False            # blob = the simulated rms_t series
    sim_id = SIM_WIENER
    ndim = 2

    def blob_size(self, d):
        return len(self.tdata)

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
    blobs: bool = False            # blob = the noisy observations (x0, y0, x1, y1, ...)
    sim_id = SIM_LV
    ndim = 4

    def blob_size(self, d):
        return len(self.obs)

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
    blobs: bool = False            # blob = (pairs, odd socks) of the simulated pick
    sim_id = SIM_SOCKS
    ndim = 2

    def blob_size(self, d):
        return 2

    def params(self):
        return (float(self.pairs), float(self.odds), float(self.n_picked))


class UserSimulator(DeviceSimulator):
    """A simulator supplied as HIP source text -- the device counterpart of the reference's
    ``dist!(θ, ve)`` closure (src/abcdez_smc.jl:137).  Compiled with hiprtc when the engine is created.  ``source`` defines ONE of

    * ``length(prior) <= 16`` -- the whole row in one thread::

        __device__ double abz_user_dist(const double* theta, int d, const double* data, int n_data,
                                        const double* sim_p, abz_user_rng& rng);

    * ``17 <= length(prior) <= 256`` -- the row spread over the lanes of a wavefront (8 components on each of 4 or 8 lanes up to 64
      parameters, 16 or 32 components on each of 8 lanes beyond), every lane calling::

        __device__ double abz_user_dist_lanes(const double* theta, const abz_user_lanes& g, int d, const double* data,
                                              int n_data, const double* sim_p, abz_user_rng& rng);

      with ITS ``ABZ_USER_C`` components in ``theta``; ``g.comp(q)`` is the index in the row of ``theta[q]`` (indices ``>= d``
      are padding), ``g.sum(v)`` adds ``v[0 .. C)`` over the whole group in one canonical tree (the same value on every lane and for
      every lane count), and the function returns the distance on every lane.  Draws are addressed, not sequential:
      ``rng.normal_pair_at(k, z0, z1)``, ``rng.uniform_at(k)`` -- key them by component (``g.comp(q) / 2`` for a pair), never by lane.

    * the STAGED form (any ``length(prior) <= 16``; it pays off from 3 parameters on: those rows sweep in two launches)::

        #define ABZ_USER_ROUNDS 8      // the simulation in this many steps
        #define ABZ_USER_STATE 3       // doubles carried between steps (<= 8, zero before round 0)
        __device__ double abz_user_round(const double* theta, int d, const double* data, int n_data, const double* sim_p,
                                         abz_user_rng& rng, int round, double* state);

      returning after every step a LOWER BOUND of the final distance that never decreases (the last round returns the distance):
      a proposal whose bound has passed ϵ is rejected for certain and leaves the simulation early, as the built-in
      Lotka–Volterra simulator's proposals do; results are bit for bit those of running every round (csrc/abz_user_rounds.h).

    ``theta`` arrives ``push_p``-cast; ``data`` / ``sim_p`` are the arrays given here (``varexternal``'s role);
    ``rng.uniform()``, ``rng.normal()``, ``rng.normal_pair(z0, z1)``, ``rng.bits()`` draw from the particle's
    counter-based stream.

    Blobs (``length(prior) <= 16``): with ``n_blob`` > 0 the source must also define::

        __device__ void abz_user_blob(const double* theta, int d, const double* data, int n_data,
                                      const double* sim_p, abz_user_rng& rng, double* blob, int n_blob);

    which receives a fresh copy of the SAME random stream as the ``abz_user_dist`` call that produced the particle's
    distance, so it can repeat the simulation and write its output to ``blob[0 .. n_blob)``.
    """

    sim_id = SIM_USER
    ndim = None

    def __init__(self, source: str, params: Sequence[float] = (), data: Sequence[float] = (), n_blob: int = 0):
        if "abz_user_dist" not in source and "abz_user_round" not in source:
            raise ValueError("the source must define abz_user_dist (rows of up to 16 parameters), abz_user_dist_lanes (17 .. 256) or "
                             "abz_user_round (the staged form)")
        if not 0 <= int(n_blob) <= 64:
            raise ValueError("n_blob must be in 0..64")
        if n_blob and "abz_user_blob" not in source:
            raise ValueError("n_blob > 0: the source must define abz_user_blob")
        self.n_blob = int(n_blob)
        self.blobs = self.n_blob > 0
        if len(params) > 8:
            raise ValueError("at most 8 scalar parameters (sim_p)")
        self.source = str(source)
        self._params = tuple(float(v) for v in params)
        self._data = tuple(float(v) for v in data)

    def blob_size(self, d):
        return self.n_blob

    def params(self):
        return self._params

    def data(self):
        return self._data
