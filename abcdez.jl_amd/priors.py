"""Prior types of the drop-in: univariate distributions, ``Factored`` and ``push_p``.

Host-side mirror of what ABCdeZ.jl takes from Distributions.jl plus its own
``Factored`` (reference: src/abcdez_priors.jl:18-61) and ``push_p``
(src/abcdez_types.jl:20-23).  The device only ever sees the per-dimension
descriptor produced by :func:`prior_descriptor` (family id, two parameters, the
precomputed log-normaliser and the discrete flag).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Sequence, Tuple, Union

# family ids -- keep in sync with include/abcdez_spec.h (ABZ_PRIOR_*)
PRIOR_PAD, PRIOR_NORMAL, PRIOR_UNIFORM, PRIOR_DUNIFORM, PRIOR_BETA, PRIOR_NEGBIN = 0, 1, 2, 3, 4, 5
(PRIOR_EXPONENTIAL, PRIOR_GAMMA, PRIOR_LOGNORMAL, PRIOR_CAUCHY, PRIOR_LAPLACE, PRIOR_WEIBULL, PRIOR_INVGAMMA, PRIOR_TRUNCNORMAL,
 PRIOR_LOGISTIC, PRIOR_TDIST, PRIOR_PARETO, PRIOR_POISSON, PRIOR_BINOMIAL) = range(6, 19)
# wrappers around the families above; their records live in the model's ext table (ABZ_PRIOR_TRUNCATED / ABZ_PRIOR_MIXTURE)
PRIOR_TRUNCATED, PRIOR_MIXTURE, PRIOR_AFFINE = 19, 20, 21
MAX_MIX = 16

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def _rint(x: float) -> float:
    """Julia ``round(Int, x)``: ties to even (src/abcdez_types.jl:23)."""
    return float(round(x))  # Python's round() is ties-to-even for floats


class UnivariateDistribution:
    """Base of the supported prior factors."""

    discrete = False
    family = PRIOR_PAD

    def descriptor(self) -> Tuple[int, int, float, float, float]:
        raise NotImplementedError

    def __len__(self) -> int:  # length(prior) for a univariate prior is 1
        return 1


@dataclass(frozen=True)
class Normal(UnivariateDistribution):
    mu: float = 0.0
    sigma: float = 1.0
    family = PRIOR_NORMAL

    def __post_init__(self):
        if not self.sigma > 0.0:
            raise ValueError("Normal: sigma must be positive")

    def logpdf(self, x: float) -> float:
        z = (x - self.mu) / self.sigma
        return -0.5 * z * z + (-math.log(self.sigma) - _HALF_LOG_2PI)

    def pdf(self, x: float) -> float:
        return math.exp(self.logpdf(x))

    def insupport(self, x: float) -> bool:
        return math.isfinite(x)

    def rand(self, rng) -> float:
        return self.mu + self.sigma * rng.standard_normal()

    def descriptor(self):
        return (PRIOR_NORMAL, 0, float(self.mu), float(self.sigma), -math.log(self.sigma) - _HALF_LOG_2PI)


@dataclass(frozen=True)
class Uniform(UnivariateDistribution):
    a: float = 0.0
    b: float = 1.0
    family = PRIOR_UNIFORM

    def __post_init__(self):
        if not self.b > self.a:
            raise ValueError("Uniform: need a < b")

    def insupport(self, x: float) -> bool:
        return self.a <= x <= self.b  # closed support, as Distributions.Uniform

    def logpdf(self, x: float) -> float:
        return -math.log(self.b - self.a) if self.insupport(x) else -math.inf

    def pdf(self, x: float) -> float:
        return 1.0 / (self.b - self.a) if self.insupport(x) else 0.0

    def rand(self, rng) -> float:
        return self.a + (self.b - self.a) * rng.random()

    def descriptor(self):
        return (PRIOR_UNIFORM, 0, float(self.a), float(self.b), -math.log(self.b - self.a))


@dataclass(frozen=True)
class DiscreteUniform(UnivariateDistribution):
    a: int = 0
    b: int = 1
    family = PRIOR_DUNIFORM
    discrete = True

    def __post_init__(self):
        if not self.b >= self.a:
            raise ValueError("DiscreteUniform: need a <= b")

    def insupport(self, x) -> bool:
        return self.a <= x <= self.b and float(x) == _rint(float(x))

    def logpdf(self, x) -> float:
        return -math.log(self.b - self.a + 1) if self.insupport(x) else -math.inf

    def pdf(self, x) -> float:
        return 1.0 / (self.b - self.a + 1) if self.insupport(x) else 0.0

    def rand(self, rng) -> int:
        return int(rng.integers(self.a, self.b + 1))

    def descriptor(self):
        return (PRIOR_DUNIFORM, 1, float(self.a), float(self.b), -math.log(self.b - self.a + 1))


@dataclass(frozen=True)
class Beta(UnivariateDistribution):
    """Beta(α, β) on [0, 1] (prior of the Socks problem, test/runtests.jl:445)."""

    alpha: float = 1.0
    beta: float = 1.0
    family = PRIOR_BETA

    def __post_init__(self):
        if not (self.alpha > 0 and self.beta > 0):
            raise ValueError("Beta: need α, β > 0")

    def _logB(self) -> float:
        return math.lgamma(self.alpha) + math.lgamma(self.beta) - math.lgamma(self.alpha + self.beta)

    def insupport(self, x: float) -> bool:
        return 0.0 <= x <= 1.0

    def logpdf(self, x: float) -> float:
        if not self.insupport(x):
            return -math.inf
        t1 = 0.0 if self.alpha == 1 else ((self.alpha - 1) * math.log(x) if x > 0 else -math.inf * (self.alpha - 1))
        t2 = 0.0 if self.beta == 1 else ((self.beta - 1) * math.log(1 - x) if x < 1 else -math.inf * (self.beta - 1))
        return t1 + t2 - self._logB()

    def pdf(self, x: float) -> float:
        return math.exp(self.logpdf(x))

    def rand(self, rng) -> float:
        return float(rng.beta(self.alpha, self.beta))

    def descriptor(self):
        return (PRIOR_BETA, 0, float(self.alpha), float(self.beta), -self._logB())


@dataclass(frozen=True)
class NegativeBinomial(UnivariateDistribution):
    """NegativeBinomial(r, p): failures before the r-th success, real r > 0 (Distributions.jl convention;
    prior of the Socks problem, test/runtests.jl:443-444)."""

    r: float = 1.0
    p: float = 0.5
    family = PRIOR_NEGBIN
    discrete = True

    def __post_init__(self):
        if not (self.r > 0 and 0 < self.p <= 1):
            raise ValueError("NegativeBinomial: need r > 0 and 0 < p <= 1")

    def insupport(self, x) -> bool:
        return x >= 0 and float(x) == _rint(float(x))

    def logpdf(self, x) -> float:
        if not self.insupport(x):
            return -math.inf
        k = float(x)
        return (math.lgamma(k + self.r) - math.lgamma(k + 1) - math.lgamma(self.r) + self.r * math.log(self.p)
                + (k * math.log1p(-self.p) if self.p < 1 else (0.0 if k == 0 else -math.inf)))

    def pdf(self, x) -> float:
        return math.exp(self.logpdf(x))

    def rand(self, rng) -> int:
        return int(rng.negative_binomial(self.r, self.p))

    def descriptor(self):
        return (PRIOR_NEGBIN, 1, float(self.r), float(self.p), self.r * math.log(self.p) - math.lgamma(self.r))

    def c1(self) -> float:
        return math.log1p(-self.p) if self.p < 1 else -math.inf


# ---- further univariate families of Distributions.jl, in its parametrisations (the reference takes any `Distribution` as prior:
# src/abcdez_smc.jl:165, 215; src/abcdez_mc.jl:102).  descriptor() = (family, discrete, p0, p1, c0, c1, reserved) as laid out in
# include/abcdez_spec.h (ABZ_PRIOR_*); logpdf() here is the host mirror, the device and the oracle evaluate the header.

class _Continuous(UnivariateDistribution):
    def pdf(self, x: float) -> float:
        return math.exp(self.logpdf(x))


class _Counting(UnivariateDistribution):
    discrete = True

    def pdf(self, x) -> float:
        return math.exp(self.logpdf(x))


@dataclass(frozen=True)
class Exponential(_Continuous):
    """Exponential(θ): scale θ (mean θ), support x ≥ 0."""
    theta: float = 1.0
    family = PRIOR_EXPONENTIAL

    def __post_init__(self):
        if not (self.theta > 0 and math.isfinite(self.theta)):
            raise ValueError("Exponential: need θ > 0")

    def insupport(self, x) -> bool:
        return 0.0 <= x < math.inf

    def logpdf(self, x) -> float:
        return -math.log(self.theta) - x / self.theta if self.insupport(x) else -math.inf

    def rand(self, rng) -> float:
        return float(rng.exponential(self.theta))

    def descriptor(self):
        return (PRIOR_EXPONENTIAL, 0, float(self.theta), 0.0, -math.log(self.theta), 1.0 / self.theta, 0.0)


@dataclass(frozen=True)
class Gamma(_Continuous):
    """Gamma(α, θ): shape α, scale θ, support x ≥ 0."""
    alpha: float = 1.0
    theta: float = 1.0
    family = PRIOR_GAMMA

    def __post_init__(self):
        if not (self.alpha > 0 and self.theta > 0 and math.isfinite(self.alpha) and math.isfinite(self.theta)):
            raise ValueError("Gamma: need α, θ > 0")

    def insupport(self, x) -> bool:
        return 0.0 <= x < math.inf

    def _c0(self) -> float:
        return -math.lgamma(self.alpha) - self.alpha * math.log(self.theta)

    def logpdf(self, x) -> float:
        if not self.insupport(x):
            return -math.inf
        a1 = self.alpha - 1.0
        t1 = 0.0 if a1 == 0 else (a1 * math.log(x) if x > 0 else -math.inf * a1)
        return t1 - x / self.theta + self._c0()

    def rand(self, rng) -> float:
        return float(rng.gamma(self.alpha, self.theta))

    def descriptor(self):
        return (PRIOR_GAMMA, 0, float(self.alpha), float(self.theta), self._c0(), 1.0 / self.theta, 0.0)


def Chisq(nu: float) -> Gamma:
    """Chisq(ν) = Gamma(ν/2, 2)"""
    return Gamma(0.5 * nu, 2.0)


def Erlang(k: int = 1, theta: float = 1.0) -> Gamma:
    """Erlang(k, θ) = Gamma(k, θ) with an integer shape"""
    if int(k) != k or k < 1:
        raise ValueError("Erlang: the shape must be a positive integer")
    return Gamma(float(k), theta)


@dataclass(frozen=True)
class LogNormal(_Continuous):
    """LogNormal(μ, σ): log x ~ Normal(μ, σ), support x > 0."""
    mu: float = 0.0
    sigma: float = 1.0
    family = PRIOR_LOGNORMAL

    def __post_init__(self):
        if not (self.sigma > 0 and math.isfinite(self.sigma) and math.isfinite(self.mu)):
            raise ValueError("LogNormal: need σ > 0")

    def insupport(self, x) -> bool:
        return 0.0 < x < math.inf

    def logpdf(self, x) -> float:
        if not self.insupport(x):
            return -math.inf
        lx = math.log(x)
        z = (lx - self.mu) / self.sigma
        return -0.5 * z * z - math.log(self.sigma) - _HALF_LOG_2PI - lx

    def rand(self, rng) -> float:
        return float(rng.lognormal(self.mu, self.sigma))

    def descriptor(self):
        return (PRIOR_LOGNORMAL, 0, float(self.mu), float(self.sigma), -math.log(self.sigma) - _HALF_LOG_2PI, 1.0 / self.sigma, 0.0)


@dataclass(frozen=True)
class Cauchy(_Continuous):
    """Cauchy(μ, σ): location μ, scale σ."""
    mu: float = 0.0
    sigma: float = 1.0
    family = PRIOR_CAUCHY

    def __post_init__(self):
        if not (self.sigma > 0 and math.isfinite(self.sigma) and math.isfinite(self.mu)):
            raise ValueError("Cauchy: need σ > 0")

    def insupport(self, x) -> bool:
        return math.isfinite(x)

    def logpdf(self, x) -> float:
        if not self.insupport(x):
            return -math.inf
        z = (x - self.mu) / self.sigma
        return -math.log(math.pi * self.sigma) - math.log1p(z * z)

    def rand(self, rng) -> float:
        return self.mu + self.sigma * float(rng.standard_cauchy())

    def descriptor(self):
        return (PRIOR_CAUCHY, 0, float(self.mu), float(self.sigma), -math.log(math.pi * self.sigma), 1.0 / self.sigma, 0.0)


@dataclass(frozen=True)
class Laplace(_Continuous):
    """Laplace(μ, θ): location μ, scale θ."""
    mu: float = 0.0
    theta: float = 1.0
    family = PRIOR_LAPLACE

    def __post_init__(self):
        if not (self.theta > 0 and math.isfinite(self.theta) and math.isfinite(self.mu)):
            raise ValueError("Laplace: need θ > 0")

    def insupport(self, x) -> bool:
        return math.isfinite(x)

    def logpdf(self, x) -> float:
        return -math.log(2.0 * self.theta) - abs(x - self.mu) / self.theta if self.insupport(x) else -math.inf

    def rand(self, rng) -> float:
        return float(rng.laplace(self.mu, self.theta))

    def descriptor(self):
        return (PRIOR_LAPLACE, 0, float(self.mu), float(self.theta), -math.log(2.0 * self.theta), 1.0 / self.theta, 0.0)


@dataclass(frozen=True)
class Weibull(_Continuous):
    """Weibull(α, θ): shape α, scale θ, support x ≥ 0."""
    alpha: float = 1.0
    theta: float = 1.0
    family = PRIOR_WEIBULL

    def __post_init__(self):
        if not (self.alpha > 0 and self.theta > 0 and math.isfinite(self.alpha) and math.isfinite(self.theta)):
            raise ValueError("Weibull: need α, θ > 0")

    def insupport(self, x) -> bool:
        return 0.0 <= x < math.inf

    def logpdf(self, x) -> float:
        if not self.insupport(x):
            return -math.inf
        a1 = self.alpha - 1.0
        t = x / self.theta
        t1 = 0.0 if a1 == 0 else (a1 * math.log(t) if t > 0 else -math.inf * a1)
        return math.log(self.alpha / self.theta) + t1 - t ** self.alpha

    def rand(self, rng) -> float:
        return self.theta * float(rng.weibull(self.alpha))

    def descriptor(self):
        return (PRIOR_WEIBULL, 0, float(self.alpha), float(self.theta), math.log(self.alpha / self.theta), 1.0 / self.theta, 0.0)


def Rayleigh(sigma: float = 1.0) -> Weibull:
    """Rayleigh(σ) = Weibull(2, √2 σ)"""
    return Weibull(2.0, math.sqrt(2.0) * sigma)


@dataclass(frozen=True)
class InverseGamma(_Continuous):
    """InverseGamma(α, θ): shape α, scale θ, support x > 0."""
    alpha: float = 1.0
    theta: float = 1.0
    family = PRIOR_INVGAMMA

    def __post_init__(self):
        if not (self.alpha > 0 and self.theta > 0 and math.isfinite(self.alpha) and math.isfinite(self.theta)):
            raise ValueError("InverseGamma: need α, θ > 0")

    def insupport(self, x) -> bool:
        return 0.0 < x < math.inf

    def _c0(self) -> float:
        return self.alpha * math.log(self.theta) - math.lgamma(self.alpha)

    def logpdf(self, x) -> float:
        if not self.insupport(x):
            return -math.inf
        return self._c0() - (self.alpha + 1.0) * math.log(x) - self.theta / x

    def rand(self, rng) -> float:
        return self.theta / float(rng.gamma(self.alpha, 1.0))

    def descriptor(self):
        return (PRIOR_INVGAMMA, 0, float(self.alpha), float(self.theta), self._c0(), 0.0, 0.0)


def _norm_cdf_diff(a: float, b: float) -> float:
    """Φ(b) − Φ(a) without cancellation in either tail"""
    r2 = math.sqrt(2.0)
    if a > 0:                              # right tail: difference of survival functions
        return 0.5 * (math.erfc(a / r2) - math.erfc(b / r2))
    if b < 0:
        return 0.5 * (math.erfc(-b / r2) - math.erfc(-a / r2))
    return 1.0 - 0.5 * math.erfc(-a / r2) - 0.5 * math.erfc(b / r2)


@dataclass(frozen=True)
class TruncatedNormal(_Continuous):
    """``truncated(Normal(μ, σ), lo, hi)``: the Normal restricted to [lo, hi] (either bound may be infinite).  The initial
    population is drawn by rejection from the parent Normal, so the interval must hold at least 1 % of its mass."""
    mu: float = 0.0
    sigma: float = 1.0
    lo: float = -math.inf
    hi: float = math.inf
    family = PRIOR_TRUNCNORMAL

    def __post_init__(self):
        if not (self.sigma > 0 and math.isfinite(self.sigma) and math.isfinite(self.mu)):
            raise ValueError("truncated(Normal): need σ > 0")
        if not self.lo < self.hi:
            raise ValueError("truncated(Normal): need lo < hi")
        if self.mass() < 0.01:
            raise ValueError(f"truncated(Normal): [lo, hi] holds {self.mass():.3g} of the Normal's mass; the device draws the initial "
                             "population by rejection from the parent and needs at least 0.01")

    def mass(self) -> float:
        return _norm_cdf_diff((self.lo - self.mu) / self.sigma, (self.hi - self.mu) / self.sigma)

    def insupport(self, x) -> bool:
        return self.lo <= x <= self.hi and math.isfinite(x)

    def _c0(self) -> float:
        return -math.log(self.sigma) - _HALF_LOG_2PI - math.log(self.mass())

    def logpdf(self, x) -> float:
        if not self.insupport(x):
            return -math.inf
        z = (x - self.mu) / self.sigma
        return -0.5 * z * z + self._c0()

    def rand(self, rng) -> float:
        while True:
            x = self.mu + self.sigma * float(rng.standard_normal())
            if self.lo <= x <= self.hi:
                return x

    def descriptor(self):
        return (PRIOR_TRUNCNORMAL, 0, float(self.mu), float(self.sigma), self._c0(), float(self.lo), float(self.hi))


def truncated(dist, lower: float = None, upper: float = None):
    """``truncated(d, lower, upper)`` of Distributions.jl: a Normal parent keeps its own family (ABZ_PRIOR_TRUNCNORMAL), any other
    univariate family is wrapped (:class:`Truncated`, ABZ_PRIOR_TRUNCATED)."""
    lo = -math.inf if lower is None else float(lower)
    hi = math.inf if upper is None else float(upper)
    if type(dist) is Normal:
        return TruncatedNormal(dist.mu, dist.sigma, lo, hi)
    return Truncated(dist, lo, hi)


@dataclass(frozen=True)
class Logistic(_Continuous):
    """Logistic(μ, θ): location μ, scale θ."""
    mu: float = 0.0
    theta: float = 1.0
    family = PRIOR_LOGISTIC

    def __post_init__(self):
        if not (self.theta > 0 and math.isfinite(self.theta) and math.isfinite(self.mu)):
            raise ValueError("Logistic: need θ > 0")

    def insupport(self, x) -> bool:
        return math.isfinite(x)

    def logpdf(self, x) -> float:
        if not self.insupport(x):
            return -math.inf
        z = abs(x - self.mu) / self.theta
        return -math.log(self.theta) - z - 2.0 * math.log1p(math.exp(-z))

    def rand(self, rng) -> float:
        return float(rng.logistic(self.mu, self.theta))

    def descriptor(self):
        return (PRIOR_LOGISTIC, 0, float(self.mu), float(self.theta), -math.log(self.theta), 1.0 / self.theta, 0.0)


@dataclass(frozen=True)
class TDist(_Continuous):
    """TDist(ν): Student's t with ν degrees of freedom."""
    nu: float = 1.0
    family = PRIOR_TDIST

    def __post_init__(self):
        if not (self.nu > 0 and math.isfinite(self.nu)):
            raise ValueError("TDist: need ν > 0")

    def insupport(self, x) -> bool:
        return math.isfinite(x)

    def _c0(self) -> float:
        return math.lgamma(0.5 * (self.nu + 1.0)) - math.lgamma(0.5 * self.nu) - 0.5 * math.log(self.nu * math.pi)

    def logpdf(self, x) -> float:
        if not self.insupport(x):
            return -math.inf
        return self._c0() - 0.5 * (self.nu + 1.0) * math.log1p(x * x / self.nu)

    def rand(self, rng) -> float:
        return float(rng.standard_t(self.nu))

    def descriptor(self):
        return (PRIOR_TDIST, 0, float(self.nu), 0.5 * (self.nu + 1.0), self._c0(), 1.0 / self.nu, 0.0)


@dataclass(frozen=True)
class Pareto(_Continuous):
    """Pareto(α, θ): shape α, scale θ, support x ≥ θ."""
    alpha: float = 1.0
    theta: float = 1.0
    family = PRIOR_PARETO

    def __post_init__(self):
        if not (self.alpha > 0 and self.theta > 0 and math.isfinite(self.alpha) and math.isfinite(self.theta)):
            raise ValueError("Pareto: need α, θ > 0")

    def insupport(self, x) -> bool:
        return self.theta <= x < math.inf

    def _c0(self) -> float:
        return math.log(self.alpha) + self.alpha * math.log(self.theta)

    def logpdf(self, x) -> float:
        return self._c0() - (self.alpha + 1.0) * math.log(x) if self.insupport(x) else -math.inf

    def rand(self, rng) -> float:
        return self.theta * (1.0 + float(rng.pareto(self.alpha)))

    def descriptor(self):
        return (PRIOR_PARETO, 0, float(self.alpha), float(self.theta), self._c0(), 0.0, 0.0)


@dataclass(frozen=True)
class Poisson(_Counting):
    """Poisson(λ), λ ≤ 700 (the device draws the initial population by inversion from exp(−λ))."""
    lam: float = 1.0
    family = PRIOR_POISSON

    def __post_init__(self):
        if not 0 < self.lam <= 700:
            raise ValueError("Poisson: need 0 < λ <= 700")

    def insupport(self, x) -> bool:
        return 0 <= x < 2.0 ** 52 and float(x) == _rint(float(x))

    def logpdf(self, x) -> float:
        if not self.insupport(x):
            return -math.inf
        k = float(x)
        return k * math.log(self.lam) - self.lam - math.lgamma(k + 1.0)

    def rand(self, rng) -> int:
        return int(rng.poisson(self.lam))

    def descriptor(self):
        return (PRIOR_POISSON, 1, float(self.lam), 0.0, -float(self.lam), math.log(self.lam), 0.0)


@dataclass(frozen=True)
class Binomial(_Counting):
    """Binomial(n, p), 0 < p < 1, with n log(1 − min(p, 1 − p)) > −700 (inversion from the thinner end)."""
    n: int = 1
    p: float = 0.5
    family = PRIOR_BINOMIAL

    def __post_init__(self):
        if int(self.n) != self.n or self.n < 1:
            raise ValueError("Binomial: n must be a positive integer")
        if not 0 < self.p < 1:
            raise ValueError("Binomial: need 0 < p < 1 (a degenerate prior has no device descriptor)")
        if self.n * math.log1p(-min(self.p, 1.0 - self.p)) <= -700:
            raise ValueError("Binomial: n too large for the device's inversion sampler (n log(1 - min(p, 1 - p)) must stay above -700)")

    def insupport(self, x) -> bool:
        return 0 <= x <= self.n and float(x) == _rint(float(x))

    def logpdf(self, x) -> float:
        if not self.insupport(x):
            return -math.inf
        k, n = float(x), float(self.n)
        return (math.lgamma(n + 1.0) - math.lgamma(k + 1.0) - math.lgamma(n - k + 1.0) + k * math.log(self.p)
                + (n - k) * math.log1p(-self.p))

    def rand(self, rng) -> int:
        return int(rng.binomial(int(self.n), self.p))

    def descriptor(self):
        n = float(self.n)
        return (PRIOR_BINOMIAL, 1, n, float(self.p), math.lgamma(n + 1.0) + n * math.log1p(-self.p),
                math.log(self.p) - math.log1p(-self.p), 0.0)


def Geometric(p: float = 0.5) -> NegativeBinomial:
    """Geometric(p) = NegativeBinomial(1, p): failures before the first success"""
    return NegativeBinomial(1.0, p)


# ---- cumulative distribution functions (hosts only: the mass of a truncation interval; the device never evaluates a cdf) ----
def _gammainc_p(a: float, x: float) -> float:
    """regularised lower incomplete gamma P(a, x): series below a + 1, Lentz continued fraction of Q above"""
    if x <= 0.0:
        return 0.0
    if math.isinf(x):
        return 1.0
    lg = a * math.log(x) - x - math.lgamma(a)
    if x < a + 1.0:
        ap, term, tot = a, 1.0 / a, 1.0 / a
        for _ in range(2000):
            ap += 1.0
            term *= x / ap
            tot += term
            if abs(term) < abs(tot) * 1e-17:
                break
        return min(1.0, tot * math.exp(lg))
    tiny = 1e-300
    b = x + 1.0 - a
    c, d = 1.0 / tiny, 1.0 / b
    h = d
    for i in range(1, 2000):
        an = -i * (i - a)
        b += 2.0
        d = an * d + b
        d = tiny if abs(d) < tiny else d
        c = b + an / c
        c = tiny if abs(c) < tiny else c
        d = 1.0 / d
        dl = d * c
        h *= dl
        if abs(dl - 1.0) < 1e-16:
            break
    return max(0.0, 1.0 - math.exp(lg) * h)


def _betainc(a: float, b: float, x: float) -> float:
    """regularised incomplete beta I_x(a, b) (Lentz continued fraction, with the symmetry that keeps it convergent)"""
    if x <= 0.0:
        return 0.0
    if x >= 1.0:
        return 1.0
    lbt = math.lgamma(a + b) - math.lgamma(a) - math.lgamma(b) + a * math.log(x) + b * math.log1p(-x)

    def cf(a, b, x):
        tiny = 1e-300
        qab, qap, qam = a + b, a + 1.0, a - 1.0
        c, d = 1.0, 1.0 - qab * x / qap
        d = tiny if abs(d) < tiny else d
        d = 1.0 / d
        h = d
        for m in range(1, 3000):
            m2 = 2 * m
            aa = m * (b - m) * x / ((qam + m2) * (a + m2))
            d = 1.0 + aa * d
            d = tiny if abs(d) < tiny else d
            c = 1.0 + aa / c
            c = tiny if abs(c) < tiny else c
            d = 1.0 / d
            h *= d * c
            aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2))
            d = 1.0 + aa * d
            d = tiny if abs(d) < tiny else d
            c = 1.0 + aa / c
            c = tiny if abs(c) < tiny else c
            d = 1.0 / d
            dl = d * c
            h *= dl
            if abs(dl - 1.0) < 1e-16:
                break
        return h
    if x < (a + 1.0) / (a + b + 2.0):
        return min(1.0, math.exp(lbt) * cf(a, b, x) / a)
    return max(0.0, 1.0 - math.exp(lbt) * cf(b, a, 1.0 - x) / b)


def _phi(z: float) -> float:
    return 0.5 * math.erfc(-z / math.sqrt(2.0))


def prior_cdf(dist, x: float) -> float:
    """P(X <= x) of a univariate family of this module"""
    x = float(x)
    if math.isnan(x):
        return math.nan
    t = type(dist)
    if t is Normal:
        return _phi((x - dist.mu) / dist.sigma)
    if t is Uniform:
        return min(1.0, max(0.0, (x - dist.a) / (dist.b - dist.a)))
    if t is DiscreteUniform:
        return min(1.0, max(0.0, (math.floor(x) - dist.a + 1.0) / (dist.b - dist.a + 1.0)))
    if t is Beta:
        return _betainc(dist.alpha, dist.beta, x)
    if t is NegativeBinomial:
        return 0.0 if x < 0 else (1.0 if math.isinf(x) else _betainc(dist.r, math.floor(x) + 1.0, dist.p))
    if t is Exponential:
        return 0.0 if x <= 0 else -math.expm1(-x / dist.theta)
    if t is Gamma:
        return _gammainc_p(dist.alpha, x / dist.theta)
    if t is LogNormal:
        return 0.0 if x <= 0 else _phi((math.log(x) - dist.mu) / dist.sigma)
    if t is Cauchy:
        return 0.5 + math.atan((x - dist.mu) / dist.sigma) / math.pi
    if t is Laplace:
        z = (x - dist.mu) / dist.theta
        return 0.5 * math.exp(z) if z < 0 else 1.0 - 0.5 * math.exp(-z)
    if t is Weibull:
        return 0.0 if x <= 0 else -math.expm1(-((x / dist.theta) ** dist.alpha))
    if t is InverseGamma:
        return 0.0 if x <= 0 else 1.0 - _gammainc_p(dist.alpha, dist.theta / x)
    if t is TruncatedNormal:
        lo = _phi((dist.lo - dist.mu) / dist.sigma)
        return min(1.0, max(0.0, (_phi((x - dist.mu) / dist.sigma) - lo) / dist.mass()))
    if t is Logistic:
        z = (x - dist.mu) / dist.theta
        return 1.0 / (1.0 + math.exp(-z)) if z >= 0 else math.exp(z) / (1.0 + math.exp(z))
    if t is TDist:
        if math.isinf(x):
            return 1.0 if x > 0 else 0.0
        tail = 0.5 * _betainc(0.5 * dist.nu, 0.5, dist.nu / (dist.nu + x * x))
        return 1.0 - tail if x >= 0 else tail
    if t is Pareto:
        return 0.0 if x <= dist.theta else 1.0 - (dist.theta / x) ** dist.alpha
    if t is Poisson:
        return 0.0 if x < 0 else (1.0 if math.isinf(x) else 1.0 - _gammainc_p(math.floor(x) + 1.0, dist.lam))
    if t is Binomial:
        k = math.floor(x)
        return 0.0 if k < 0 else (1.0 if k >= dist.n else _betainc(dist.n - k, k + 1.0, 1.0 - dist.p))
    raise TypeError(f"no cdf for {t.__name__}")


def _full7(f) -> tuple:
    """(family, discrete, p0, p1, c0, c1, reserved) of a base family, as ModelSpec lays it out"""
    q = tuple(f.descriptor())
    if len(q) == 7:
        return q
    fam, disc, p0, p1, c0 = q
    c1 = 1.0 / p1 if fam == PRIOR_NORMAL else (f.c1() if fam == PRIOR_NEGBIN else 0.0)
    return (fam, disc, p0, p1, c0, c1, 0.0)


class Truncated(UnivariateDistribution):
    """``truncated(d, lo, hi)`` of Distributions.jl for ANY univariate family of this module (ABZ_PRIOR_TRUNCATED): the parent's
    density inside [lo, hi] over the mass of the interval.  The device draws the initial population by rejection from the parent,
    so the interval must hold at least 1 % of the parent's mass."""
    family = PRIOR_TRUNCATED

    def __init__(self, parent, lo: float = -math.inf, hi: float = math.inf):
        if not isinstance(parent, UnivariateDistribution) or isinstance(parent, (Truncated, MixtureModel, Affine)):
            raise TypeError("truncated(): the parent must be one of the base univariate families")
        if not lo < hi:
            raise ValueError("truncated(): need lower < upper")
        self.parent, self.lo, self.hi = parent, float(lo), float(hi)
        self.discrete = bool(parent.discrete)
        below = prior_cdf(parent, math.ceil(self.lo) - 1.0 if self.discrete and math.isfinite(self.lo) else self.lo) if math.isfinite(self.lo) else 0.0
        if not self.discrete and math.isfinite(self.lo):
            below = prior_cdf(parent, self.lo)
        above = prior_cdf(parent, self.hi) if math.isfinite(self.hi) else 1.0
        self._mass = above - below
        if not self._mass >= 0.01:
            raise ValueError(f"truncated({type(parent).__name__}): [lo, hi] holds {self._mass:.3g} of the parent's mass; the device draws the "
                             "initial population by rejection from the parent and needs at least 0.01")
        self._logmass = math.log(self._mass)

    def mass(self) -> float:
        return self._mass

    def insupport(self, x) -> bool:
        return self.lo <= x <= self.hi and self.parent.insupport(x)

    def logpdf(self, x) -> float:
        return self.parent.logpdf(x) - self._logmass if self.insupport(x) else -math.inf

    def pdf(self, x) -> float:
        return math.exp(self.logpdf(x))

    def rand(self, rng):
        while True:
            x = self.parent.rand(rng)
            if self.lo <= x <= self.hi:
                return x

    def ext_record(self):
        """[lo, hi, log mass, parent descriptor (7)] -- what the descriptor's p0 points at in abz_model.ext"""
        return [self.lo, self.hi, self._logmass] + [float(v) for v in _full7(self.parent)]

    def descriptor_at(self, offset: int):
        return (PRIOR_TRUNCATED, int(self.discrete), float(offset), 0.0, 0.0, 0.0, 0.0)

    def __repr__(self) -> str:
        return f"truncated({self.parent!r}, {self.lo}, {self.hi})"


class Affine(UnivariateDistribution):
    """``μ + σ * d`` of Distributions.jl (``LocationScale`` / ``AffineDistribution``) for a continuous base family d and σ > 0
    (ABZ_PRIOR_AFFINE): logpdf(x) = d.logpdf((x − μ) / σ) − log σ; the initial population draws μ + σ · (a draw of d)."""
    family = PRIOR_AFFINE
    discrete = False

    def __init__(self, parent, mu: float = 0.0, sigma: float = 1.0):
        if not isinstance(parent, UnivariateDistribution) or isinstance(parent, (Truncated, MixtureModel, Affine)) or parent.discrete:
            raise TypeError("Affine: the parent must be one of the continuous base univariate families")
        if not (sigma > 0 and math.isfinite(sigma) and math.isfinite(mu)):
            raise ValueError("Affine: need a finite μ and σ > 0")
        self.parent, self.mu, self.sigma = parent, float(mu), float(sigma)

    def insupport(self, x) -> bool:
        return self.parent.insupport((x - self.mu) / self.sigma)

    def logpdf(self, x) -> float:
        return self.parent.logpdf((x - self.mu) * (1.0 / self.sigma)) - math.log(self.sigma)

    def pdf(self, x) -> float:
        return math.exp(self.logpdf(x))

    def rand(self, rng):
        return self.mu + self.sigma * self.parent.rand(rng)

    def ext_record(self):
        """[μ, σ, 1 / σ, log σ, parent descriptor (7)]"""
        return [self.mu, self.sigma, 1.0 / self.sigma, math.log(self.sigma)] + [float(v) for v in _full7(self.parent)]

    def descriptor_at(self, offset: int):
        return (PRIOR_AFFINE, 0, float(offset), 0.0, 0.0, 0.0, 0.0)

    def __repr__(self) -> str:
        return f"{self.mu} + {self.sigma} * {self.parent!r}"


class MixtureModel(UnivariateDistribution):
    """``MixtureModel(components, weights)`` of Distributions.jl with univariate components of this module's base families, all
    continuous or all discrete (ABZ_PRIOR_MIXTURE): logpdf = log-sum-exp of log w_j + logpdf_j; the initial population draws the
    component by inversion of the cumulative weights, then from that component."""
    family = PRIOR_MIXTURE

    def __init__(self, components, weights=None):
        comps = tuple(components)
        if not 1 <= len(comps) <= MAX_MIX:
            raise ValueError(f"MixtureModel: 1 .. {MAX_MIX} components")
        for c in comps:
            if not isinstance(c, UnivariateDistribution) or isinstance(c, (Truncated, MixtureModel, Affine)):
                raise TypeError("MixtureModel: components must be base univariate families (no nesting)")
        if len({bool(c.discrete) for c in comps}) != 1:
            raise TypeError("MixtureModel: the components must be all continuous or all discrete (push_p has one rule per parameter)")
        w = [1.0 / len(comps)] * len(comps) if weights is None else [float(v) for v in weights]
        if len(w) != len(comps) or not all(v > 0 for v in w) or abs(sum(w) - 1.0) > 1e-9:
            raise ValueError("MixtureModel: one positive weight per component, summing to 1")
        tot = math.fsum(w)
        self.components, self.weights = comps, tuple(v / tot for v in w)
        self.discrete = bool(comps[0].discrete)

    def insupport(self, x) -> bool:
        return any(c.insupport(x) for c in self.components)

    def logpdf(self, x) -> float:
        t = [math.log(w) + c.logpdf(x) for w, c in zip(self.weights, self.components)]
        m = max(t)
        if m == -math.inf:
            return -math.inf
        acc = 0.0
        for v in t:
            acc += math.exp(v - m)
        return m + math.log(acc)

    def pdf(self, x) -> float:
        return math.exp(self.logpdf(x))

    def rand(self, rng):
        u, cum = rng.random(), 0.0
        for w, c in zip(self.weights, self.components):
            cum += w
            if u < cum:
                return c.rand(rng)
        return self.components[-1].rand(rng)

    def ext_record(self):
        """K records [log w_j, cumulative weight, component descriptor (7)]"""
        out, cum = [], 0.0
        for j, (w, c) in enumerate(zip(self.weights, self.components)):
            cum = 1.0 if j == len(self.components) - 1 else cum + w
            out += [math.log(w), cum] + [float(v) for v in _full7(c)]
        return out

    def descriptor_at(self, offset: int):
        return (PRIOR_MIXTURE, int(self.discrete), float(len(self.components)), float(offset), 0.0, 0.0, 0.0)

    def __repr__(self) -> str:
        return f"MixtureModel({list(self.components)!r}, {list(self.weights)!r})"


class Factored:
    """Product of independent univariate priors (src/abcdez_priors.jl:18-21).

    >>> prior = Factored(Normal(0, 1), Uniform(-1, 1))
    """

    def __init__(self, *args: UnivariateDistribution):
        if not args:
            raise ValueError("Factored needs at least one distribution")
        for a in args:
            if not isinstance(a, UnivariateDistribution):
                raise TypeError("Factored takes univariate distributions")
        self.p: Tuple[UnivariateDistribution, ...] = tuple(args)

    def __len__(self) -> int:  # src/abcdez_priors.jl:61
        return len(self.p)

    def pdf(self, x: Sequence[float]) -> float:  # src/abcdez_priors.jl:27-33
        s = self.p[0].pdf(x[0])
        for i in range(1, len(self.p)):
            s *= self.p[i].pdf(x[i])
        return s

    def logpdf(self, x: Sequence[float]) -> float:  # src/abcdez_priors.jl:40-46
        s = self.p[0].logpdf(x[0])
        for i in range(1, len(self.p)):
            s += self.p[i].logpdf(x[i])
        return s

    def rand(self, rng) -> tuple:  # src/abcdez_priors.jl:53-54
        return tuple(p.rand(rng) for p in self.p)

    def __repr__(self) -> str:
        return "Factored(" + ", ".join(repr(p) for p in self.p) + ")"


class Product(Factored):
    """``product_distribution([...])`` of Distributions.jl in the ``prior`` position: a multivariate distribution over a
    VECTOR of independent univariate components (test/runtests.jl:45).  Same densities and draws as :class:`Factored`;
    what differs is ``push_p`` -- the reference broadcasts the WHOLE distribution over the vector
    (``push_p(density::Distribution, p) = push_p.(Ref(density), p)``, src/abcdez_types.jl:21), so a product with any
    continuous component casts every element to float, and only an all-discrete product rounds."""

    def __init__(self, dists: Sequence[UnivariateDistribution]):
        super().__init__(*tuple(dists))
        self.discrete = all(d.discrete for d in self.p)

    def rand(self, rng) -> list:
        return [p.rand(rng) for p in self.p]

    def __repr__(self) -> str:
        return "product_distribution([" + ", ".join(repr(p) for p in self.p) + "])"


def product_distribution(dists: Sequence[UnivariateDistribution]) -> Product:
    return Product(dists)


class _WhitenedNormal(Normal):
    """one component of a correlated Normal prior after whitening: Normal(0, 1) carrying -log L_kk - log(2 pi)/2"""

    def __init__(self, c0: float):
        object.__setattr__(self, "mu", 0.0)
        object.__setattr__(self, "sigma", 1.0)
        object.__setattr__(self, "_c0", float(c0))

    def descriptor(self):
        return (PRIOR_NORMAL, 0, 0.0, 1.0, self._c0)


class MvNormal(Product):
    """``Distributions.MvNormal(mu, Sigma)`` in the ``prior`` position: a multivariate prior that is NOT a product.  Particles
    are vectors, ``push_p`` casts every element to float (src/abcdez_types.jl:16,21).  With Sigma = L L^T (Cholesky),
    theta = mu + L z, z ~ N(0, I), and logpdf(theta) = sum_k [-z_k^2 / 2 - log L_kk - log(2 pi)/2] with z = L^-1 (theta - mu):
    the device evaluates the per-dimension Normal tree over the WHITENED components (include/abcdez_spec.h, abz_model.mv)."""

    def __init__(self, mu: Sequence[float], cov):
        import numpy as np

        self.mu = np.ascontiguousarray(np.asarray(mu, dtype=np.float64))
        self.cov = np.ascontiguousarray(np.asarray(cov, dtype=np.float64))
        d = self.mu.size
        if self.cov.shape != (d, d) or not np.allclose(self.cov, self.cov.T):
            raise ValueError("MvNormal: cov must be a symmetric d x d matrix matching mu")
        try:
            self.L = np.linalg.cholesky(self.cov)
        except np.linalg.LinAlgError as e:
            raise ValueError("MvNormal: cov must be positive definite") from e
        W = np.linalg.solve(self.L, np.eye(d))                 # L^-1, lower triangular up to rounding
        self.W = np.tril(W)
        super().__init__([_WhitenedNormal(-math.log(self.L[k, k]) - _HALF_LOG_2PI) for k in range(d)])
        self.discrete = False

    def mv_maps(self, ld: int):
        """[mu[ld] | W[ld][ld] | L[ld][ld]] row-major, zero-padded: what abz_model.mv points at"""
        import numpy as np

        d = self.mu.size
        out = np.zeros(ld + 2 * ld * ld)
        out[:d] = self.mu
        Wp, Lp = np.zeros((ld, ld)), np.zeros((ld, ld))
        Wp[:d, :d], Lp[:d, :d] = self.W, np.tril(self.L)
        out[ld:ld + ld * ld] = Wp.ravel()
        out[ld + ld * ld:] = Lp.ravel()
        return np.ascontiguousarray(out)

    def logpdf(self, x) -> float:
        import numpy as np

        z = self.W @ (np.asarray(x, dtype=np.float64) - self.mu)
        return float(sum(-0.5 * zk * zk + f._c0 for zk, f in zip(z, self.p)))

    def insupport(self, x) -> bool:
        return all(math.isfinite(float(v)) for v in x)

    def rand(self, rng) -> list:
        import numpy as np

        return list(self.mu + self.L @ rng.standard_normal(self.mu.size))

    def __repr__(self) -> str:
        return f"MvNormal(mu={self.mu.tolist()}, cov={self.cov.tolist()})"


Prior = Union[UnivariateDistribution, Factored]


def prior_factors(prior: Prior) -> Tuple[UnivariateDistribution, ...]:
    if isinstance(prior, Factored):
        return prior.p
    if isinstance(prior, UnivariateDistribution):
        return (prior,)
    raise TypeError(f"unsupported prior type {type(prior).__name__}: the device knows the univariate families of "
                    "abcdez_amd.priors (and Factored / product_distribution / MvNormal of them)")


def prior_length(prior: Prior) -> int:
    return len(prior_factors(prior))


def push_p(density, p):
    """Cast parameter values to the prior's domain (src/abcdez_types.jl:20-23).

    continuous -> ``float(p)``; discrete -> ``round(Int, p)`` (ties to even);
    ``Factored`` / sequences broadcast.
    """
    if isinstance(density, Product):           # the whole (multivariate) distribution is broadcast over the vector
        return [int(_rint(float(v))) if density.discrete else float(v) for v in p]
    if isinstance(density, Factored):
        return tuple(push_p(d, v) for d, v in zip(density.p, p))
    if isinstance(p, (tuple, list)):
        return type(p)(push_p(density, v) for v in p)
    if density.discrete:
        return int(_rint(float(p)))
    return float(p)
