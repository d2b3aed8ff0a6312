"""abcdez_amd -- MI355X-native engine for ABCdeZ.jl's per-generation population loop.

Drop-in surface (reference: src/ABCdeZ.jl:9,15,18 exports ``Factored``, ``abcdemc!``,
``abcdesmc!``; Python identifiers cannot carry ``!``):

    from abcdez_amd import Factored, Normal, abcdesmc, abcdemc, Normal1D
    r = abcdesmc(Normal(0, 10 ** 0.5), Normal1D(3.0), 0.3, None, nparticles=1000)

The hot path runs only as HIP kernels through ``libabcdez_hip.so``; there is no CPU
fallback (importing the engine without the library / a GPU raises).
"""
from .kernels import (  # noqa: F401
    ALL_KERNELS, Epa0toeps, Epa0toϵ, EpaStrict0toeps, EpaStrict0toϵ, Indicator0toeps, Indicator0toϵ,
    IndicatorStrict0toeps, IndicatorStrict0toϵ,
)
from .model import ModelSpec  # noqa: F401
from .priors import (Affine, Beta, Binomial, Cauchy, Chisq, DiscreteUniform, Erlang, Exponential, Factored, Gamma, Geometric,  # noqa: F401
                     InverseGamma, Laplace, Logistic, LogNormal, MixtureModel, MvNormal, NegativeBinomial, Normal, Pareto, Poisson, Product,
                     Rayleigh, TDist, Truncated, TruncatedNormal, Uniform, Weibull, prior_cdf, product_distribution, push_p, truncated)
from .simulators import (  # noqa: F401
    DeviceSimulator, DiracSquare, LotkaVolterraRK4, Mixture01, MVNormal, Normal1D, NormalTimesDU, Quad2D, Socks, UserSimulator, WienerRMS,
)
from .smc import abcdesmc, get_ess, load_checkpoint, quantile_type7, save_checkpoint, wsample_stratified  # noqa: F401
from .mc import abcdemc  # noqa: F401

__version__ = "0.1.0"
