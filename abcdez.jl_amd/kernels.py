"""The four unnormalised ABC kernels (reference: src/abcdez_types.jl:26-73).

Host-side value types; the device evaluates the same truth table from the kernel id
(include/abcdez_spec.h, ``abz_kernel_*``).  ``ABCk`` keyword of :func:`abcdesmc` takes
one of these classes, exactly like the reference (src/abcdez_smc.jl:218).
"""
from __future__ import annotations

import math

# ids -- keep in sync with include/abcdez_spec.h (ABZ_K_*)
K_INDICATOR, K_INDICATOR_STRICT, K_EPA, K_EPA_STRICT = 0, 1, 2, 3


class _ABCKernel:
    kind = -1
    strict = False
    epa = False

    def __init__(self, eps: float):
        eps = float(eps)
        if not eps >= 0.0:  # src/abcdez_types.jl:30,42,55,67
            raise ValueError("Expected ϵ ≥ 0.0")
        self.eps = eps

    # the reference's field is called ϵ; keep an ASCII alias and the unicode name
    @property
    def ϵ(self) -> float:  # noqa: PLC2401
        return self.eps

    def insupport(self, x: float) -> bool:
        if not 0.0 <= x:
            return False
        return x < self.eps if self.strict else x <= self.eps

    def pdf(self, x: float) -> float:
        if not self.insupport(x):
            return 0.0
        if not self.epa:
            return 1.0
        t = x / self.eps
        return 1.0 - t * t

    def logpdf(self, x: float) -> float:
        if not self.insupport(x):
            return -math.inf
        if not self.epa:
            return 0.0
        t = x / self.eps
        v = 1.0 - t * t
        return math.log(v) if v > 0.0 else -math.inf

    def __repr__(self) -> str:
        return f"{type(self).__name__}({self.eps})"


class Indicator0toϵ(_ABCKernel):  # src/abcdez_types.jl:26-36
    kind = K_INDICATOR


class IndicatorStrict0toϵ(_ABCKernel):  # src/abcdez_types.jl:38-48
    kind = K_INDICATOR_STRICT
    strict = True


class Epa0toϵ(_ABCKernel):  # src/abcdez_types.jl:51-61
    kind = K_EPA
    epa = True


class EpaStrict0toϵ(_ABCKernel):  # src/abcdez_types.jl:63-73
    kind = K_EPA_STRICT
    strict = True
    epa = True


# ASCII aliases
Indicator0toeps = Indicator0toϵ
IndicatorStrict0toeps = IndicatorStrict0toϵ
Epa0toeps = Epa0toϵ
EpaStrict0toeps = EpaStrict0toϵ

ALL_KERNELS = (Indicator0toϵ, IndicatorStrict0toϵ, Epa0toϵ, EpaStrict0toϵ)


def kernel_kind(ABCk) -> int:
    if isinstance(ABCk, type) and issubclass(ABCk, _ABCKernel) and ABCk.kind >= 0:
        return ABCk.kind
    raise TypeError("ABCk must be one of Indicator0toϵ, IndicatorStrict0toϵ, Epa0toϵ, EpaStrict0toϵ")
