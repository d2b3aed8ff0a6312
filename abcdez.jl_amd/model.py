"""ctypes mirror of ``abz_model`` (include/abcdez_spec.h) and its builder.

The model descriptor is what the reference passes around as
``(prior, dist!, varexternal, rng)`` plus the ``ABCk`` keyword
(src/abcdez_smc.jl:215-220, src/abcdez_mc.jl:102-104).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .kernels import IndicatorStrict0toϵ, kernel_kind
from .priors import Product, PRIOR_NEGBIN, PRIOR_NORMAL, PRIOR_PAD, Prior, prior_factors
from .simulators import DeviceSimulator

MAX_D = 256


class PriorDim(C.Structure):
    _fields_ = [
        ("family", C.c_int32),
        ("discrete", C.c_int32),
        ("p0", C.c_double),
        ("p1", C.c_double),
        ("c0", C.c_double),
        ("c1", C.c_double),
        ("reserved", C.c_double),
    ]


class Model(C.Structure):
    _fields_ = [
        ("d", C.c_int32),
        ("ld", C.c_int32),
        ("sim_id", C.c_int32),
        ("abck", C.c_int32),
        ("seed", C.c_uint64),
        ("n_data", C.c_int32),
        ("n_blob", C.c_int32),
        ("sim_p", C.c_double * 8),
        ("data", C.c_void_p),
        ("prior", PriorDim * MAX_D),
        ("mv", C.c_void_p),
        ("ext", C.c_void_p),
        ("n_ext", C.c_int32),
        ("reserved0", C.c_int32),
    ]


def next_pow2(n: int) -> int:
    p = 1
    while p < n:
        p <<= 1
    return p


def philox_key(rng) -> int:
    """`rng` of the reference signatures (src/abcdez_smc.jl:220, src/abcdez_mc.jl:104) -> the 64-bit Philox key: an int is the key
    itself; a numpy Generator / RandomState (the host-language counterpart of the reference's AbstractRNG) gives it with one
    draw, so seeding that generator makes the run reproducible the way seeding Julia's rng does for the CPU methods"""
    if hasattr(rng, "integers"):                 # numpy.random.Generator
        return int(rng.integers(0, 1 << 64, dtype="uint64"))
    if hasattr(rng, "randint") and hasattr(rng, "bytes"):      # numpy.random.RandomState
        return int.from_bytes(rng.bytes(8), "little")
    return int(rng) & 0xFFFFFFFFFFFFFFFF


class ModelSpec:
    """Host-side description; ``.cstruct(data_ptr)`` yields the C struct."""

    def __init__(self, prior: Prior, sim: DeviceSimulator, ABCk=IndicatorStrict0toϵ, seed: int = 1):
        if not isinstance(sim, DeviceSimulator):
            raise TypeError(
                "dist! must be a DeviceSimulator (built-in on-device simulator); arbitrary host closures "
                "cannot run on the GPU -- see INTEGRATION.md"
            )
        factors = prior_factors(prior)
        d = len(factors)
        if d > MAX_D:
            raise ValueError(f"length(prior) = {d} exceeds the supported maximum {MAX_D}")
        if sim.ndim is not None and sim.ndim != d:
            raise ValueError(f"{type(sim).__name__} needs length(prior) == {sim.ndim}, got {d}")
        self.prior = prior
        self.sim = sim
        self.d = d
        self.ld = next_pow2(d)
        self.abck = kernel_kind(ABCk)
        self.ABCk = ABCk
        self.seed = philox_key(seed)
        self.data = np.ascontiguousarray(np.asarray(sim.data(), dtype=np.float64))
        self.n_blob = int(sim.blob_size(d)) if getattr(sim, "blobs", False) else 0     # doubles per blob, 0 = off
        if self.n_blob > 64:
            raise ValueError(f"blobs of {self.n_blob} doubles exceed the supported maximum 64")
        self.discrete = tuple(bool(f.discrete) for f in factors)
        # (family, discrete, p0, p1, c0, c1, reserved): the first three families compute c1 here.  The wrapper families
        # (truncated(...), MixtureModel) keep their records in the model's ext table and point at them by offset.
        ext, self._desc = [], []
        for f in factors:
            if hasattr(f, "ext_record"):
                self._desc.append(tuple(f.descriptor_at(len(ext))))
                ext += f.ext_record()
            else:
                self._desc.append(self._full_descriptor(f))
        self.ext = np.ascontiguousarray(np.asarray(ext, dtype=np.float64)) if ext else None
        if isinstance(prior, Product):
            # push_p broadcasts the whole product over the vector (types.jl:21): one rule for every component
            self.discrete = tuple(bool(prior.discrete) for _ in factors)
            self._desc = [(q[0], int(prior.discrete)) + tuple(q[2:]) for q in self._desc]
        # a correlated Normal prior (priors.MvNormal): [mu | L^-1 | L], host memory -- the library copies it to the device
        self.mv = prior.mv_maps(self.ld) if hasattr(prior, "mv_maps") else None

    @staticmethod
    def _full_descriptor(f):
        q = tuple(f.descriptor())
        if len(q) == 7:
            return q
        fam, disc, p0, p1, c0 = q
        c1 = 1.0 / p1 if fam == PRIOR_NORMAL else (f.c1() if fam == PRIOR_NEGBIN else 0.0)
        return (fam, disc, p0, p1, c0, c1, 0.0)

    def cstruct(self, data_ptr: Optional[int]) -> Model:
        m = Model()
        m.d, m.ld, m.sim_id, m.abck = self.d, self.ld, self.sim.sim_id, self.abck
        m.seed = self.seed
        m.n_data = int(self.data.size)
        m.n_blob = self.n_blob
        params = tuple(self.sim.params())
        for i in range(8):
            m.sim_p[i] = params[i] if i < len(params) else 0.0
        m.data = data_ptr or None
        for k in range(MAX_D):
            fam, disc, p0, p1, c0, c1, reserved = self._desc[k] if k < self.d else (PRIOR_PAD, 0, 0.0, 0.0, 0.0, 0.0, 0.0)
            m.prior[k].family, m.prior[k].discrete = fam, disc
            m.prior[k].p0, m.prior[k].p1, m.prior[k].c0, m.prior[k].c1, m.prior[k].reserved = p0, p1, c0, c1, reserved
        m.mv = self.mv.ctypes.data if self.mv is not None else None
        m.ext = self.ext.ctypes.data if self.ext is not None else None
        m.n_ext = int(self.ext.size) if self.ext is not None else 0
        m.reserved0 = 0
        return m
