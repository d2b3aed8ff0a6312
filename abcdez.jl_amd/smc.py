"""Host driver ``abcdesmc`` -- ABC-DE SMC with evidence estimate.

A literal restatement of the reference's host loop ``abcdesmc!``
(src/abcdez_smc.jl:215-394): this side owns argument validation, the ε-schedule,
the log-evidence accumulator, the ESS test, the ``Kmcmc`` early exit, γ0 tuning,
history vectors and the stop criteria.  Every population-sized operation
(the ★ rows of SURVEY.md section 8a) is one call into the engine, i.e. one C-ABI
call into ``libabcdez_hip.so``.
"""
from __future__ import annotations

import logging
import math
import unicodedata
import warnings
from types import SimpleNamespace

import numpy as np

from .kernels import IndicatorStrict0toϵ
from .model import ModelSpec
from .priors import prior_length

log = logging.getLogger("abcdez_amd")


class Result(SimpleNamespace):
    """Result record with the reference's field names.  Python NFKC-normalises identifiers
    (the reference's ``ϵ`` U+03F5 becomes ``ε`` U+03B5 in source code); string lookups such as
    ``getattr(r, "ϵ")`` are normalised the same way so both spellings work."""

    def __getattr__(self, name):
        norm = unicodedata.normalize("NFKC", name)
        if norm != name:
            return getattr(self, norm)
        raise AttributeError(name)


def save_checkpoint(path, ckpt: dict) -> None:
    """Write a checkpoint dict (``result.checkpoint()``) to ``path`` (numpy ``.npz``; scalars and histories as JSON)."""
    import json

    arrays = {k: v for k, v in ckpt["state"].items() if isinstance(v, np.ndarray)}
    meta = {"kind": ckpt["kind"], "host": ckpt["host"], "history": ckpt.get("history"),
            "state": {k: v for k, v in ckpt["state"].items() if not isinstance(v, np.ndarray)}}
    with open(path, "wb") as f:      # np.savez on a file object: no ".npz" appended to the name
        np.savez(f, __meta__=np.frombuffer(json.dumps(meta).encode("utf-8"), dtype=np.uint8), **arrays)


def load_checkpoint(path) -> dict:
    import json

    with np.load(path) as z:
        meta = json.loads(bytes(z["__meta__"]).decode("utf-8"))
        state = {k: z[k] for k in z.files if k != "__meta__"}
    state.update(meta["state"])
    return {"kind": meta["kind"], "state": state, "host": meta["host"], "history": meta.get("history")}


def get_ess(Wns) -> float:
    """Effective sample size ``1/sum(Wns.^2)`` (src/abcdez_smc.jl:8), host version."""
    w = np.asarray(Wns, dtype=np.float64)
    return 1.0 / float(np.sum(w * w))


def quantile_type7(xj: float, xj1: float, g: float) -> float:
    """Interpolation step of Julia's default ``quantile`` (type 7), used at smc:301."""
    return xj + g * (xj1 - xj)


def wsample_stratified(weights, rng: int = 1, draw: int = 0, engine=None):
    """Stratified resampling indices for normalised ``weights`` (src/abcdez_smc.jl:15-56).

    The reference's tests call ``ABCdeZ.wsample_stratified!(rng, weights, inds)`` directly to
    turn a weighted population (continuous-weight kernels) into posterior samples
    (test/runtests.jl:13-19).  Runs on the device through ``abcdez_wsample_stratified``;
    returns 0-based indices (the reference's are 1-based).  ``r.P[wsample_stratified(r.Wns)]``
    is the Python spelling of ``r.P[weightinds(r.Wns)]``.
    """
    import torch

    from .priors import Normal
    from .simulators import DiracSquare

    w = np.ascontiguousarray(weights, dtype=np.float64)
    if not abs(float(w.sum()) - 1.0) < 1e-8:
        raise ValueError("Sum of weights expected to be 1.0 (approximately)")   # test/runtests.jl:14
    if engine is not None:
        ops = engine.ops
    else:
        from .engine import HipOps

        ops = HipOps(ModelSpec(Normal(0.0, 1.0), DiracSquare(1.5), seed=rng))   # only the seed matters here
    wt = torch.from_numpy(w).to(ops.device)
    inds = torch.zeros(w.size, dtype=torch.int32, device=ops.device)
    ops.wsample_stratified(wt, draw, inds)
    return inds.cpu().numpy().astype(np.int64)


def _check(cond: bool, msg: str) -> None:
    if not cond:
        raise ValueError(msg)  # the reference raises ErrorException via error(...)


def _make_engine(spec: ModelSpec, nparticles: int, engine, process_group, storage: str = "packed"):
    if engine is not None:
        if not callable(engine):
            return engine
        import inspect

        if "storage" in inspect.signature(engine).parameters:
            return engine(spec, nparticles, process_group, storage=storage)
        return engine(spec, nparticles, process_group)
    from .engine import HipEngine  # fails loudly if the HIP library or the GPU is missing

    return HipEngine(spec, nparticles, process_group, storage=storage)


def abcdesmc(prior, dist, ϵ_target, varexternal=None, *,
             nparticles: int = 100, α: float = 0.95, δess: float = 0.5,
             nsims_max: int = 10 ** 7, Kmcmc: int = 3, Kmcmc_min: float = 1.0,
             ABCk=IndicatorStrict0toϵ, facc_stop: float = 0.0, facc_min: float = 0.0, facc_tune: float = 0.975,
             verbose: bool = True, verboseout: bool = True, rng: int = 1, parallel: bool = False,
             engine=None, process_group=None, max_iters: int = 1_000_000, resume=None):
    """Run ABC with differential-evolution moves in an SMC setup (src/abcdez_smc.jl:215).

    Same positional arguments, keywords and defaults as the reference, except:
    ``dist`` is a :class:`~abcdez_amd.simulators.DeviceSimulator`; ``rng`` is the
    64-bit Philox seed; ``parallel`` is ignored (the population always runs on the
    GPU); ``varexternal`` is accepted and ignored (device simulators carry their own
    constants).  Returns a namespace with the reference's fields
    ``P, Wns, C, ϵ, logZ, blobs`` and, with ``verboseout``, ``ϵs, ranges_ϵ, logZs,
    esss, faccs, γ0s, Kmcmcs`` (src/abcdez_smc.jl:388-393).

    Checkpoint / resume: ``max_iters`` stops the loop after that many generations (in total); the result's
    ``checkpoint()`` returns the population and the host loop's variables, and ``resume=<that dict>`` (or the
    path of a file written by :func:`save_checkpoint`) continues the run.  All randomness is counter-based,
    so a resumed run reproduces the uninterrupted one bit for bit.
    """
    # ---- initialisation / validation: src/abcdez_smc.jl:223-235
    _check(0.0 <= α < 1.0, "α must be in 0 <= α < 1")
    _check(0.0 <= δess <= 1.0, "δess must be in 0 <= δess <= 1")
    _check(0.0 <= facc_stop <= 1.0, "facc_stop must be in 0 <= facc_stop <= 1")
    _check(0.0 <= facc_min <= 1.0, "facc_min must be in 0 <= facc_min <= 1")
    _check(0.0 <= facc_tune <= 1.0, "facc_tune must be in 0 <= facc_tune <= 1")
    _check(0.0 <= ϵ_target, "ϵ_target must be non-negative")
    _check(1 <= Kmcmc, "Kmcmc must be at least 1")
    _check(0.0 <= Kmcmc_min <= math.inf, "Kmcmc_min must be in 0 <= Kmcmc_min <= Inf")
    _check(1 <= nsims_max, "nsims_max must be at least 1")
    if not Kmcmc_min > facc_min:
        warnings.warn("Kmcmc_min should be larger than facc_min")
    d = prior_length(prior)
    # guard against α = δess = 0 (the reference would divide by zero here)
    denom = min(α, δess)
    nparticles_min = math.ceil(3 * d / denom) if denom > 0 else 3
    _check(nparticles_min <= nparticles, f"nparticles must be at least {nparticles_min}")

    spec = ModelSpec(prior, dist, ABCk, seed=rng)
    eng = _make_engine(spec, nparticles, engine, process_group)
    if verbose:                        # smc:238-239 (the executor of :237 has no counterpart: the engine named here runs the population)
        log.info("Running abcdesmc! with executor %s (%d rank(s))", type(getattr(eng, "ops", eng)).__name__, getattr(eng, "world", 1))
        log.info("Running abcdesmc! with ϵ_target=%s nparticles=%d α=%s δess=%s nsims_max=%d Kmcmc=%d Kmcmc_min=%s ABCk=%s "
                 "facc_stop=%s facc_min=%s facc_tune=%s rng=%s parallel=%s verboseout=%s", ϵ_target, nparticles, α, δess, nsims_max,
                 Kmcmc, Kmcmc_min, ABCk.__name__, facc_stop, facc_min, facc_tune, spec.seed, parallel, verboseout)

    ess_min = nparticles * δess        # smc:259
    γσ = 1e-5                          # smc:281
    if resume is None:
        # prior draws, log-prior, first distances, redraw until finite: smc:242-252 (S1)
        eng.init_population()

        ϵ = math.inf                       # smc:255
        ϵ_k = math.inf                     # smc:256 (the kernel object is (ABCk, ϵ_k))
        ABCk(ϵ_k)                          # ctor check, types.jl:30
        logZ = 0.0                         # smc:263
        eng.reset_weights()                # Wns = 1/N, alive = true: smc:266-270
        ess = 0.0
        nsims = 0                          # sum(nsims), smc:274
        facc = 1.0
        Ki = Kmcmc
        γ0 = 2.38 / math.sqrt(2 * d)       # smc:280
        updates = 0                        # Σ over sweeps of n_alive (the bench metric's numerator)
        iters = 0

        if verboseout:                     # smc:284-292
            ϵs = [ϵ]
            ranges_ϵ = [eng.extrema()]
            logZs = [logZ]
            esss = [eng.get_ess()]
            faccs = [facc]
            γ0s = [γ0]
            Kmcmcs = [Ki]
    else:
        ck = load_checkpoint(resume) if isinstance(resume, (str, bytes)) or hasattr(resume, "__fspath__") else resume
        _check(ck.get("kind") == "abcdesmc", "resume: not an abcdesmc checkpoint")
        eng.upload_state(ck["state"])
        h = ck["host"]
        ϵ, ϵ_k, logZ, ess, facc, γ0 = (float(h[k]) for k in ("eps", "eps_k", "logZ", "ess", "facc", "gamma0"))
        nsims, Ki, updates, iters = (int(h[k]) for k in ("nsims", "Ki", "updates", "iters"))
        if verboseout:
            hist = ck.get("history") or {}
            ϵs, logZs, esss, faccs, γ0s = (list(map(float, hist.get(k, []))) for k in ("eps", "logZ", "ess", "facc", "gamma0"))
            ranges_ϵ = [tuple(map(float, r)) for r in hist.get("ranges", [])]
            Kmcmcs = [int(v) for v in hist.get("Kmcmc", [])]

    while iters < max_iters:           # smc:295
        iters += 1
        if facc < facc_min:            # smc:320 (depends on the generation before only: tuned ahead of the engine's one call)
            γ0 *= facc_tune
        # smc:301-353 in ONE engine call -- on the unsharded HIP population one library call (abcdez_smc_generation_packed):
        #   new ϵ = max(min(quantile(Δs[alive], α), ϵ), ϵ_target) (smc:301; S9: the order statistics come from the device, the schedule
        #   stays here: ϵ and ϵ_target go in, the new ϵ comes back); target weights, normalisation, alive mask smc:305-311 (S5), ESS
        #   smc:323 (S6), extrema(Δs) of the generation that just ended (smc:364)                        [one host synchronisation]
        #   resampling when ess < ess_min, smc:323-326 (S7, S8)
        #   for i in 1:Kmcmc: sweep; naccs, nsims; (naccs / n_alive >= Kmcmc_min) && (Ki = i; break), smc:336-353 (S2, S3), the test
        #   of :352 on the device between the sweeps                                                      [one host synchronisation]
        # Unless this is known to be the last generation the engine may start the next generation's quantile select behind the
        # sweeps (it only reads Δs; its arguments will be α and ϵ_target)
        g = eng.smc_generation(α, ϵ, ϵ_target, ϵ_k, ess_min, γ0, γσ, Kmcmc, Kmcmc_min, select_ahead=iters < max_iters)
        ϵ, wnorm, ess, n_alive, range_prev = g["eps"], g["wnorm"], g["ess"], g["n_alive"], g["range"]
        if verboseout and iters > 1 and len(ranges_ϵ) < len(ϵs):
            ranges_ϵ.append(range_prev)
        ABCk(ϵ)
        # evidence, smc:315
        logZ += math.log(wnorm) if wnorm > 0.0 else (-math.inf if wnorm == 0.0 else math.nan)
        naccs = sum(g["naccs"])        # smc:318, :345
        Ki = g["Ki"] if g["Ki"] else Kmcmc
        nsims += sum(g["nsims"])
        updates += n_alive * g["Ki"]
        facc = naccs / (n_alive * Ki) if n_alive > 0 else math.nan   # smc:357
        ϵ_k = ϵ                        # smc:360
        if verboseout:                 # smc:362-370; ranges_ϵ gets this generation's extrema from the next prologue
            ϵs.append(ϵ)
            logZs.append(logZ)
            esss.append(ess)
            faccs.append(facc)
            γ0s.append(γ0)
            Kmcmcs.append(Ki)
        if verbose:                    # smc:372 -- range_ϵ = extrema(Δs) of THIS generation: one more small reduction, made for the log line only
            log.info("Finished run: iteration=%d nsim=%d ϵ=%s range_ϵ=%s ess=%s facc=%s logZ=%s", iters, nsims, ϵ, eng.extrema(), ess,
                     facc, logZ)
        if n_alive < 3:                # smc:375
            warnings.warn("No alive particles")
            break
        if ϵ <= ϵ_target or nsims >= nsims_max or facc < facc_stop:   # smc:376
            break
    eng.discard_select_ahead()             # a run that stops on nsims_max / facc_stop leaves the next select armed
    if verboseout and len(ranges_ϵ) < len(ϵs):
        ranges_ϵ.append(eng.extrema())     # smc:364 for the last generation

    if verbose:                        # smc:379
        log.info("Final run: iteration=%d nsim=%d ϵ=%s range_ϵ=%s ess=%s facc=%s logZ=%s", iters, nsims, ϵ, eng.extrema(), ess, facc, logZ)

    res = eng.result()                 # P is push_p-cast, smc:382
    out = Result(P=res["P"], Wns=res["Wns"], C=res["C"], ϵ=ϵ, logZ=logZ, blobs=res.get("blobs"))
    out.eps = ϵ
    out.iters, out.nsims, out.updates = iters, nsims, updates
    out.engine = eng
    host = dict(eps=ϵ, eps_k=ϵ_k, logZ=logZ, ess=ess, facc=facc, gamma0=γ0, nsims=nsims, Ki=Ki, updates=updates, iters=iters)
    history = dict(eps=list(ϵs), logZ=list(logZs), ess=list(esss), facc=list(faccs), gamma0=list(γ0s),
                   ranges=[tuple(r) for r in ranges_ϵ], Kmcmc=list(Kmcmcs)) if verboseout else None
    out.checkpoint = lambda: {"kind": "abcdesmc", "state": eng.download_state(), "host": dict(host), "history": history}
    if verboseout:
        out.ϵs, out.ranges_ϵ, out.logZs, out.esss = ϵs, ranges_ϵ, logZs, esss
        out.faccs, out.γ0s, out.Kmcmcs = faccs, γ0s, Kmcmcs
    return out
