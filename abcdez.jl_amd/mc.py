"""Host driver ``abcdemc`` -- ABC-DE MCMC (posterior samples only).

Literal restatement of the reference's host loop ``abcdemc!``
(src/abcdez_mc.jl:102-172); the per-generation sweep ``abcdemc_swarm!``
(src/abcdez_mc.jl:5-61) and the reductions steering it run on the device.
"""
from __future__ import annotations

import contextlib
import logging
import math

from .model import ModelSpec
from .priors import prior_length
from .smc import Result, _check, _make_engine, load_checkpoint

log = logging.getLogger("abcdez_amd")


def abcdemc(prior, dist, ϵ_target, varexternal=None, *,
            nparticles: int = 50, generations: int = 20, verbose: bool = True, rng: int = 1,
            parallel: bool = False, engine=None, process_group=None, resume=None):
    """Run ABC with differential-evolution moves in an MCMC setup (src/abcdez_mc.jl:102).

    Same arguments and defaults as the reference (see :func:`abcdesmc` for the
    differences in ``dist``/``rng``/``parallel``).  Returns ``(P, C, reached_ϵ, blobs)``
    (src/abcdez_mc.jl:171) as a namespace.  ``resume=result.checkpoint()`` (or a file written by
    ``save_checkpoint``) continues an earlier run up to ``generations`` in total, bit for bit.
    """
    α = 0.0                                                   # mc:107
    _check(0.0 <= ϵ_target, "ϵ_target must be non-negative")  # mc:108
    _check(5 <= nparticles, "nparticles must be at least 5")  # mc:109
    _check(1 <= generations, "generations must be at least 1")  # mc:110

    spec = ModelSpec(prior, dist, seed=rng)
    eng = _make_engine(spec, nparticles, engine, process_group, storage="classic")
    # the whole run on a stream of its own when the caller sits on the legacy default stream (engine.run_scope), which
    # serialises with every other blocking stream of the process (and could not be captured when graph replay -- optional,
    # off by default: measured slower -- is switched on)
    with (eng.run_scope() if hasattr(eng, "run_scope") else contextlib.nullcontext()):
        return _abcdemc_run(eng, spec, prior, ϵ_target, nparticles, generations, verbose, resume, α, parallel)


def _abcdemc_run(eng, spec, prior, ϵ_target, nparticles, generations, verbose, resume, α, parallel=False):
    if verbose:                                               # mc:113-114
        log.info("Running abcdemc! with executor %s (%d rank(s))", type(getattr(eng, "ops", eng)).__name__, getattr(eng, "world", 1))
        log.info("Running abcdemc! with ϵ_target=%s nparticles=%d generations=%d α=%s rng=%s parallel=%s", ϵ_target, nparticles,
                 generations, α, spec.seed, parallel)

    if resume is None:
        eng.init_population()                                 # mc:117-125 (S1)
        nsims = 0                                             # mc:128
        iters = 0
    else:
        ck = load_checkpoint(resume) if isinstance(resume, (str, bytes)) or hasattr(resume, "__fspath__") else resume
        _check(ck.get("kind") == "abcdemc", "resume: not an abcdemc checkpoint")
        eng.upload_state(ck["state"])
        nsims, iters = int(ck["host"]["nsims"]), int(ck["host"]["iters"])
    γ0 = 2.38 / math.sqrt(2 * prior_length(prior))            # mc:129
    γσ = 1e-5                                                 # mc:130
    complete = 1 - eng.count_gt(ϵ_target) / nparticles        # mc:133
    ϵ_l, ϵ_h = eng.extrema()                                  # mc:146, first generation; afterwards the sweep reports them
    # abcdemc!'s loop has no data-dependent exit, so the host runs up to `ahead` generations ahead of the device: a
    # generation is ISSUED (eps_pop of mc:147 is evaluated on the device from the extrema the sweep before left there) and its
    # reductions (mc:156, mc:146) are COLLECTED later, in order -- the loop below is the reference's, with the log lines lagging
    ahead = 4
    converged = False                                         # a collected generation had ϵ_h <= ϵ_target (stays true: mc:19)

    def collect():
        nonlocal nsims, complete, ϵ_l, ϵ_h, converged
        nsim, n_above, ϵ_l, ϵ_h, _ = eng.mc_generation_collect()
        nsims += nsim
        ncomplete = 1 - n_above / nparticles                  # mc:156
        if verbose and (ncomplete != complete or complete >= (nparticles - 1) / nparticles):
            log.info("Finished run: completion=%s nsim=%d range_ϵ=%s", ncomplete, nsims, (ϵ_l, ϵ_h))
        complete = ncomplete
        converged = converged or ϵ_h <= ϵ_target

    first = True
    while iters < generations:                                # mc:134
        iters += 1
        # ϵ_pop = max(ϵ_target, ϵ_l + α (ϵ_h - ϵ_l)) mc:147; the enumeration behind mc:23 (only while some Δ_i > ϵ_target) +
        # abcdemc_swarm! mc:149 (S2, S4); the reductions of mc:156 and of the next generation's mc:146 ride along
        eng.mc_generation_issue(α, ϵ_target, γ0, γσ, lo_hi=(ϵ_l, ϵ_h) if first else None, do_rank=not converged)
        first = False
        while eng.mc_generations_in_flight() > ahead:
            collect()
    while eng.mc_generations_in_flight() > 0:
        collect()

    conv = ϵ_h <= ϵ_target                                    # mc:163
    if verbose:
        log.info("End: completion=%s converged=%s nsim=%d range_ϵ=%s", complete, conv, nsims, (ϵ_l, ϵ_h))
    res = eng.result()                                        # mc:166
    out = Result(P=res["P"], C=res["C"], reached_ϵ=conv, blobs=res.get("blobs"))
    out.reached_eps = conv
    out.nsims, out.updates, out.complete = nsims, generations * nparticles, complete
    out.engine = eng
    out.checkpoint = lambda: {"kind": "abcdemc", "state": eng.download_state(),
                              "host": {"nsims": nsims, "iters": iters}, "history": None}
    return out
