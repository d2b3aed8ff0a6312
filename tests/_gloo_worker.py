"""Worker for tests/test_distributed_gloo.py: one rank of a world_size-N gloo job.

Runs the product's host drivers + PopulationEngine (sharding, per-sweep all-gathers,
counter all-reduces) with the CPU oracle as the compute backend, and writes the result
of rank 0 to <outdir>/result_<name>.npz."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch.distributed as dist

import abcdez_amd as A
from oracle import oracle as O


def cases(hip: bool = False):
    """hip: the product engine runs the case (a user-supplied simulator is compiled from its source); otherwise the CPU oracle runs the
    built-in simulator that source restates -- the results must be the same bits"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from user_sources import USER_MVN_LANES

    y20 = tuple(1.0 + 0.02 * k for k in range(20))
    return {
        # population sizes divisible by 1, 2, 3, 4 and 8 ranks
        "normal1d": (A.Normal(0, math.sqrt(10)), A.Normal1D(3.0, blobs=True), 0.3, 2016),     # blobs: stamps travel too
        "mvn8": (A.Factored(*[A.Normal(0, 1)] * 8), A.MVNormal((1.0,) * 8, blobs=True), 2.5, 1032),
        "quad2d": (A.Factored(A.Normal(0, 5), A.Normal(0, 5)), A.Quad2D(0.5), 0.05, 600),
        # BASELINE.json configs[2] in miniature: the two-phase sweep on 256-byte rows (4 lanes x 8 components)
        "mvn32": (A.Factored(*[A.Normal(0, 1)] * 32), A.MVNormal((1.0,) * 32), 7.5, 1056),
        # BASELINE.json configs[3] in miniature: Lotka-Volterra RK4, 4 x Uniform(0, 2) prior, 8 observations
        "lv": (A.Factored(*[A.Uniform(0.0, 2.0)] * 4),
               A.LotkaVolterraRK4((1.0, 0.5, 1.46, 0.43, 1.77, 0.62, 1.52, 1.13, 0.95, 1.31, 0.66, 1.09, 0.61, 0.79, 0.75, 0.6),
                                  dt=0.05, steps_per_obs=10, blobs=True), 1.2, 528),
        # further prior families (include/abcdez_spec.h ABZ_PRIOR_EXPONENTIAL ...): the replicas rebuild the log-priors of the
        # accepted rows themselves
        "further5": (A.Factored(A.Gamma(2.5, 0.6), A.truncated(A.Normal(1.0, 2.0), 0.0, 4.0), A.LogNormal(0.0, 0.5), A.Poisson(2.0),
                                A.TDist(4.0)), A.MVNormal((1.0, 0.5, 0.8, 2.0, 0.3)), 2.0, 528),
        # the wrapper families (ABZ_PRIOR_TRUNCATED / ABZ_PRIOR_MIXTURE): the HIP engine compiles this model's sweep, REPLAY and abcdemc
        # kernels at run time (csrc/abz_jit.hip); the replicas rebuild log-sum-exp log-priors
        "wrapped4": (A.Factored(A.truncated(A.Gamma(2.0, 1.0), 0.3, 5.0), A.MixtureModel([A.Normal(-1.0, 0.5), A.Laplace(1.0, 1.5)], [0.4, 0.6]),
                                A.Normal(0.5, 1.0), A.truncated(A.Poisson(4.0), 1, 9)), A.MVNormal((1.0, 0.5, 0.8, 3.0)), 1.5, 528),
        # a USER-SUPPLIED simulator in the cooperative form (20 parameters on 4 lanes of 8 components, 12 padding components): sweep,
        # replay and abcdemc kernels compiled at run time from the user's source; the oracle runs the built-in simulator it restates
        "user_mvn20": (A.Factored(*([A.Normal(0, 1)] * 18 + [A.Gamma(2.0, 1.0), A.Uniform(-3, 4)])),
                       A.UserSimulator(USER_MVN_LANES, params=(1.0,), data=y20) if hip else A.MVNormal(y20), 6.2, 528),
    }


def main():
    outdir = sys.argv[1]
    # "oracle": the CPU oracle as compute backend, collectives by torch.distributed (gloo) on CPU tensors -- pins engine.py's host logic
    # "hip" / "hip_ar": the product engine on cuda:0, every rank sharing the one GPU of the test box; the collectives are issued by the
    #          LIBRARY (abcdez_smc_sweeps_sharded, abcdez_mc_generation_sharded_async, abcdez_comm_allgather) over its host transport
    #          (abcdez_comm_init_host) with gloo's all-gather underneath; "hip_ar" also hands it gloo's all-reduce
    # "rccl1": the product engine in a ONE-rank RCCL group with the sharded code path forced on, so that a single-GPU box runs the
    #          real RCCL collectives end to end (RCCL refuses two ranks on one device)
    mode = sys.argv[2] if len(sys.argv) > 2 else "oracle"
    names = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    if mode == "hip_ar":
        os.environ["ABZ_HOST_ALLREDUCE"] = "1"
        mode = "hip"
    dist.init_process_group("nccl" if mode == "rccl1" else "gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pg = dist.group.WORLD if (world > 1 or mode in ("rccl1", "hip1")) else None
    engine = O.oracle_engine
    if mode in ("rccl1", "hip1"):       # hip1: the host transport in a group of one rank
        import functools

        import torch
        from abcdez_amd.engine import HipEngine

        torch.cuda.set_device(0)
        engine = functools.partial(HipEngine, force_collectives=True)
    if mode == "hip":
        import torch
        from abcdez_amd.engine import HipEngine

        torch.cuda.set_device(0)          # every rank shares the one GPU of the test box
        engine = HipEngine
    want_kind = {"rccl1": 1, "hip": 2 if world > 1 else 0, "hip1": 2}.get(mode)
    for name, (prior, sim, eps, N) in cases(hip=mode in ("hip", "hip1", "rccl1")).items():
        if names is not None and name not in names:
            continue
        # default storage: packed population; sharded = accept-flag exchange + replay on the replicas
        r = A.abcdesmc(prior, sim, eps, None, nparticles=N, verbose=False, rng=21, engine=engine,
                       process_group=pg)
        assert r.engine.packed and r.engine.sharded_packed == (world > 1 or mode in ("rccl1", "hip1"))
        if want_kind is not None:       # the collectives of the HIP ops are the library's own, over the transport the mode names
            assert r.engine._native_comm == (want_kind != 0) and r.engine.ops.comm_kind() == want_kind, (mode, r.engine.ops.comm_kind())
        m = A.abcdemc(prior, sim, eps, None, nparticles=N, generations=25, verbose=False, rng=22,
                      engine=engine, process_group=pg)
        if want_kind is not None:
            assert m.engine.ops.comm_kind() == want_kind
        res, mres = r.engine.result(), m.engine.result()
        # every rank must hold the same full population after the all-gathers
        np.savez(os.path.join(outdir, f"result_{name}_rank{rank}.npz"), theta=res["theta"], C=res["C"], Wns=res["Wns"],
                 logZ=r.logZ, eps_hist=np.array(r.ϵs), nsims=r.nsims, iters=r.iters, mc_theta=mres["theta"],
                 mc_C=mres["C"], mc_nsims=m.nsims, world=world, logpi=res["logpi"],
                 **({"blobs": res["blobs"], "mc_blobs": mres["blobs"]} if res["blobs"] is not None else {}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
