#!/usr/bin/env python3
"""The differential campaign of tools/fuzz_parity.py for the SHARDED population: every rank of a gloo group (all of them on the one
GPU of the box, the library's collectives over its host transport, abcdez_comm_init_host) runs the random case sharded and then, in
the same process, unsharded; its replica of the sharded run must equal its own unsharded run bit for bit -- whole abcdesmc run with
all histories, and a few abcdemc generations.  (Unsharded HIP against the CPU oracle is fuzz_parity.py's half.)

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29533 \\
        tools/fuzz_sharded.py --cases 200 > gpurun_out/fuzz_sharded_w3.jsonl
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch                                             # noqa: E402
import torch.distributed as dist                         # noqa: E402

import abcdez_amd as A                                   # noqa: E402
from abcdez_amd.engine import HipEngine                  # noqa: E402
from fuzz_parity import random_case, same, user_form_of                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=0.0)
    ap.add_argument("--big", action="store_true", help="populations of 50,000 to 600,000 particles (chunks of many workgroups), at most 8 generations")
    ap.add_argument("--user", action="store_true", help="the Normal simulator as user-supplied source (run-time compiled sweep / replay / abcdemc kernels)")
    args = ap.parse_args()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    pg = dist.group.WORLD
    t0 = time.time()
    n_bad = ran = 0
    for seed in range(args.first, args.first + args.cases):
        stop = torch.tensor([1 if args.seconds and time.time() - t0 > args.seconds else 0])
        dist.broadcast(stop, src=0)                      # every rank leaves the loop at the same case
        if int(stop):
            break
        c = random_case(seed, big=args.big)
        if args.big:
            c["mc"]["generations"] = min(c["mc"]["generations"], 4)
        if args.user:
            if c["sim"] == "mvn" and c["simulator"].blobs:
                c["simulator"] = A.MVNormal(tuple(c["simulator"].data()), sigma=c["simulator"].params()[0])
            usim = user_form_of(c["simulator"], c["d"], np.random.default_rng(seed))
            if usim is None:
                continue
            c["simulator"] = usim
        prior, sim, kern = c["prior"], c["simulator"], c["ABCk"]
        N = -(-c["N"] // 12) * 12                        # divisible by worlds of 1, 2, 3 and 4 ranks
        kw = dict(nparticles=N, verbose=False, rng=seed + 1, ABCk=kern, max_iters=8 if args.big else 30, **c["smc"])
        # the target: a quantile of the initial distances, taken from the unsharded engine (the same on every rank)
        probe = HipEngine(A.ModelSpec(prior, sim, kern, seed=seed + 1), N)
        probe.init_population()
        eps = float(np.quantile(probe.result()["C"], c["q"])) if c["q"] > 0 else 0.0
        del probe
        t = time.time()
        bad = []
        try:
            s = A.abcdesmc(prior, sim, eps, None, engine=HipEngine, process_group=pg, **kw)
            u = A.abcdesmc(prior, sim, eps, None, engine=HipEngine, **kw)
            assert s.engine.sharded_packed == (world > 1) and not u.engine.sharded_packed
            if world > 1:
                assert s.engine.ops.comm_kind() == 2
            if not (s.iters == u.iters and s.nsims == u.nsims and (s.logZ == u.logZ or (math.isnan(s.logZ) and math.isnan(u.logZ)))):
                bad.append(("iters/nsims/logZ", (s.iters, s.nsims, s.logZ), (u.iters, u.nsims, u.logZ)))
            for k in ("ϵs", "esss", "faccs", "γ0s", "Kmcmcs", "logZs", "P", "Wns", "C"):
                if not same(getattr(s, k), getattr(u, k)):
                    bad.append((k,))
            if not same(np.array(s.ranges_ϵ), np.array(u.ranges_ϵ)):
                bad.append(("ranges_ϵ",))
            if sim.blobs and not same(s.blobs, u.blobs):
                bad.append(("blobs",))
            mkw = dict(nparticles=N, verbose=False, rng=seed + 2, **c["mc"])
            ms = A.abcdemc(prior, sim, eps, None, engine=HipEngine, process_group=pg, **mkw)
            mu = A.abcdemc(prior, sim, eps, None, engine=HipEngine, **mkw)
            if not (ms.nsims == mu.nsims and same(ms.P, mu.P) and same(ms.C, mu.C) and ms.reached_ϵ == mu.reached_ϵ):
                bad.append(("abcdemc", ms.nsims, mu.nsims))
            if sim.blobs and not same(ms.blobs, mu.blobs):
                bad.append(("abcdemc blobs",))
            info = dict(eps=eps, iters=s.iters, nsims=s.nsims, logZ=s.logZ)
        except Exception as e:
            info, bad = dict(error=f"{type(e).__name__}: {e}"[:300]), [("exception",)]
        flag = torch.tensor([1 if bad else 0])
        dist.all_reduce(flag)                            # a difference on ANY replica ends the campaign on every rank
        ran += 1
        if rank == 0 or bad:
            print(json.dumps(dict(seed=seed, rank=rank, world=world, d=c["d"], families=c["nfam"], sim=c["sim"], kernel=kern.__name__, N=N,
                                  q=c["q"], blobs=bool(sim.blobs), **c["smc"], mc=c["mc"], **info, seconds=round(time.time() - t, 2),
                                  same=not bad, **({"differences": [str(b)[:200] for b in bad]} if bad else {})), ensure_ascii=False), flush=True)
        if int(flag):
            n_bad = int(flag)
            break
    if rank == 0:
        print(json.dumps(dict(summary=True, world=world, first=args.first, ran=ran, replicas_that_differ=n_bad, seconds=round(time.time() - t0, 1))), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if n_bad else 0)


if __name__ == "__main__":
    main()
