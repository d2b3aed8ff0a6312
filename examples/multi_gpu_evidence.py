#!/usr/bin/env python3
"""BASELINE.json configs[4] as a script: the two models of examples/minimal_example.jl at 2^23 particles,
sharded over the GPUs of one node (one process per GPU, RCCL over xGMI), Bayes factor against the exact
finite-eps value.  Also shows blobs (posterior-predictive draws) and checkpoint / resume.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/multi_gpu_evidence.py
    python examples/multi_gpu_evidence.py 1048576            # single GPU, smaller population

Every rank ends with the same full result (the population is replicated; ranks sweep their own shard and
exchange one byte per particle and sweep), bit-identical to a single-GPU run of the same seed.
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
import torch.distributed as dist
from scipy import stats

from abcdez_amd import Normal, Normal1D, abcdesmc

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
pg = None
if world > 1:
    dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    pg = dist.group.WORLD

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 23
N -= N % world
data, eps = 3.0, 0.3
logZ = {}
for name, s2 in (("model 1", 10.0), ("model 2", 100.0)):
    sim = Normal1D(data, blobs=True)                      # r.blobs = the simulated x behind every particle
    part = abcdesmc(Normal(0.0, math.sqrt(s2)), sim, eps, None, nparticles=N, verbose=False, nsims_max=10 ** 12,
                    process_group=pg, max_iters=20)     # stop after 20 generations ...
    r = abcdesmc(Normal(0.0, math.sqrt(s2)), sim, eps, None, nparticles=N, verbose=False, nsims_max=10 ** 12,
                 process_group=pg, resume=part.checkpoint())   # ... and continue: same result as an uninterrupted run
    logZ[name] = r.logZ
    exact = stats.norm.cdf((data + eps) / math.sqrt(s2 + 1)) - stats.norm.cdf((data - eps) / math.sqrt(s2 + 1))
    if rank == 0:
        alive = r.Wns > 0
        print(f"{name}: logZ {r.logZ:.5f} (exact {math.log(exact):.5f}), posterior mean {r.P[alive].mean():.4f} "
              f"(exact {s2 / (s2 + 1) * data:.4f}), posterior-predictive mean {r.blobs[alive].mean():.4f}, "
              f"{r.iters} generations, {r.updates:.3e} particle-updates on {world} GPU(s)")
        assert np.array_equal(np.abs(r.blobs - data), r.C)
if rank == 0:
    print(f"Bayes factor {math.exp(logZ['model 1'] - logZ['model 2']):.4f} (exact 2.1043)")
if pg is not None:
    dist.destroy_process_group()
