#!/usr/bin/env python3
"""Cost of the sharded row store's replay step, measured on ONE GPU.

Emulates rank 0 of a G-rank job at 2^22 particles per rank (the bench.py workload, d = 32 MVN):
a full population of G * 2^22 particles lives on the GPU (as it would on every rank), the accept
flags of one sweep are produced by sweeping everything on a scratch copy, then the timed part is what
rank 0 does per sweep: abcdez_smc_swarm_rows_shard over its own alive ranks + abcdez_smc_replay_rows over
the others'.  Prints one JSON line per G (times from HIP events on the library's stream).

    python tools/bench_replay.py [--gpus-emulated 2 4 8]
"""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import abcdez_amd as A
from abcdez_amd.engine import HipEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus-emulated", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--particles-per-gpu", type=int, default=1 << 22)
    ap.add_argument("--dim", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    d = args.dim
    prior = A.Factored(*[A.Normal(0.0, 1.0) for _ in range(d)])
    sim = A.MVNormal(tuple([1.0] * d))
    gamma0 = 2.38 / math.sqrt(2 * d)
    for G in args.gpus_emulated:
        N = args.particles_per_gpu * G
        e = HipEngine(A.ModelSpec(prior, sim, seed=1), N)
        ld = e.ops.layout()[0]
        e.init_population()
        e.reset_weights()
        eps = math.inf
        for _ in range(3):                                   # a few generations so acceptance is at its typical level
            eps = min(e.quantile_alive(0.95), eps)
            e.smc_reweight(math.inf, eps)
            e.alive_compact()
            for _ in range(3):
                e.smc_swarm(eps, gamma0, 1e-5)
        eps = min(e.quantile_alive(0.95), eps)
        e.smc_reweight(math.inf, eps)
        n = e.alive_compact()
        a_in, a_out = e.alive_row[e.ar], e.alive_row[1 - e.ar]
        s0, s1 = e.buf[0][0], e.buf[1][0]
        lp, dl = e.buf[e.cur][1], e.buf[e.cur][2]
        flags = torch.zeros(N, dtype=torch.uint8, device="cuda")
        # accept flags of the whole sweep, on scratch copies of the mutable state
        c0, c1, clp, cdl, cout = s0.clone(), s1.clone(), lp.clone(), dl.clone(), torch.empty_like(a_out)
        nacc, _ = e.ops.smc_swarm_rows_shard(a_in, cout, n, 0, n, c0, c1, clp, cdl, flags, eps, gamma0, 1e-5, e.sweep)
        del c0, c1, clp, cdl
        r_hi = n // G
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_sw, t_rp = [], []
        for _ in range(args.reps):                           # the same sweep again and again: writes go to the other slots
            ev[0].record()
            e.ops.smc_swarm_rows_shard(a_in, a_out, n, 0, r_hi, s0, s1, lp.clone(), dl.clone(), flags.clone(), eps, gamma0,
                                       1e-5, e.sweep)
            ev[1].record()
            e.ops.smc_replay_rows(a_in, a_out, n, 0, r_hi, s0, s1, flags, gamma0, 1e-5, e.sweep)
            ev[2].record()
            torch.cuda.synchronize()
            t_sw.append(ev[0].elapsed_time(ev[1]))
            t_rp.append(ev[1].elapsed_time(ev[2]))
        assert torch.equal(a_out[:n], cout[:n])                    # shard + replay == the full sweep
        acc_remote = int((flags & 1).sum().item()) - int((a_out[:r_hi] != a_in[:r_hi]).sum().item())
        rp = min(t_rp)
        print(json.dumps({
            "gpus_emulated": G, "particles_total": N, "n_alive": n, "acceptance": nacc / n,
            "own_sweep_ms_incl_host": min(t_sw), "replay_ms": rp, "replayed_ranks": n - r_hi,
            "replayed_accepted": acc_remote,
            "replay_GBps": (acc_remote * 32 * ld + (n - r_hi) * 9) / (rp * 1e-3) / 1e9,
            "xgmi_bytes_per_sweep_flags": N - N // G, "xgmi_bytes_per_sweep_rows_allgather": (N - N // G) * (8 * ld + 16),
        }), flush=True)
        del e, s0, s1, lp, dl, a_in, a_out, flags, cout
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
