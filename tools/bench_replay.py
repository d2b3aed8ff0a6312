#!/usr/bin/env python3
"""Cost of the sharded packed population's per-sweep and per-generation steps, measured on ONE GPU.

Emulates rank 0 of a G-rank job at 2^22 particles per rank (the bench.py workload, d = 32 MVN): a full population of
G * 2^22 particles lives on the GPU (as it would on every rank).  The accept flags of one sweep are produced by sweeping
everything on a scratch copy; the timed parts are what rank 0 does
  per sweep:       abcdez_smc_swarm_packed over its own chunk of the prefix + abcdez_smc_replay_packed over the others'
  per generation:  abcdez_smc_prologue_packed over the whole replicated population (extrema, quantile, reweight, partition)
Prints one JSON line per G (device times from HIP events on the library's stream), with the generation time and scaling
efficiency they add up to (3 sweeps per generation; xGMI: 7 links x 50 GB/s effective per GPU, 20 us per collective).

    python tools/bench_replay.py [--gpus-emulated 2 4 8] [--config smc32|lv|evidence1d] [--total-particles N]

--total-particles N: STRONG scaling (the population is N whatever G: BASELINE.json's 8-GPU configurations -- Lotka-Volterra
2^20, two-model evidence 2^23); default: WEAK scaling at --particles-per-gpu (bench.py --gpus G).
"""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import abcdez_amd as A
from abcdez_amd.engine import PACKED_ALIGN, HipEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus-emulated", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--particles-per-gpu", type=int, default=1 << 22)
    ap.add_argument("--dim", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--config", default="smc32", choices=["smc32", "lv", "evidence1d"])
    ap.add_argument("--total-particles", type=int, default=0, help="strong scaling: the population whatever G")
    args = ap.parse_args()
    d = args.dim
    if args.config == "smc32":
        prior = A.Factored(*[A.Normal(0.0, 1.0) for _ in range(d)])
        sim = A.MVNormal(tuple([1.0] * d))
    elif args.config == "lv":
        import json as _json
        g = _json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lv_data.json")))
        d = 4
        prior = A.Factored(*[A.Uniform(0.0, 2.0)] * 4)
        sim = A.LotkaVolterraRK4(tuple(g["obs"]), x0=g["x0"], y0=g["y0"], dt=g["dt"], steps_per_obs=g["steps_per_obs"], noise=g["noise"])
    else:
        d = 1
        prior, sim = A.Normal(0.0, math.sqrt(10.0)), A.Normal1D(3.0)
    gamma0 = 2.38 / math.sqrt(2 * d)
    base = None
    for G in args.gpus_emulated:
        N = args.total_particles if args.total_particles else args.particles_per_gpu * G
        e = HipEngine(A.ModelSpec(prior, sim, seed=1), N)
        ld = e.ops.layout()[0]
        e.init_population()
        e.reset_weights()
        eps, eps_k = math.inf, math.inf
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_pro = []
        for g in range(4):                                   # a few generations so acceptance is at its typical level
            ev[0].record()
            eps, wnorm, ess, n, _ = e.smc_prologue(0.95, eps, 0.0, eps_k, 0.5 * N)
            ev[1].record()
            torch.cuda.synchronize()
            t_pro.append(ev[0].elapsed_time(ev[1]))
            e.alive_compact()
            if g < 3:
                for _ in range(3):
                    e.smc_swarm(eps, gamma0, 1e-5)
            eps_k = eps
        b_in, b_out = e.bits[e.bc], e.bits[1 - e.bc]
        s0, s1 = e.buf[0][0], e.buf[1][0]
        lp, dl = e.buf[e.cur][1], e.buf[e.cur][2]
        flags = torch.zeros(N + G * PACKED_ALIGN, dtype=torch.uint8, device="cuda")
        # accept flags of the whole sweep, on scratch copies of the mutable state
        c0, c1, clp, cdl, cout = s0.clone(), s1.clone(), lp.clone(), dl.clone(), b_out.clone()
        nacc, _ = e.ops.smc_swarm_packed(b_in, cout, n, 0, n, c0, c1, clp, cdl, flags, eps, gamma0, 1e-5, e.sweep)
        del c0, c1, clp, cdl
        chunk = -(-(-(-n // G)) // PACKED_ALIGN) * PACKED_ALIGN
        r_hi = min(chunk, n)
        t_sw, t_rp = [], []
        for _ in range(args.reps):                           # the same sweep again and again: writes go to the other slots
            lp1, dl1, fl1, lp2 = lp.clone(), dl.clone(), flags.clone(), lp.clone()   # scratch state, outside the timed region
            ev[0].record()
            e.ops.smc_swarm_packed(b_in, b_out, n, 0, r_hi, s0, s1, lp1, dl1, fl1, eps, gamma0, 1e-5, e.sweep, want_counts=False)
            ev[1].record()
            e.ops.smc_replay_packed(b_in, b_out, n, 0, r_hi, s0, s1, lp2, flags, gamma0, 1e-5, e.sweep)
            ev[2].record()
            torch.cuda.synchronize()
            t_sw.append(ev[0].elapsed_time(ev[1]))
            t_rp.append(ev[1].elapsed_time(ev[2]))
        assert torch.equal(b_out, cout)                      # shard + replay == the full sweep
        acc_all = int((flags[:n] & 1).sum().item())
        acc_remote = acc_all - int((flags[:r_hi] & 1).sum().item())
        rp = min(t_rp) if G > 1 else 0.0
        # what a generation of the job adds up to on every rank: 3 x (own sweep + flag all-gather + replay) + the replicated
        # prologue + the part of the distance all-gather the replay of the last sweep does not cover
        link = 7 * 50e9
        t_flags = ((n - r_hi) / link * 1e3 + 0.020) if G > 1 else 0.0
        t_dist = max(0.0, (8 * (n - r_hi) / link * 1e3 + 0.020) - rp) if G > 1 else 0.0
        gen_ms = 3 * (min(t_sw) + t_flags + rp) + min(t_pro[1:]) + t_dist
        if base is None:
            base = (G, gen_ms, n)
        eff = (base[1] / gen_ms) * (n / base[2]) * (base[0] / G)      # updates per second per GPU relative to the first G measured
        print(json.dumps({
            "config": args.config, "scaling": "strong" if args.total_particles else "weak",
            "generation_ms_model": gen_ms, "flag_allgather_ms_model": t_flags, "distance_allgather_exposed_ms_model": t_dist,
            "efficiency_vs_first_row": eff,
            "gpus_emulated": G, "particles_total": N, "n_alive": n, "acceptance": nacc / n, "own_positions": r_hi,
            "own_sweep_ms": min(t_sw), "replay_ms": rp, "replayed_positions": n - r_hi,
            "replayed_accepted": acc_remote,
            "replay_GBps": (acc_remote * 32 * ld + (n - r_hi) * 1.125) / (rp * 1e-3) / 1e9 if rp > 0 else 0.0,
            "prologue_ms_replicated": min(t_pro[1:]),
            "xgmi_bytes_per_sweep_flags": n - r_hi, "xgmi_bytes_per_generation_distances": 8 * (n - r_hi),
            "xgmi_bytes_per_sweep_rows_allgather_it_replaces": (n - r_hi) * (8 * ld + 16),
        }), flush=True)
        del e, s0, s1, lp, dl, b_in, b_out, flags, cout
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
