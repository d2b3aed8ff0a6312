#!/usr/bin/env python3
"""User-supplied simulators (hiprtc) against the built-in ones they restate, same process, alternating, results compared bit for bit:

  mvn32  -- BASELINE.json configs[2] (d = 32 Normal simulator, N = 2^22): the built-in simulator against the COOPERATIVE user form
            (abz_user_dist_lanes, tests/user_sources.py): sweep-kernel time between HIP events and generations per second
  lv     -- BASELINE.json configs[3] (Lotka-Volterra RK4, N = 2^20), whole run to eps = 1: the built-in simulator (rounds with early
            exit), the STAGED user form (abz_user_round: the same early exit through the user's running bound) and the opaque user
            form (abz_user_dist: every simulated proposal runs all 1500 steps)

    python tools/user_sim_ab.py mvn32|lv [reps]      -> JSON lines on stdout
"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import abcdez_amd as A
from abcdez_amd.engine import HipEngine
from user_sources import USER_LV, USER_LV_ROUNDS, USER_MVN_LANES

which = sys.argv[1] if len(sys.argv) > 1 else "mvn32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
only = sys.argv[3] if len(sys.argv) > 3 else ""          # substring of the simulator names to run (profiling one variant)
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream(0))


def checksum(t):
    return int(t.contiguous().view(torch.int64).sum().item())


if which == "mvn32":
    d, N = 32, 1 << 22
    prior = A.Factored(*[A.Normal(0.0, 1.0)] * d)
    sims = {"built-in": A.MVNormal((1.0,) * d), "user (abz_user_dist_lanes)": A.UserSimulator(USER_MVN_LANES, params=(1.0,), data=(1.0,) * d)}
    g0 = 2.38 / math.sqrt(2 * d)
    ref = None
    for rep in range(reps):
        for name, sim in sims.items():
            if only not in name:
                continue
            eng = HipEngine(A.ModelSpec(prior, sim, seed=1), N)
            eng.init_population(); eng.reset_weights()
            eps = eps_k = math.inf

            def gen():
                global eps, eps_k
                eps, wnorm, ess, n_alive, _ = eng.smc_prologue(0.95, eps, 6.0, eps_k, 0.5 * N)
                if ess < 0.5 * N:
                    eng.smc_resample(); n_alive = N
                eng.alive_compact()
                na, ns, Ki = eng.smc_sweeps(eps, g0, 1e-5, 3, 1.0, next_prologue=(0.95, 6.0))
                eps_k = eps
                return n_alive * Ki
            for _ in range(5):
                gen()
            eng.ops.set_timing(3 + 256 * 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            upd = sum(gen() for _ in range(20))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ms, launches, units = eng.ops.get_timing()
            eng.ops.set_timing(False)
            fp = (checksum(eng.state[0]), checksum(eng.state[2]), eng.n_alive, eps)
            ref = ref or fp
            assert fp == ref, (name, fp, ref)          # the two simulators leave the same population, bit for bit
            print(json.dumps({"config": "smc32 (d = 32, N = 2^22), 20 generations after 5", "simulator": name, "rep": rep,
                              "updates_per_s": upd / dt, "ms_per_generation": dt / 20 * 1e3,
                              "sweep_ms_per_launch": ms / max(launches, 1), "sweep_launches": launches,
                              "sweep_updates_per_s": units / (ms * 1e-3) if ms > 0 else None,
                              "layout": eng.ops.layout()}), flush=True)
            del eng
            torch.cuda.empty_cache()
else:
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "lv_data.json")))
    prior = A.Factored(*[A.Uniform(0.0, 2.0)] * 4)
    params = (g["x0"], g["y0"], g["dt"], float(g["steps_per_obs"]), g["noise"])
    sims = {"built-in (rounds of two observations, early exit)": A.LotkaVolterraRK4(tuple(g["obs"]), x0=g["x0"], y0=g["y0"], dt=g["dt"],
                                                                                    steps_per_obs=g["steps_per_obs"], noise=g["noise"]),
            "user, staged (abz_user_round, 8 rounds, early exit)": A.UserSimulator(USER_LV_ROUNDS % {"rounds": 8}, params=params, data=tuple(g["obs"])),
            "user, staged (abz_user_round, 16 rounds, early exit)": A.UserSimulator(USER_LV_ROUNDS % {"rounds": 16}, params=params, data=tuple(g["obs"])),
            "user, opaque (abz_user_dist: every call to the end)": A.UserSimulator(USER_LV, params=params, data=tuple(g["obs"]))}
    ref = None
    for rep in range(reps):
        for name, sim in sims.items():
            if only not in name:
                continue
            torch.cuda.synchronize()
            t = time.perf_counter()
            eng = HipEngine(A.ModelSpec(prior, sim, seed=1), 1 << 20)      # context creation: the hiprtc compile of a user form
            torch.cuda.synchronize()
            t_create = time.perf_counter() - t
            t = time.perf_counter()
            r = A.abcdesmc(prior, sim, 1.0, None, nparticles=1 << 20, verbose=False, rng=1, nsims_max=10 ** 12, engine=eng)
            dt = time.perf_counter() - t
            fp = (r.logZ, r.iters, r.nsims, float(np.sum(r.P)))
            ref = ref or fp
            assert fp == ref, (name, fp, ref)
            print(json.dumps({"config": "lv (N = 2^20), whole run to eps = 1", "simulator": name, "rep": rep, "seconds": dt,
                              "seconds_context_creation": t_create, "generations": r.iters, "nsims": r.nsims, "updates": r.updates,
                              "logZ": r.logZ, "seconds_is": "abcde_init!, every generation, the result download (context creation -- the "
                              "hiprtc compile of a user form -- timed apart)"}), flush=True)
            del eng, r
            torch.cuda.empty_cache()
