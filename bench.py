#!/usr/bin/env python3
"""bench.py -- particle-updates/s of the abcdesmc population loop on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
d = 32 MVN simulator (prior 32 x N(0,1), x = theta + z, y = 1-vector, Euclidean
distance), abcdesmc with alpha = 0.95, delta_ess = 0.5, Kmcmc = 3, IndicatorStrict,
2^22 particles per GPU (weak scaling).  One *step* = one SMC generation of the
reference's main loop (src/abcdez_smc.jl:295-377): eps-quantile, reweight, (resample),
alive compaction and up to Kmcmc DE-Metropolis sweeps.  One particle-update = one alive
particle through one sweep.  State is resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]

For N > 1 launch with torch.distributed.run (one rank per GPU, RCCL).  Rank 0 prints ONE
JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def measured_traffic_per_update():
    """HBM bytes per particle-update of the sweep kernel from the committed PMC passes
    (profiles/r01_hbm_traffic.json: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE,
    separate rocprofv3 --pmc runs of this same command).  None if the file is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            return float(json.load(f)["total_bytes_per_update"])
    except (OSError, KeyError, ValueError):
        return None


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--particles-per-gpu", type=int, default=1 << 22)
    p.add_argument("--dim", type=int, default=32)
    p.add_argument("--lanes", type=int, default=0, help="lanes per particle (0 = library default)")
    p.add_argument("--cpu-particles", type=int, default=1 << 22, help="population of the CPU-oracle baseline sample")
    p.add_argument("--cpu-steps", type=int, default=10)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--force-collectives", action="store_true",
                   help="diagnostic: run the sharded code path (RCCL flag all-gather + replay) in a group of one rank")
    return p.parse_args()


class Generation:
    """The reference's generation loop body (smc:295-377) on an engine, one call per step."""

    def __init__(self, eng, d, eps_target, alpha=0.95, delta_ess=0.5, Kmcmc=3, Kmcmc_min=1.0):
        self.e, self.d = eng, d
        self.eps_target, self.alpha, self.delta_ess = eps_target, alpha, delta_ess
        self.Kmcmc, self.Kmcmc_min = Kmcmc, Kmcmc_min
        self.eps = math.inf
        self.eps_k = math.inf
        self.logZ = 0.0
        self.gamma0 = 2.38 / math.sqrt(2 * d)
        self.updates = 0
        self.sweeps = 0
        self.nsims = 0
        self.resamples = 0

    def step(self):
        e = self.e
        self.eps = max(min(e.quantile_alive(self.alpha), self.eps), self.eps_target)      # smc:301
        wnorm, ess, n_alive = e.smc_reweight(self.eps_k, self.eps)                         # smc:305-311
        self.logZ += math.log(wnorm)                                                       # smc:315
        if ess < e.N * self.delta_ess:                                                     # smc:323-326
            e.smc_resample()
            n_alive = e.N
            self.resamples += 1
        e.alive_compact()
        naccs = 0
        for i in range(1, self.Kmcmc + 1):                                                 # smc:336-353
            nacc, nsim = e.smc_swarm(self.eps, self.gamma0, 1e-5, last=(i == self.Kmcmc))
            naccs += nacc
            self.nsims += nsim
            self.updates += n_alive
            self.sweeps += 1
            if naccs / n_alive >= self.Kmcmc_min:
                break
        self.eps_k = self.eps


def cpu_baseline_faithful(spec, d, eps_target, cores, n=1 << 16, sweeps=3):
    """The LITERAL restatement of abcdesmc_swarm! (oracle ref_smc_swarm: donors by rejection around the O(N)
    scan of wsample(rng, 1:N, alive), src/abcdez_smc.jl:119-126) at a size where its O(N^2) sweep still
    finishes in seconds -- documents the wall the reference hits long before 4 M particles."""
    import ctypes as C

    import numpy as np
    from oracle import oracle as O

    eng = O.oracle_engine(spec, n)
    eng.init_population()
    eng.reset_weights()
    eps = eng.quantile_alive(0.95)
    eng.smc_reweight(math.inf, eps)
    th, lp, dl = (t.numpy() for t in eng.buf[eng.cur])
    alive = eng.alive.numpy()
    nth, nlp, ndl = np.zeros_like(th), np.zeros_like(lp), np.zeros_like(dl)
    m = O.OracleModel(spec)
    nacc, nsim = C.c_int64(), C.c_int64()
    t0 = time.perf_counter()
    for k in range(sweeps):
        O.lib().ref_smc_swarm(m.ptr, alive.ctypes.data, n, th.ctypes.data, lp.ctypes.data, dl.ctypes.data, nth.ctypes.data,
                              nlp.ctypes.data, ndl.ctypes.data, eps, 2.38 / math.sqrt(2 * d), 1e-5, k, C.byref(nacc),
                              C.byref(nsim))
    dt = time.perf_counter() - t0
    return {"value": sweeps * int(alive.sum()) / dt, "unit": "particle-updates/s", "cores": cores, "kind": "port",
            "sample": f"{sweeps} literal sweeps (O(N) donor scans, as the reference) at {n} particles, {dt:.2f} s; "
                      f"cost per update grows linearly with the population"}


def cpu_baseline(args, prior, sim, eps_target):
    """The oracle (a C port of the reference algorithm, OpenMP) on a bounded sample of the workload."""
    import abcdez_amd as A
    from oracle import oracle as O

    spec = A.ModelSpec(prior, sim, seed=1)
    eng = O.oracle_engine(spec, args.cpu_particles)
    eng.init_population()
    eng.reset_weights()
    g = Generation(eng, args.dim, eps_target)
    g.step()                                  # untimed warm-up generation
    u0 = g.updates
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        g.step()
    dt = time.perf_counter() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    faithful = cpu_baseline_faithful(spec, args.dim, eps_target, cores)
    return faithful, {
        "value": (g.updates - u0) / dt,
        "unit": "particle-updates/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{args.cpu_steps} generations of the same d={args.dim} MVN abcdesmc workload at "
                  f"{args.cpu_particles} particles (oracle/abcdez_oracle.c, OpenMP, {dt:.1f} s)",
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import abcdez_amd as A
    from abcdez_amd.engine import HipEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    pg = None
    if world > 1 or args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
        pg = dist.group.WORLD

    d = args.dim
    prior = A.Factored(*[A.Normal(0.0, 1.0) for _ in range(d)])
    sim = A.MVNormal(tuple([1.0] * d))
    eps_target = 6.0 * math.sqrt(d / 32.0)
    N = args.particles_per_gpu * world
    spec = A.ModelSpec(prior, sim, seed=1)
    eng = HipEngine(spec, N, pg, lanes=args.lanes, force_collectives=args.force_collectives)
    ld, L, C = eng.ops.layout()
    eng.init_population()
    eng.reset_weights()
    gen = Generation(eng, d, eps_target)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        gen.step()
    eng.ops.set_timing(True)
    u0, s0 = gen.updates, gen.sweeps
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gen.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    updates = gen.updates - u0          # global particle-updates (all ranks)
    kern_ms, launches, units = eng.ops.get_timing()   # this rank's sweep kernel
    sweeps_timed, resamples_timed = gen.sweeps - s0, gen.resamples
    phases = {}
    if eng.sharded_rows:
        # device-event breakdown of the sharded sweep, taken on three EXTRA generations after the timed region
        # (recording eight events per sweep would cost the timed loop several percent)
        eng.enable_phase_timing()
        for _ in range(3):
            gen.step()
        phases = eng.phase_timing()

    if rank == 0:
        # algorithmic bytes per particle-update: reads 24 ld + 17, writes 8 ld + 16
        b_read, b_write = 24 * ld + 17, 8 * ld + 16
        avg_ms = kern_ms / max(launches, 1)
        units_per_launch = units / max(launches, 1)
        ach = (b_read + b_write) * units_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        tpu = measured_traffic_per_update() if (d == 32 and L == 4) else None
        out = {
            "metric": "particle-updates/sec per SMC generation",
            "value": updates / dt,
            "unit": "particle-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"abcdesmc d={d} MVN simulator + Euclidean distance, {args.particles_per_gpu} particles/GPU "
                            f"(BASELINE.json configs[2]); alpha=0.95 delta_ess=0.5 Kmcmc=3 IndicatorStrict",
                "particles_total": N, "d": d, "lanes_per_particle": L, "comps_per_lane": C,
                "sweeps": sweeps_timed, "resamples": resamples_timed, "eps": gen.eps, "logZ": gen.logZ,
                "parallelism": (f"particle-shard x{world}, replicated row store: per-sweep accept-flag all-gather + replay, "
                                "per-generation distance all-gather") if world > 1 else "single GPU",
            },
            "roofline": {
                "kernel": "smc_swarm_kernel", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": tpu * units_per_launch if tpu else None,          # bytes per launch (PMC), cf. algorithmic below
                "algorithmic_bytes_per_launch": (b_read + b_write) * units_per_launch,
                "bytes_per_update": b_read + b_write, "updates_per_launch": units_per_launch,
                "avg_launch_ms": avg_ms, "launches": launches,
                "read_only_achieved": b_read * units_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                "read_only_frac": b_read * units_per_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if avg_ms > 0 else 0.0,
                "kernel_updates_per_s": units_per_launch / (avg_ms * 1e-3) if avg_ms > 0 else 0.0,
            },
        }
        if eng.sharded_rows:    # rank 0's device-event breakdown of the sharded sweep
            out["sharded_phases_ms"] = {k: {"calls": c, "avg_ms": (t / c if c else 0.0)} for k, (c, t) in phases.items()}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline_reference_faithful"], out["cpu_baseline"] = cpu_baseline(args, prior, sim, eps_target)
        try:                      # RCCL's start-up banner sits in libc's stdio buffer: push it out BEFORE the result line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if pg is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
