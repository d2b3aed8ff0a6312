#!/usr/bin/env python3
"""bench.py -- particle-updates/s of the ABC-DE population loop on MI355X.

Default workload = BASELINE.json configs[2], the configuration the metric is quoted on:
d = 32 MVN simulator (prior 32 x N(0,1), x = theta + z, y = 1-vector, Euclidean distance),
abcdesmc with alpha = 0.95, delta_ess = 0.5, Kmcmc = 3, IndicatorStrict, 2^22 particles per GPU
(weak scaling).  One *step* = one SMC generation of the reference's main loop
(src/abcdez_smc.jl:295-377): extrema(Ds) of the history (smc:364), eps-quantile, reweight, ESS, partition of the packed
population (one library call), (resample), up to Kmcmc DE-Metropolis sweeps (one library call).  One particle-update = one alive
particle through one sweep.  State is resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config smc32|mc1d|lv|evidence1d]

The other single-GPU configurations of BASELINE.json print the same JSON shape:
    mc1d        configs[1]: 1-D Normal, abcdemc, 2^20 particles (step = one generation, mc:134-161)
    lv          configs[3]: Lotka-Volterra RK4 (dt 0.01, 1500 steps per update), abcdesmc, 2^20 particles
    evidence1d  configs[4]: the two models of examples/minimal_example.jl, abcdesmc, 2^23 particles

N > 1: `python bench.py --gpus N ...` starts its own ranks (a child `python -m torch.distributed.run`, one rank per GPU over
RCCL; the parent never imports torch or touches a GPU) and relays rank 0's ONE JSON line and the children's exit code;
launched under torch.distributed.run (WORLD_SIZE set) it is a rank.  `--scaling weak` (default for smc32 / mc1d): the
configuration's population PER GPU; `--scaling strong` (default for lv / evidence1d, whose BASELINE sizes are totals over
8 GPUs): the configuration's population in total, split over the ranks.
"""
import argparse
import json
import math
import os
import sys
import time

# The CPU-baseline leg of one configuration runs the OpenMP oracle on every host core right before the next configuration's timed
# window: idle OpenMP workers must sleep, not spin, or they compete with the thread that polls the GPU's read-backs.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
# RCCL between processes (the driver launches the ranks itself under torch.distributed.run): this driver stack only has dmabuf IPC
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured streaming, 5.5-5.8 random rows)
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X vector fp64 (FMA = 2 flop): half the 157.3 TFLOP/s fp32 vector rate of the guide
FP64_VALU_PEAK_ORIGIN = ("nominal: 256 CUs x 4 SIMDs x 16 fp64 FMA lanes x 2 flop x 2.4 GHz = 78.6 TFLOP/s, i.e. half the 157.3 TFLOP/s "
                         "fp32 VECTOR rate MI355X_MICROARCH.md lists (the guide gives no fp64 vector figure; its 78.6 'FP64 matrix' "
                         "entry is the same number); tools/valu_rate.hip measures 5.4 instead of 4 cycles per v_fma_f64 wave-instruction "
                         "under dense fp64, so the sustained rate of the part is ~0.74 of this peak (profiles/r02_valu_rate.jsonl)")
HBM_ACHIEVABLE_GBS = 6290.0  # MI355X_MICROARCH.md: measured float4 copy rate (79 % of spec); random 2,304-B rows gathered once: 5.7-5.8 TB/s
PROFILE_TAG = "r06"


def profile_json(name):
    """a committed measurement under profiles/ (PMC traffic, access-pattern ceiling); None if absent"""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--no-pattern", action="store_true",
                   help="do not run tools/layout_bench for roofline.pattern_ceiling (profiling runs: keeps its kernels out of the trace)")
    p.add_argument("--steps", type=int, default=None)
    p.add_argument("--warmup", type=int, default=None)
    p.add_argument("--config", default="smc32", choices=sorted(CONFIGS))
    p.add_argument("--particles-per-gpu", type=int, default=None)
    p.add_argument("--particles-total", type=int, default=None, help="strong scaling: the population split over the ranks")
    p.add_argument("--scaling", default=None, choices=["weak", "strong"],
                   help="weak: the configuration's population per GPU; strong: in total (BASELINE.json configs[3] = --config lv "
                        "--gpus 8, configs[4] = --config evidence1d --gpus 8).  Default: weak for smc32 / mc1d, strong for lv / evidence1d")
    p.add_argument("--dim", type=int, default=32, help="smc32 only")
    p.add_argument("--lanes", type=int, default=0, help="lanes per particle (0 = library default)")
    p.add_argument("--cpu-particles", type=int, default=None, help="population of the CPU-oracle baseline sample")
    p.add_argument("--cpu-steps", type=int, default=None)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-whole-run", action="store_true")
    p.add_argument("--no-other-configs", action="store_true",
                   help="default run only: skip the other single-GPU configurations (mc1d, lv, evidence1d) reported as other_configs")
    p.add_argument("--storage", default=None, choices=["packed"], help="abcdesmc storage (one choice left: the packed population)")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="nccl = RCCL, one rank per GPU (the contract); gloo = rehearsal of the N > 1 path on a box with fewer GPUs "
                        "than ranks: the ranks share the visible GPUs, collectives go through the host")
    p.add_argument("--default-stream", action="store_true", help="diagnostic: run on the legacy default stream")
    p.add_argument("--graphs", action="store_true", help="diagnostic: replay abcdemc generations as HIP graphs (off by default: measured slower)")
    p.add_argument("--timing-mode", type=int, default=3, choices=[2, 3],
                   help="abcdesmc kernel timing (abcdez_ctx_set_timing): 3 = one HIP-event pair around ALL the sweeps of every 2nd timed "
                        "generation (default); 2 = a pair of its own around ONE sweep of every 2nd generation (rounds 2-4)")
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
        self.naccs = 0
        self.resamples = 0
        self.generations = 0
        self.range = (0.0, math.inf)

    def step(self):
        e = self.e
        # smc:301 (eps), smc:305-311 (reweight), ESS, extrema(Ds) of the generation that just ended (smc:364); the resampling when
        # ESS < N delta_ess (smc:323-326); the Kmcmc sweeps (smc:336-353) -- one engine call, on one GPU one library call
        g = e.smc_generation(self.alpha, self.eps, self.eps_target, self.eps_k, e.N * self.delta_ess, self.gamma0, 1e-5, self.Kmcmc,
                             self.Kmcmc_min)
        self.eps, self.range = g["eps"], g["range"]
        self.logZ += math.log(g["wnorm"])                                                  # smc:315
        self.resamples += 1 if g["resampled"] else 0
        self.naccs += sum(g["naccs"])
        self.nsims += sum(g["nsims"])
        self.updates += g["n_alive"] * g["Ki"]
        self.sweeps += g["Ki"]
        self.eps_k = self.eps
        self.generations += 1

    def done(self):
        return self.eps <= self.eps_target                                                 # smc:376


class McGeneration:
    """abcdemc!'s generation loop body (mc:134-161): rank pass while unconverged, one sweep, driver reductions."""

    def __init__(self, eng, d, eps_target):
        self.e, self.eps_target = eng, eps_target
        self.gamma0 = 2.38 / math.sqrt(2 * d)
        self.complete = 1 - eng.count_gt(eps_target) / eng.N                               # mc:133, as the driver (abcdez_amd/mc.py)
        self.lo, self.hi = eng.extrema()                                                   # mc:146 (first generation)
        self.updates = self.sweeps = self.nsims = self.generations = self.ranked = 0
        self.first, self.converged = True, False

    AHEAD = 4      # abcdemc!'s loop has no data-dependent exit: the host may issue generations ahead of their results

    def step(self):
        e = self.e
        # mc:147 (alpha = 0: eps_pop = max(eps_target, min Ds), evaluated on the device), rank pass, sweep mc:149
        e.mc_generation_issue(0.0, self.eps_target, self.gamma0, 1e-5, lo_hi=(self.lo, self.hi) if self.first else None,
                              do_rank=not self.converged)
        self.first = False
        self.ranked += not self.converged                                                  # mc:23's candidate sets are built
        self.updates += e.N
        self.sweeps += 1
        self.generations += 1
        while e.mc_generations_in_flight() > self.AHEAD:
            self.collect()

    def collect(self):
        nsim, n_above, self.lo, self.hi, _ = self.e.mc_generation_collect()                # mc:156, mc:146
        self.nsims += nsim
        self.complete = 1 - n_above / self.e.N
        self.converged = self.converged or self.hi <= self.eps_target

    def flush(self):
        while self.e.mc_generations_in_flight() > 0:
            self.collect()


# ---------------------------------------------------------------------------------------------- workloads
def lv_fixture():
    with open(os.path.join(ROOT, "tests", "golden", "lv_data.json")) as f:
        return json.load(f)


def cfg_smc32(A, args):
    d = args.dim
    prior = A.Factored(*[A.Normal(0.0, 1.0) for _ in range(d)])
    sim = A.MVNormal(tuple([1.0] * d))
    return dict(kind="smc", prior=prior, sim=sim, d=d, eps_target=6.0 * math.sqrt(d / 32.0), ppg=1 << 22, steps=20, warmup=5,
                cpu_particles=1 << 22, cpu_steps=10, storage="packed",
                workload=f"abcdesmc d={d} MVN simulator + Euclidean distance (BASELINE.json configs[2]); "
                         "alpha=0.95 delta_ess=0.5 Kmcmc=3 IndicatorStrict",
                exact_logZ=EXACT["logZ_mvn32_eps6"] if d == 32 else None,
                exact_post_mean=EXACT["post_mean_mvn32_eps6"] if d == 32 else None)


def cfg_mc1d(A, args):
    return dict(kind="mc", prior=A.Normal(0.0, math.sqrt(10.0)), sim=A.Normal1D(3.0), d=1, eps_target=0.3, ppg=1 << 20,
                steps=100, warmup=0, cpu_particles=1 << 20, cpu_steps=8, storage="classic",
                workload="abcdemc 1-D Normal simulator, data 3, eps 0.3 (BASELINE.json configs[1]); steps = the run's "
                         "generations from the initial population on (rank pass while unconverged + one fused sweep)",
                exact_logZ=None, exact_post_mean=EXACT["post_mean_normal1d_sigma2_10"])


def cfg_lv(A, args):
    g = lv_fixture()
    sim = A.LotkaVolterraRK4(tuple(g["obs"]), x0=g["x0"], y0=g["y0"], dt=g["dt"], steps_per_obs=g["steps_per_obs"],
                             noise=g["noise"])
    return dict(kind="smc", prior=A.Factored(*[A.Uniform(0.0, 2.0)] * 4), sim=sim, d=4, eps_target=1.0, ppg=1 << 20, steps=12,
                warmup=3, cpu_particles=1 << 15, cpu_steps=6, storage="packed",
                workload="abcdesmc Lotka-Volterra RK4 on device, dt 0.01 x 1500 steps per particle-update, 16 noisy (x, y) "
                         "observations, Euclidean distance (BASELINE.json configs[3]); alpha=0.95 Kmcmc=3",
                exact_logZ=None, exact_post_mean=None, scaling="strong")


def cfg_evidence1d(A, args):
    return dict(kind="smc", prior=A.Normal(0.0, math.sqrt(10.0)), sim=A.Normal1D(3.0), d=1, eps_target=0.3, ppg=1 << 23,
                steps=12, warmup=3, cpu_particles=1 << 21, cpu_steps=6, storage="packed",
                workload="abcdesmc two-model evidence of examples/minimal_example.jl (BASELINE.json configs[4]); timed: "
                         "generations of model 1 (prior N(0, sqrt 10)); both models then run to eps 0.3",
                exact_logZ=EXACT["logZ_normal1d_sigma2_10"], exact_post_mean=EXACT["post_mean_normal1d_sigma2_10"], scaling="strong")


def _exact():
    """closed forms of the BASELINE configurations (tests/golden/reference_known_answers.json, made by
    tests/golden/make_golden.py with scipy): finite-eps evidences and posterior means"""
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        a = json.load(f)["analytic"]
    return {"logZ_mvn32_eps6": a["Z_mvn32_eps6"]["logZ"], "post_mean_mvn32_eps6": a["Z_mvn32_eps6"]["posterior_mean_per_component"],
            "logZ_normal1d_sigma2_10": a["Z_exact_finite_eps_sigma2_10"]["logZ"],
            "logZ_normal1d_sigma2_100": a["Z_exact_finite_eps_sigma2_100"]["logZ"],
            "post_mean_normal1d_sigma2_10": a["Z_exact_finite_eps_sigma2_10"]["posterior_mean"],
            "post_mean_normal1d_sigma2_100": a["Z_exact_finite_eps_sigma2_100"]["posterior_mean"]}


EXACT = _exact()
CONFIGS = {"smc32": cfg_smc32, "mc1d": cfg_mc1d, "lv": cfg_lv, "evidence1d": cfg_evidence1d}


def make_loop(cfg, eng):
    if cfg["kind"] == "mc":
        return McGeneration(eng, cfg["d"], cfg["eps_target"])
    return Generation(eng, cfg["d"], cfg["eps_target"])


# ---------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline_faithful(spec, d, cores, n=1 << 16, sweeps=3):
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


def cpu_baseline(A, args, cfg):
    """The oracle (a C port of the reference algorithm, OpenMP) on a bounded sample of the same workload."""
    from oracle import oracle as O

    n = args.cpu_particles or cfg["cpu_particles"]
    steps = args.cpu_steps or cfg["cpu_steps"]
    spec = A.ModelSpec(cfg["prior"], cfg["sim"], seed=1)
    eng = O.oracle_engine(spec, n, storage=cfg["storage"])
    eng.init_population()
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    if cfg["kind"] == "mc":
        g = make_loop(cfg, eng)
        t0 = time.perf_counter()
        for _ in range(steps):
            g.step()
        dt = time.perf_counter() - t0
        return None, {"value": g.updates / dt, "unit": "particle-updates/s", "cores": cores, "kind": "port",
                      "sample": f"the first {steps} generations of the same abcdemc run at {n} particles "
                                f"(oracle/abcdez_oracle.c: OpenMP sweep; qsort rank pass in the generations that draw by rank, {dt:.1f} s)"}
    eng.reset_weights()
    g = make_loop(cfg, eng)
    g.step()                                  # untimed warm-up generation
    u0 = g.updates
    t0 = time.perf_counter()
    for _ in range(steps):
        g.step()
    dt = time.perf_counter() - t0
    faithful = cpu_baseline_faithful(spec, cfg["d"], cores) if args.config == "smc32" else None
    return faithful, {
        "value": (g.updates - u0) / dt, "unit": "particle-updates/s", "cores": cores, "kind": "port",
        "sample": f"{steps} generations of the same workload at {n} particles (oracle/abcdez_oracle.c, OpenMP, {dt:.1f} s)",
    }


# ---------------------------------------------------------------------------------------------- roofline
def lv_flops_per_update(sim):
    """fp64 flops of one Lotka-Volterra particle-update (abz_device.h sim_dist<ABZ_SIM_LV>): per RK4 step 20 fma
    (2 flop) + 8 mul + 2 add = 50 flop in 30 VALU instructions; per observation 2 fma (noise) + 2 sub + 2 fma
    (squared error) = 10 flop, Box-Muller pair not counted."""
    nobs = len(sim.obs) // 2
    return (nobs - 1) * sim.steps_per_obs * 50 + nobs * 10


def roofline(cfg, kind, ld, kern_ms, launches, units, acc_rate, positions=1 << 22, sim_rate=1.0):
    avg_ms = kern_ms / max(launches, 1)
    upl = units / max(launches, 1)
    rate = upl / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
    if args_config == "lv":
        # flops of the simulator calls that are MADE: the in-support proposals (sim_rate = nsims / updates of the timed window).  An
        # out-of-support proposal (smc:135) costs no arithmetic -- in rounds 1-4 its lane idled through its wave-mates' simulations
        # and the figure charged every update a full simulation (issue slots, not flops); the two-phase body packs the calls densely
        fl = lv_flops_per_update(cfg["sim"])
        ach = fl * rate * sim_rate / 1e12
        return {"kernel": "smc_lv_phase1_kernel + smc_lv_phase2_kernel (one sweep = two launches, both inside the event pair)", "bound": "valu", "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS, "traffic": None, "peak_origin": FP64_VALU_PEAK_ORIGIN,
                "note": "fp64 vector-ALU bound (no matrix work on this path): 1500 RK4 steps per SIMULATED proposal (in support of "
                        "the prior, smc:135-137); achieved = flops per simulation x simulations per second (a simulation that leaves "
                        "early on its running distance is charged in full: an upper bound of the arithmetic done); the row traffic "
                        "(161 B per update + 128 B per handed-over proposal) is 0.1 % of the launch.  Late in a run every proposal is "
                        "in support and runs to the end: the second launch then sustains 57 TFLOP/s = 0.73 of this peak, the "
                        "rate v_fma_f64 sustains on the part (profiles/r05_lv_run_by_tenth.json)", "flops_per_update": fl, "simulated_fraction_of_updates": sim_rate,
                "rk4_steps_per_s": rate * sim_rate * (len(cfg["sim"].obs) // 2 - 1) * cfg["sim"].steps_per_obs,
                "updates_per_launch": upl, "avg_launch_ms": avg_ms, "launches": launches, "kernel_updates_per_s": rate}
    d = cfg["d"]
    if kind == "mc":
        # SURVEY.md 8d uses the same per-update figure for both drivers (d = 1: 41 B read, 24 B written); the abcdemc sweep
        # really reads one row more (the base particle of mc:23) and 12 B of the enumeration
        b_read, b_write, kernel = 24 * ld + 17, 8 * ld + 16, "mc_swarm_kernel"
        b_moved = 32 * ld + 16 + 12 + b_write
    else:
        b_read, b_write, kernel = 24 * ld + 17, 8 * ld + 16, "smc_swarm_packed_kernel"
        # what the packed population has to move per update: the reads, an accepted row + its log-prior / distance
        # (the slot bits are 1/8 B per update)
        b_moved = b_read + acc_rate * b_write
    to_gbs = lambda b: b * rate / 1e9
    ach = to_gbs(b_read)
    out = {
        "kernel": kernel, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
        "frac_is": "HBM-READ fraction of north_star / SURVEY.md 8d: (24 ld + 17) B read per update x updates per launch "
                   "/ average launch time / 8 TB/s",
        "bytes_read_per_update": b_read, "updates_per_launch": upl, "avg_launch_ms": avg_ms, "launches": launches,
        "kernel_updates_per_s": rate,
        "launches_are": (LAUNCHES_ARE[TIMING_MODE] if kind != "mc" else
                         "abcdemc: the sweep of every 10th generation between its own pair of HIP events on the library's stream"),
        "total_bytes_frac": to_gbs(b_read + b_write) / HBM_PEAK_GBS,
        "total_bytes_note": f"SURVEY.md 8d total-bytes variant: {b_read + b_write} B per update, charging every update a full "
                            "row write (the double-buffered reference layout)",
        "moved_bytes_per_update": b_moved, "moved_bytes_frac": to_gbs(b_moved) / HBM_PEAK_GBS,
        "acceptance_rate": acc_rate,
    }
    if ld <= 2:
        # Rows of one or two doubles: every random donor / base-particle read is an 8- or 16-byte access that leaves the L2 as ONE
        # 128-byte line request (calibrated on a known stream, profiles/r05_pmc_gather_calibration.json) -- the sweep is bound by the
        # rate at which the fabric fills lines (Infinity Cache + HBM behind it), not by the 41 algorithmic bytes of SURVEY.md 8d.
        # achieved = line-fill bytes per update x kernel updates/s, with the line-fill bytes MEASURED where a counter file of this
        # configuration is committed (TCC requests of the timed sweeps; the tables are partly L2-resident, so fewer than one fill
        # per gather reaches the fabric) and the model 128 B x gathers + streamed bytes otherwise.
        gathers = 3 if kind == "mc" else 2
        streamed = 8 * ld + 17 + (12 if kind == "mc" else 0)
        model = 128 * gathers + streamed
        trn = profile_json(f"{PROFILE_TAG}_hbm_traffic_{args_config}.json")
        fills = trn["total_bytes_per_update"] if trn and trn.get("ld") == ld else model
        out["hbm_read_frac_algorithmic_bytes"] = out["frac"]
        out["bound"] = "fabric-line-fills"
        out["achieved"] = fills * rate / 1e9
        out["frac"] = out["achieved"] / HBM_PEAK_GBS
        # what the part has been measured to deliver for this pattern (MI355X_MICROARCH.md, random rows gathered from a table that the
        # 256 MiB Infinity Cache holds): 8.6 TB/s for a 38 MB table, 7.4-7.9 TB/s for 151 MB
        table_mb = positions * 8 * ld / 1e6
        out["achievable_peak_gbs"] = 8600.0 if table_mb <= 40 else 7900.0
        out["frac_of_achievable"] = out["achieved"] / out["achievable_peak_gbs"]
        out["line_fill_bytes_per_update"] = fills
        out["line_fill_bytes_per_update_model"] = model
        out["line_fill_bytes_are"] = ("measured: fabric requests of the timed sweeps' dispatches (rocprofv3 --pmc, committed file below) per update"
                                      if fills is not model else f"model: {gathers} random gathers x 128 B + {streamed} B streamed")
        out["frac_is"] = (f"FABRIC LINE-FILL fraction: a random {8 * ld}-byte row read is one 128-byte line request (model: {gathers} gathers x 128 B + "
                          f"{streamed} B streamed = {model} B per update); achieved = line-fill bytes per update x kernel updates/s against the 8 TB/s "
                          "spec (the guide measures 7.4-8.6 TB/s for random rows of an Infinity-Cache-resident table, 6.0-6.3 TB/s from HBM); "
                          "hbm_read_frac_algorithmic_bytes is SURVEY.md 8d's (24 ld + 17)-byte figure, the wrong ceiling for this access pattern")
    kc = profile_json(f"{PROFILE_TAG}_{args_config}_kernel_avg_check.json")
    if kc and kc.get("trace_avg_ms_timed_sweeps") and kc.get("updates_per_launch_timed_sweeps"):
        r_tr = kc["updates_per_launch_timed_sweeps"] / (kc["trace_avg_ms_timed_sweeps"] * 1e-3)
        out["frac_trace"] = (out["line_fill_bytes_per_update"] if ld <= 2 else b_read) * r_tr / 1e9 / HBM_PEAK_GBS
        out["frac_trace_source"] = (f"NOT measured in this run: rocprofv3 --kernel-trace of this command on another box, average over EVERY sweep "
                                    f"that ran in the timed steps ({kc.get('timed_sweeps')} launches, {kc['trace_avg_ms_timed_sweeps']:.4f} ms, "
                                    f"{kc['updates_per_launch_timed_sweeps']:.0f} updates each; warm-up launches excluded), "
                                    f"profiles/{PROFILE_TAG}_{args_config}_kernel_avg_check.json")
    # flat scalars next to frac: the same achieved bytes against the rate the part has been MEASURED to stream at
    if ld > 2:
        out["achievable_peak_gbs"] = HBM_ACHIEVABLE_GBS
        out["frac_of_achievable"] = out["achieved"] / HBM_ACHIEVABLE_GBS
    tr = profile_json(f"{PROFILE_TAG}_hbm_traffic_{args_config}.json")
    if tr and tr.get("ld") == ld:
        out["traffic"] = tr["total_bytes_per_update"] * upl
        out["traffic_source"] = (f"NOT measured in this run: {tr['total_bytes_per_update']:.1f} B per update (FETCH_SIZE x2 + "
                                 f"WRITE_SIZE, separate rocprofv3 --pmc passes of this very command line, counted over the sweep "
                                 f"dispatches of its TIMED steps only -- {tr.get('dispatches')} dispatches, acceptance "
                                 f"{tr.get('acceptance_rate')} --, profiles/{PROFILE_TAG}_hbm_traffic_"
                                 f"{args_config}.json) x this run's updates per launch")
        out["traffic_over_moved_bytes"] = tr["total_bytes_per_update"] / b_moved
    else:
        out["traffic"] = None
    if args_config == "smc32" and ld == 32:
        what = ("the sweep's memory-access pattern with all arithmetic removed (tools/layout_bench.hip, variant P), same "
                "population layout: a reference point, not a strict bound -- its rate depends on how many waves are in flight")
        live = pattern_ceiling_live(int(round(upl)), int(round(100 * acc_rate)), positions) if PATTERN_LIVE else None
        pc = profile_json(f"{PROFILE_TAG}_pattern_ceiling.json")
        if live:
            # flat (a nested object does not survive every consumer of this line): the arithmetic-free access pattern of the sweep,
            # measured in this process on this GPU right after the timed window, and the kernel's rate over it
            out["pattern_ceiling_read_frac"] = live["particles_per_s"] * b_read / 1e9 / HBM_PEAK_GBS
            out["kernel_over_pattern"] = rate / live["particles_per_s"]
            if live.get("best_single_launches"):
                out["kernel_over_pattern_single_launches"] = rate / live["best_single_launches"]
            out["pattern_ceiling"] = {"updates_per_s": live["particles_per_s"], "read_frac": live["particles_per_s"] * b_read / 1e9 / HBM_PEAK_GBS,
                                      "kernel_over_ceiling": rate / live["particles_per_s"],
                                      "updates_per_s_single_launches": live.get("best_single_launches"),
                                      "kernel_over_ceiling_single_launches": (rate / live["best_single_launches"]) if live.get("best_single_launches") else None,
                                      "single_launches_are": "the same pattern kernel, 20 launches each between its own pair of events -- how the sweep "
                                                             "kernel's avg_launch_ms is taken (launch gaps in, no overlap between consecutive launches); "
                                                             "best of the same occupancy caps",
                                      "source": f"measured in this run on this GPU right after the timed window, in process: tools/layout_bench.hip (packed pattern) "
                                                f"{live['prefix']} {live['accepted_percent']} (prefix = mean alive count of the timed "
                                                f"sweeps, accepted = their acceptance rate; mean of 5 x 20 launches; best of four occupancy "
                                                f"caps: {live['waves_per_simd_cap'] or 8} waves per SIMD)", "what": what}
        elif pc:
            out["pattern_ceiling"] = {"updates_per_s": pc["updates_per_s"], "read_frac": pc["updates_per_s"] * b_read / 1e9 / HBM_PEAK_GBS,
                                      "source": "NOT measured in this run: " + pc.get("source", f"profiles/{PROFILE_TAG}_pattern_ceiling.json"),
                                      "what": what}
    return out


PATTERN_LIVE = True
TIMING_MODE = 3
LAUNCHES_ARE = {
    3: ("abcdesmc: ONE pair of HIP events on the library's stream around ALL the sweep launches of EVERY timed generation "
        "(abcdez_ctx_set_timing mode 3, stride 1: every sweep of the timed steps is inside a pair): avg_launch_ms = elapsed / sweeps that ran.  The one-block group_check launches "
        "between the sweeps (~5 us each) are INSIDE the pair, so this is an upper bound of the sweep kernel's duration; the sweeps "
        "run back to back as they do un-instrumented (frac_trace: the rocprofv3 average of every sweep of the timed steps)"),
    2: ("abcdesmc: ONE sweep of every timed generation between its own pair of HIP events on the library's stream -- the "
        "generation's 1st, 2nd, 3rd sweep in rotation (abcdez_ctx_set_timing mode 2, stride 1; a pair costs ~9 us of queue "
        "time).  A bracketed launch starts on a drained queue, so this average is a few per cent ABOVE the un-instrumented "
        "kernel: frac_trace"),
}


def pattern_ceiling_live(prefix, accepted_percent, positions):
    """runs the arithmetic-free access pattern on the same GPU, in this process (tools/liblayout_bench.so, ~2 s); None if the
    tool is not built"""
    import ctypes as C
    path = os.path.join(ROOT, "tools", "liblayout_bench.so")
    if not os.path.exists(path) or prefix < 64 or prefix > positions or positions % 32:
        return None
    try:
        lib = C.CDLL(path)
        fn = lib.layout_bench_packed
        fn.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        fn.restype = C.c_int
    except (OSError, AttributeError):
        return None
    best, best_single = None, 0.0
    for cap in (0, 5, 4, 3):        # the pattern itself runs fastest at 4 waves per SIMD (fewer streams in flight): take the best
        mean, mn = C.c_double(), C.c_double()
        if fn(positions, prefix, accepted_percent, cap, 5, 20, C.byref(mean), C.byref(mn)) != 0 or not mean.value > 0:
            continue
        v = {"prefix": prefix, "accepted_percent": accepted_percent, "waves_per_simd_cap": cap, "ms_mean": mean.value,
             "particles_per_s": prefix / (mean.value * 1e-3)}
        # the same pattern timed the way the sweep kernel is timed: one launch between two events (launch gaps included, no
        # overlap of one launch's tail with the next one's head)
        if fn(positions, prefix, accepted_percent, cap, 20, 1, C.byref(mean), C.byref(mn)) == 0 and mean.value > 0:
            best_single = max(best_single, prefix / (mean.value * 1e-3))
        if best is None or v["particles_per_s"] > best["particles_per_s"]:
            best = v
    if best is not None and best_single > 0:
        best["best_single_launches"] = best_single
    return best


args_config = "smc32"


def run_config(args):
    """one configuration: engine, warm-up, timed window, roofline, whole run, CPU baseline -> (result dict on rank 0 else None, pg)"""
    global args_config, PATTERN_LIVE, TIMING_MODE
    args_config = args.config
    TIMING_MODE = args.timing_mode
    PATTERN_LIVE = not args.no_pattern and args.gpus == 1
    import torch
    import torch.distributed as dist

    import abcdez_amd as A
    from abcdez_amd.engine import HipEngine

    cfg = CONFIGS[args.config](A, args)
    if args.storage and cfg["kind"] == "smc":
        cfg["storage"] = args.storage
    steps = args.steps if args.steps is not None else cfg["steps"]
    warmup = args.warmup if args.warmup is not None else cfg["warmup"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # main() starts the ranks itself when WORLD_SIZE is unset; reaching this line means a launcher with another world size
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or run "
                         f"`python bench.py --gpus {args.gpus}` without a launcher: it starts its own ranks)")
    scaling = args.scaling or cfg.get("scaling", "weak")
    if scaling == "strong" and not args.particles_per_gpu:
        total = args.particles_total or cfg["ppg"]          # the configuration's stated population, in total
        ppg = max(total // world, 64)
    else:
        ppg = args.particles_per_gpu or cfg["ppg"]
    device = local_rank if args.dist_backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    if not args.default_stream:
        # everything on a stream of its own: the legacy default stream serialises with every other blocking stream of the
        # process (and cannot be captured when --graphs asks for graph replay of the abcdemc generations)
        torch.cuda.set_stream(torch.cuda.Stream(device))
    pg = None
    if world > 1 or args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device), rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        pg = dist.group.WORLD

    d = cfg["d"]
    N = ppg * world
    spec = A.ModelSpec(cfg["prior"], cfg["sim"], seed=1)
    eng = HipEngine(spec, N, pg, lanes=args.lanes, storage=cfg["storage"], force_collectives=args.force_collectives)
    ld, L, C = eng.ops.layout()
    if args.graphs:
        eng.ops.set_graphs(True)
    eng.init_population()
    if cfg["kind"] == "smc":
        eng.reset_weights()
    gen = make_loop(cfg, eng)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        gen.step()
    if hasattr(gen, "flush"):
        gen.flush()
    # HIP events around ONE sweep of every 2nd SMC generation (the 1st, 2nd, 3rd of the generation's sweeps in turn: the first runs
    # on a population the partition has just moved and is a few per cent slower than its siblings) / the sweep of every 4th
    # abcdemc generation: an event pair costs ~9 us of queue time (tools/launch_floor.hip) -- 0.6 % of an SMC generation, 7 % of
    # an abcdemc generation
    tstride = 1 if cfg["kind"] == "smc" else 10
    eng.ops.set_timing((args.timing_mode if cfg["kind"] == "smc" and not eng.sharded_packed else 2) + 256 * tstride)
    u0, s0, a0, r0 = gen.updates, gen.sweeps, getattr(gen, "naccs", 0), getattr(gen, "resamples", 0)
    n0 = getattr(gen, "nsims", 0)
    barrier()
    t0 = time.perf_counter()
    trace = [] if os.environ.get("ABZ_BENCH_TRACE_STEPS") else None       # diagnostic: per-step host times to stderr
    for _ in range(steps):
        gen.step()
        if trace is not None:
            trace.append(time.perf_counter())
    if trace is not None:
        print(args.config, "step ms:", [round((b - a) * 1e3, 3) for a, b in zip([t0] + trace[:-1], trace)], file=sys.stderr)
    if hasattr(gen, "flush"):
        gen.flush()                     # results of the generations still in flight (abcdemc runs ahead of its read-backs)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    updates = gen.updates - u0          # global particle-updates (all ranks)
    kern_ms, launches, units = eng.ops.get_timing()   # this rank's sweep kernel
    eng.ops.set_timing(False)
    window = dict(sweeps=gen.sweeps - s0, updates=updates, naccs=getattr(gen, "naccs", 0) - a0,
                  sweep_launches=steps * (gen.Kmcmc if cfg["kind"] == "smc" else 1),     # dispatches of the sweep kernel a profiler sees in the window (a group enqueues Kmcmc; those behind a held test of smc:352 return at once)
                  resamples=getattr(gen, "resamples", 0) - r0,
                  eps=getattr(gen, "eps", None), logZ=getattr(gen, "logZ", None))
    acc_rate = (gen.naccs - a0) / max(updates, 1) if cfg["kind"] == "smc" else 0.0
    phases = {}
    if eng.sharded_packed:
        # device-event breakdown of the sharded sweep, taken on three EXTRA generations after the timed region
        # (recording eight events per sweep would cost the timed loop several percent)
        eng.enable_phase_timing()
        for _ in range(3):
            gen.step()
        phases = eng.phase_timing()
        eng._prof = None                # no event pairs in the whole run below

    # ---- the whole run next to the window: from a fresh population to eps_target (device-resident, no result download) -- the
    # second half of BASELINE.json's metric: log-Z and posterior-mean error against the closed forms (tests/golden)
    whole = None

    def post_err(e2, exact):
        pm = e2.posterior_mean()
        out = {"posterior_mean": float(pm.mean()), "posterior_mean_is": "mean over the alive particles (Wns-weighted), averaged over "
               f"the {len(pm)} exchangeable components" if len(pm) > 1 else "mean over the alive particles (Wns-weighted)"}
        if len(pm) > 1:
            out["posterior_mean_components_min_max"] = [float(pm.min()), float(pm.max())]
        if exact is not None:
            out["posterior_mean_exact"] = exact
            out["posterior_mean_err"] = float(pm.mean()) - exact
            if len(pm) > 1:
                out["posterior_mean_max_component_err"] = float(abs(pm - exact).max())
        return out

    if cfg["kind"] == "smc" and not args.no_whole_run:
        whole = {}
        models = [("model", cfg["prior"], cfg["exact_logZ"], cfg["exact_post_mean"])] if args.config != "evidence1d" else \
            [("model1_prior_N(0,sqrt10)", cfg["prior"], EXACT["logZ_normal1d_sigma2_10"], EXACT["post_mean_normal1d_sigma2_10"]),
             ("model2_prior_N(0,sqrt100)", A.Normal(0.0, math.sqrt(100.0)), EXACT["logZ_normal1d_sigma2_100"],
              EXACT["post_mean_normal1d_sigma2_100"])]
        for name, prior, lz_exact, pm_exact in models:
            e2 = eng
            if prior is not cfg["prior"]:
                e2 = HipEngine(A.ModelSpec(prior, cfg["sim"], seed=1), N, pg, lanes=args.lanes, storage=cfg["storage"])
            barrier()
            t1 = time.perf_counter()
            e2.init_population()
            e2.reset_weights()
            g2 = Generation(e2, d, cfg["eps_target"])
            while not g2.done() and g2.generations < 5000:
                g2.step()
            barrier()
            dtw = time.perf_counter() - t1
            whole[name] = {"generations": g2.generations, "sweeps": g2.sweeps, "updates": g2.updates, "seconds": dtw,
                           "value": g2.updates / dtw, "logZ": g2.logZ, "eps": g2.eps, "resamples": g2.resamples,
                           "includes": "abcde_init! + every generation down to eps_target; not the result download"}
            if lz_exact is not None:
                whole[name]["logZ_exact"] = lz_exact
                whole[name]["logZ_err"] = g2.logZ - lz_exact
            if args.config != "lv":
                whole[name].update(post_err(e2, pm_exact))
            else:
                pm = e2.posterior_mean()
                whole[name]["posterior_mean"] = [float(v) for v in pm]
                whole[name]["posterior_mean_is"] = ("(a, b, c, e) over the alive particles; no closed form -- the data were simulated "
                                                    "at theta* = (1.0, 0.4, 1.0, 0.3) (tests/golden/lv_data.json)")
        if args.config == "evidence1d":
            z1, z2 = whole[models[0][0]]["logZ"], whole[models[1][0]]["logZ"]
            bf_exact = math.exp(EXACT["logZ_normal1d_sigma2_10"] - EXACT["logZ_normal1d_sigma2_100"])
            whole["bayes_factor"] = math.exp(z1 - z2)
            whole["exact"] = {"logZ1": EXACT["logZ_normal1d_sigma2_10"], "logZ2": EXACT["logZ_normal1d_sigma2_100"], "bayes_factor": bf_exact}
            whole["bayes_factor_rel_err"] = whole["bayes_factor"] / bf_exact - 1.0
    elif cfg["kind"] == "mc" and not args.no_whole_run and warmup == 0:
        # abcdemc: the timed window IS the run (its generations from the initial population on); no evidence in abcdemc!
        whole = {"model": {"generations": gen.generations, "updates": gen.updates, "seconds": dt, "completion": gen.complete,
                           "max_distance": gen.hi, **post_err(eng, cfg["exact_post_mean"])}}

    if rank == 0:
        out = {
            "metric": "particle-updates/sec per SMC generation" if cfg["kind"] == "smc" else "particle-updates/sec per abcdemc generation",
            "value": updates / dt,
            "unit": "particle-updates/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{cfg['workload']}; {ppg} particles/GPU" + (f" ({N} in total, strong scaling)" if scaling == "strong" and world > 1 else ""),
                "name": args.config,
                "particles_total": N, "d": d, "lanes_per_particle": L, "comps_per_lane": C,
                "timed_window": window,
                "launched_incl_warmup": {"sweeps": gen.sweeps, "updates": gen.updates},     # what a profiler sees of the sweep kernel
                "step_includes": ("extrema(Ds), eps-quantile, reweight, ESS, partition (one call), resample when ESS < N/2, "
                                  "<= Kmcmc sweeps with their counter read-backs (smc:301-364)") if cfg["kind"] == "smc" else
                                 "rank pass (while max Ds > eps_target and fewer than 1 / 16 of the particles lie at or below it), one sweep with nsim / completion / extrema folded in (mc:140-161)",
                "parallelism": (f"particle-shard x{world}, replicated packed population: per-sweep accept-flag all-gather + replay, "
                                "per-generation distance all-gather") if world > 1 else "single GPU",
                # who moves the bytes between the ranks: always the library's own sharded entry points (abcdez_smc_sweeps_sharded,
                # abcdez_mc_generation_sharded_async) -- over RCCL / xGMI, or over its host transport when the ranks share GPUs
                "collectives": {0: "none", 1: "libabcdez_hip.so over RCCL (xGMI), on the library's stream",
                                2: "libabcdez_hip.so over its host transport (abcdez_comm_init_host; gloo all-gather underneath): a "
                                   "rehearsal on fewer GPUs than ranks, not a scaling measurement"}[eng.ops.comm_kind()],
            },
            "roofline": roofline(cfg, cfg["kind"], ld, kern_ms, launches, units, acc_rate, positions=N,
                                 sim_rate=((gen.nsims - n0) / max(updates, 1)) if cfg["kind"] == "smc" else 1.0),
        }
        out["roofline"]["timed_every_nth_step"] = tstride     # which steps carried the HIP-event pair (mode 3: around all their sweeps; mode 2: around one, in turn)
        out["roofline"]["timing_mode"] = args.timing_mode if cfg["kind"] == "smc" and not eng.sharded_packed else 2
        if cfg["kind"] == "mc":
            out["config"]["timed_window"].update(unconverged_generations=gen.ranked, completion=gen.complete, max_distance=gen.hi)
            if hasattr(eng.ops, "mc_rank_stats"):
                rs = eng.ops.mc_rank_stats()
                out["config"]["better_particle_draws"] = {
                    "generations_with_a_rank_pass": int(sum(rs)), "generations_by_rejection_without_one": int(eng.ops.mc_draw_stats()),
                    "rule": "by rejection once at least 1 / 16 of the particles lie at or below eps_target (include/abcdez_spec.h)"}
            gr = eng.ops.graph_stats()
            out["config"]["graph_replay"] = {"generations_replayed": gr[0], "graphs_captured": gr[1], "generations_stream_launched": gr[2],
                                             "what": "off by default (measured slower than stream launches); with --graphs one abcdemc generation "
                                                     "(rank pass + sweep + snapshot) is captured once per launch shape and replayed as a HIP "
                                                     "graph, generations whose sweep carries the bench's event pair are enqueued launch by launch"}
        if whole is not None:
            out["whole_run"] = whole
            # the second half of BASELINE.json's metric ("posterior-mean & log-Z error vs ref"), from the whole run above
            errs = {k: {kk: v[kk] for kk in ("logZ_err", "posterior_mean_err") if kk in v} for k, v in whole.items() if isinstance(v, dict)}
            errs = {k: v for k, v in errs.items() if v}
            if "bayes_factor_rel_err" in whole:
                errs["bayes_factor_rel_err"] = whole["bayes_factor_rel_err"]
            if errs:
                out["errors_vs_exact"] = errs
        if eng.sharded_packed:    # rank 0's device-event breakdown of the sharded sweep
            out["sharded_phases_ms"] = {k: {"calls": c, "avg_ms": (t / c if c else 0.0)} for k, (c, t) in phases.items()}
        if world > 1:
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = "timed on rank 0 at N = 1 only (the contract): see the --gpus 1 line"
        if world == 1 and not args.no_cpu_baseline:
            if getattr(args, "defer_cpu", None) is not None:
                # the default run measures every configuration's GPU window first and the CPU legs afterwards: 256 OpenMP threads
                # that have just finished (and the host memory they touched) disturbed the next configuration's window on some
                # boxes of the pool (a 12 ms window took 29 ms)
                args.defer_cpu.append((out, A, args, cfg))
            else:
                fill_cpu_baseline(out, A, args, cfg)
        return out, pg
    return None, pg


OTHER_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_is", "frac_of_achievable", "achievable_peak_gbs", "hbm_read_frac_algorithmic_bytes",
              "line_fill_bytes_per_update", "line_fill_bytes_per_update_model", "line_fill_bytes_are", "frac_trace", "peak_origin", "avg_launch_ms", "launches", "updates_per_launch",
              "kernel_updates_per_s", "traffic", "traffic_over_moved_bytes", "traffic_source", "acceptance_rate", "flops_per_update",
              "rk4_steps_per_s", "bytes_read_per_update")


def fill_cpu_baseline(out, A, args, cfg):
    faithful, out["cpu_baseline"] = cpu_baseline(A, args, cfg)
    if faithful:
        out["cpu_baseline_reference_faithful"] = faithful


def other_config(args, name):
    """the other single-GPU configurations of BASELINE.json, each with its own engine, window and (reduced) CPU sample"""
    import copy
    import gc

    import torch
    a = copy.copy(args)
    a.config, a.steps, a.warmup, a.particles_per_gpu, a.lanes, a.dim = name, None, None, None, 0, 32
    a.cpu_particles, a.no_pattern = None, True
    a.cpu_steps = {"mc1d": 4, "lv": 3, "evidence1d": 3}[name]
    a.no_whole_run = args.no_whole_run             # every configuration's whole run: logZ / posterior-mean error (evidence1d: both models, the Bayes factor)
    deferred = []
    a.defer_cpu = deferred if getattr(args, "defer_cpu", None) is not None else None
    r, _ = run_config(a)
    gc.collect()
    torch.cuda.empty_cache()
    out = {k: r[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype") if k in r}
    out["workload"] = r["config"]["workload"]
    out["timed_window"] = r["config"]["timed_window"]
    out["roofline"] = {k: r["roofline"][k] for k in OTHER_KEYS if k in r["roofline"]}
    for k in ("cpu_baseline", "whole_run", "errors_vs_exact"):
        if k in r:
            out[k] = r[k]
    for (_, A, aa, cfg) in deferred:               # the CPU leg of this configuration fills the summary, later
        args.defer_cpu.append((out, A, aa, cfg))
    return out


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as ONE child process tree (python -m
    torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) and relay rank 0's JSON line.  This parent has not
    imported torch and never touches a GPU (nothing is exec'ed or re-launched from a process that has); a failing rank
    makes the launcher -- and this process -- exit non-zero; nothing is retried."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout:                  # rank 0 prints one JSON line; anything else on the ranks' stdout (RCCL banners) goes to stderr
        t = ln.strip()
        if t.startswith("{") and t.endswith("}") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 1
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    args.defer_cpu = []            # GPU windows of every configuration first, CPU baselines afterwards
    out, pg = run_config(args)
    if out is not None and args.config == "smc32" and out["n_gpus"] == 1 and not args.no_other_configs and not args.force_collectives:
        # every other single-GPU configuration BASELINE.json names, in the same run (the headline stays configs[2])
        out["other_configs"] = {}
        for name in ("mc1d", "lv", "evidence1d"):
            try:
                out["other_configs"][name] = other_config(args, name)
            except Exception as e:                     # a failing side configuration must not cost the headline line
                out["other_configs"][name] = {"error": f"{type(e).__name__}: {e}"}
    for (o, A, aa, cfg) in args.defer_cpu:
        try:
            fill_cpu_baseline(o, A, aa, cfg)
        except Exception as e:
            o["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    if out is not None:
        try:                      # RCCL's start-up banner sits in libc's stdio buffer: push it out BEFORE the result line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if pg is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
