#!/usr/bin/env python3
"""Where the wall time of a generation goes on the host side: per-call wall time of the prologue and of each sweep
(call -> return, i.e. enqueue + device time + read-back) against the device time of the same kernels.
    python tools/host_overhead.py [--per-sweep]"""
import math, sys, time
sys.path.insert(0, ".")
import torch
import abcdez_amd as A
from abcdez_amd.engine import HipEngine

d, N = 32, 1 << 22
spec = A.ModelSpec(A.Factored(*[A.Normal(0.0, 1.0)] * d), A.MVNormal((1.0,) * d), seed=1)
GROUP = "--per-sweep" not in sys.argv          # default: the generation's sweeps as one call (abcdez_smc_sweeps_packed)
e = HipEngine(spec, N, storage="packed")
e.init_population(); e.reset_weights()
eps, eps_k, g0 = math.inf, math.inf, 2.38 / math.sqrt(2 * d)
T = {"prologue": 0.0, "sweep": 0.0, "compact": 0.0, "resample": 0.0}
n = {"prologue": 0, "sweep": 0}
def timed(key, f, *a):
    t = time.perf_counter(); r = f(*a); T[key] += time.perf_counter() - t; n[key] = n.get(key, 0) + 1; return r
for gen in range(40):
    if gen == 10:
        for k in T: T[k] = 0.0
        for k in n: n[k] = 0
        e.ops.set_timing("--no-timing" not in sys.argv)
        torch.cuda.synchronize(); t0 = time.perf_counter()
    eps, wnorm, ess, n_alive, _ = timed("prologue", e.smc_prologue, 0.95, eps, 6.0, eps_k, 0.5 * N)
    if ess < 0.5 * N:
        timed("resample", e.smc_resample); n_alive = N
    timed("compact", e.alive_compact)
    if GROUP:
        timed("sweep", e.smc_sweeps, eps, g0, 1e-5, 3, 1.0, (0.95, 6.0))     # + the next generation's select enqueued ahead
    else:
        for k in range(3):
            timed("sweep", e.smc_swarm, eps, g0, 1e-5)
    eps_k = eps
torch.cuda.synchronize(); wall = time.perf_counter() - t0
ms, launches, units = e.ops.get_timing()
launches = max(launches, 1)
print(f"30 generations: wall {wall*1e3/30:.3f} ms/gen; prologue {T['prologue']*1e3/n['prologue']:.3f} ms/call; "
      f"{'group of 3 sweeps' if GROUP else 'sweep'} call {T['sweep']*1e3/n['sweep']:.3f} ms vs kernel {ms/launches:.3f} ms per sweep; compact {T['compact']*1e3/30:.4f} ms/gen; "
      f"resample total {T['resample']*1e3:.3f} ms over {n.get('resample', 0)} calls")
# cost of an empty host round trip through the library
t = time.perf_counter()
for _ in range(200): e.ops.get_ess(e.wns[:2048])
print(f"tiny reduction + read-back through ctypes: {(time.perf_counter()-t)/200*1e6:.1f} us per call")
t = time.perf_counter()
for _ in range(2000): e._bind_stamps()
print(f"_bind_stamps: {(time.perf_counter()-t)/2000*1e6:.2f} us per call")
