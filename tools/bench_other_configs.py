"""ad-hoc timing of the other BASELINE configs on one GPU (not the driver's bench)"""
import sys, time, math, json
sys.path.insert(0, '.')
import torch
import abcdez_amd as A
from abcdez_amd.engine import HipEngine
def t_mc(N=1<<20, gens=100):
    spec=A.ModelSpec(A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), seed=3)
    e=HipEngine(spec,N,storage="classic"); e.init_population(); g0=2.38/math.sqrt(2)
    def gen():
        lo,hi=e.extrema()
        if hi>0.3: e.mc_rank_prepare()
        e.mc_swarm(max(0.3,lo),0.3,g0,1e-5)
    for _ in range(5): gen()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(gens): gen()
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(f"cfg2 abcdemc 1D N={N}: {gens*N/dt:.3e} updates/s, {dt/gens*1e3:.3f} ms/gen")
def t_lv(N=1<<20, gens=10):
    g=json.load(open('tests/golden/lv_data.json'))
    prior=A.Factored(*[A.Uniform(0.0,2.0)]*4)
    sim=A.LotkaVolterraRK4(tuple(g['obs']), x0=g['x0'], y0=g['y0'], dt=0.01, steps_per_obs=100, noise=g['noise'])
    spec=A.ModelSpec(prior,sim,seed=5)
    e=HipEngine(spec,N); t=time.perf_counter(); e.init_population(); torch.cuda.synchronize(); print('lv init s',time.perf_counter()-t)
    e.reset_weights(); eps=e.quantile_alive(0.95); e.smc_reweight(math.inf,eps); e.alive_compact(); g0=2.38/math.sqrt(8)
    e.smc_swarm(eps,g0,1e-5)
    torch.cuda.synchronize(); t=time.perf_counter(); n=0
    for _ in range(gens): e.smc_swarm(eps,g0,1e-5); n+=e.n_alive
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(f"cfg4 LV RK4 (1500 steps/update) N={N}: {n/dt:.3e} updates/s = {n/dt*1500:.3e} RK4 steps/s, {dt/gens*1e3:.2f} ms/sweep")
def t_1d(N=1<<23):
    spec=A.ModelSpec(A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), seed=3)
    t=time.perf_counter()
    r=A.abcdesmc(A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.3, None, nparticles=N, verbose=False, nsims_max=10**12)
    dt=time.perf_counter()-t
    print(f"cfg5-slice abcdesmc 1D N={N}: total {dt:.2f} s, iters {r.iters}, updates {r.updates:.3e} -> {r.updates/dt:.3e} updates/s, logZ {r.logZ:.5f}")
t_mc(); t_lv(); t_1d()
