import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abcdez_amd as A
prior = A.Factored(A.NegativeBinomial(900 / 195, (900 / 195) / (30 + 900 / 195)), A.Beta(15, 2)); sim = A.Socks(0, 11)
for mi in (100, 400, 800, 1717):
    t = time.time()
    r = A.abcdesmc(prior, sim, 0.01, None, nparticles=3000, verbose=False, rng=13, ABCk=A.Indicator0toϵ, max_iters=mi)
    print("max_iters", mi, "seconds", round(time.time() - t, 3), "iters", r.iters, "eps", r.ϵ, "nsims", r.nsims, flush=True)
import cProfile, pstats
cProfile.run("A.abcdesmc(prior, sim, 0.01, None, nparticles=3000, verbose=False, rng=13, ABCk=A.Indicator0toϵ, max_iters=400)", "/tmp/prof.out")
pstats.Stats("/tmp/prof.out").sort_stats("cumulative").print_stats(18)
