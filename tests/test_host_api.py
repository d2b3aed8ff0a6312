"""Host-side contract of the drop-in: keyword validation and error behaviour of
abcdesmc! / abcdemc! (src/abcdez_smc.jl:223-235, src/abcdez_mc.jl:107-110), defaults,
result fields, and the C-ABI surface (library loads, exports every declared symbol, fails
loudly without a GPU).  CPU only -- no compute call goes to the HIP library here."""
import ctypes
import inspect
import json
import math
import os
import re

import numpy as np
import pytest
import torch

import abcdez_amd as A
from abcdez_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIOR, SIM = A.Normal(0, math.sqrt(10)), A.Normal1D(3.0)


def test_defaults_match_reference():
    """src/abcdez_smc.jl:215-220, src/abcdez_mc.jl:102-104"""
    p = inspect.signature(A.abcdesmc).parameters
    want = dict(nparticles=100, α=0.95, δess=0.5, nsims_max=10 ** 7, Kmcmc=3, Kmcmc_min=1.0, facc_stop=0.0,
                facc_min=0.0, facc_tune=0.975, verbose=True, verboseout=True, parallel=False)
    for k, v in want.items():
        assert p[k].default == v, k
    assert p["ABCk"].default is A.IndicatorStrict0toϵ
    q = inspect.signature(A.abcdemc).parameters
    assert q["nparticles"].default == 50 and q["generations"].default == 20 and q["parallel"].default is False


@pytest.mark.parametrize("kw,msg", [
    (dict(α=1.0), "α must be in 0 <= α < 1"),
    (dict(α=-0.1), "α must be in 0 <= α < 1"),
    (dict(δess=1.5), "δess must be in 0 <= δess <= 1"),
    (dict(facc_stop=2.0), "facc_stop must be in"),
    (dict(facc_min=-1.0), "facc_min must be in"),
    (dict(facc_tune=1.1), "facc_tune must be in"),
    (dict(Kmcmc=0), "Kmcmc must be at least 1"),
    (dict(Kmcmc_min=-1.0), "Kmcmc_min must be in"),
    (dict(nsims_max=0), "nsims_max must be at least 1"),
    (dict(nparticles=5), "nparticles must be at least 6"),     # ceil(3*1/min(0.95, 0.5)) = 6, smc:234
])
def test_abcdesmc_rejects_bad_keywords(oracle, kw, msg):
    with pytest.raises(ValueError, match=msg):
        A.abcdesmc(PRIOR, SIM, 0.3, None, verbose=False, engine=oracle.oracle_engine, **kw)


def test_abcdesmc_rejects_negative_eps_and_warns(oracle):
    with pytest.raises(ValueError, match="ϵ_target must be non-negative"):
        A.abcdesmc(PRIOR, SIM, -0.1, None, verbose=False, engine=oracle.oracle_engine)
    with pytest.warns(UserWarning, match="Kmcmc_min should be larger than facc_min"):   # smc:232
        A.abcdesmc(PRIOR, SIM, 0.3, None, verbose=False, engine=oracle.oracle_engine, Kmcmc_min=0.1, facc_min=0.2)


@pytest.mark.parametrize("kw,msg", [
    (dict(nparticles=4), "nparticles must be at least 5"),
    (dict(generations=0), "generations must be at least 1"),
])
def test_abcdemc_rejects_bad_keywords(oracle, kw, msg):
    with pytest.raises(ValueError, match=msg):
        A.abcdemc(PRIOR, SIM, 0.3, None, verbose=False, engine=oracle.oracle_engine, **kw)
    with pytest.raises(ValueError, match="ϵ_target must be non-negative"):
        A.abcdemc(PRIOR, SIM, -1.0, None, verbose=False, engine=oracle.oracle_engine)


def test_result_fields(oracle):
    r = A.abcdesmc(PRIOR, SIM, 0.3, None, nparticles=200, verbose=False, engine=oracle.oracle_engine)
    for f in ("P", "Wns", "C", "ϵ", "logZ", "blobs", "ϵs", "ranges_ϵ", "logZs", "esss", "faccs", "γ0s", "Kmcmcs"):
        assert hasattr(r, f), f                                           # smc:388-393
    assert r.P.shape == r.Wns.shape == r.C.shape == (200,)
    assert r.ϵs[0] == math.inf and r.logZs[0] == 0.0 and r.faccs[0] == 1.0 and r.Kmcmcs[0] == 3      # smc:284-292
    assert r.γ0s[0] == 2.38 / math.sqrt(2)                               # smc:280
    assert abs(r.esss[0] - 200) < 1e-9
    r2 = A.abcdesmc(PRIOR, SIM, 0.3, None, nparticles=200, verbose=False, verboseout=False, engine=oracle.oracle_engine)
    assert not hasattr(r2, "ϵs") and r2.logZ == r.logZ
    m = A.abcdemc(PRIOR, SIM, 0.3, None, nparticles=200, generations=5, verbose=False, engine=oracle.oracle_engine)
    for f in ("P", "C", "reached_ϵ", "blobs"):
        assert hasattr(m, f), f                                           # mc:171
    f2 = A.abcdesmc(A.Factored(A.Normal(0, 1), A.Uniform(0, 1)), A.Quad2D(0.0), 5.0, None, nparticles=50, verbose=False,
                    engine=oracle.oracle_engine)
    assert f2.P.shape == (50, 2)


def test_facc_tuning_and_stops(oracle):
    r = A.abcdesmc(PRIOR, SIM, 0.3, None, nparticles=500, verbose=False, engine=oracle.oracle_engine, facc_min=0.9)
    assert r.γ0s[-1] < r.γ0s[0]                                          # smc:320: γ0 *= facc_tune
    r = A.abcdesmc(PRIOR, SIM, 0.0, None, nparticles=500, verbose=False, engine=oracle.oracle_engine, nsims_max=3000)
    assert r.nsims >= 3000 and r.ϵ > 0.0                                 # smc:376: nsims_max stop
    r = A.abcdesmc(PRIOR, SIM, 0.0, None, nparticles=500, verbose=False, engine=oracle.oracle_engine, facc_stop=0.5)
    assert r.faccs[-1] < 0.5                                             # smc:376: facc_stop
    r = A.abcdesmc(PRIOR, SIM, 0.3, None, nparticles=500, verbose=False, engine=oracle.oracle_engine, Kmcmc=5,
                   Kmcmc_min=math.inf)
    assert set(r.Kmcmcs) == {5}                                          # exactly Kmcmc sweeps (docstring smc:187-189)


def test_non_device_simulator_is_rejected():
    with pytest.raises(TypeError, match="DeviceSimulator"):
        A.abcdesmc(PRIOR, lambda th, ve: (abs(th - 3), None), 0.3, None, verbose=False)
    with pytest.raises(ValueError, match="length"):
        A.ModelSpec(A.Factored(A.Normal(0, 1), A.Normal(0, 1)), A.Normal1D(3.0))
    with pytest.raises(TypeError):
        A.abcdesmc(PRIOR, SIM, 0.3, None, ABCk="indicator", verbose=False)


# ------------------------------------------------------------------ the C-ABI library
def declared_symbols():
    text = open(os.path.join(ROOT, "include", "abcdez_hip.h")).read()
    return sorted(set(re.findall(r"ABCDEZ_API\s+[\w\s\*]*?\b(abcdez_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/abcdez_hip.h but not exported"
    bound = set(_lib.PROTOTYPES) | set(_lib.OTHER_SYMBOLS)
    assert set(names) - bound <= {"abcdez_tree_sum"} or set(names) <= bound | {"abcdez_tree_sum"}
    assert _lib.load().abcdez_version() >= 100


def test_abi_layout_of_the_hand_mirrored_structs():
    """abcdez_abi_layout() (sizeof / offsetof as the library was compiled) == the ctypes mirror in abcdez_amd/model.py;
    the Julia shim declares the same fields in the same order and asserts the same call at load time"""
    from abcdez_amd.model import Model, PriorDim

    lib = _lib.load()
    out = (ctypes.c_int32 * 32)()
    n = lib.abcdez_abi_layout(out, 32)
    mine = [ctypes.sizeof(PriorDim)] + [getattr(PriorDim, f).offset for f, _ in PriorDim._fields_] + \
           [ctypes.sizeof(Model)] + [getattr(Model, f).offset for f, _ in Model._fields_]
    assert n == len(mine) == 23 and list(out[:n]) == mine
    jl = open(os.path.join(ROOT, "julia", "ABCdeZHIP.jl"), encoding="utf-8").read()
    body = lambda name: re.search(r"struct %s\n(.*?)\nend" % name, jl, re.S).group(1)
    fields = lambda name: re.findall(r"(\w+)::", re.sub(r"#.*", "", body(name)))
    assert fields("AbzPriorDim") == [f for f, _ in PriorDim._fields_]
    assert fields("AbzModel") == [f for f, _ in Model._fields_]
    assert "abcdez_abi_layout" in jl and "check_abi()" in jl


def _split_top(text):
    """split at the commas that are not inside (), {} or []"""
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _julia_ccalls(jl):
    """[(symbol, return type, [argument types])] of every `ccall((:abcdez_x, LIB), RET, (T1, T2, ...), ...)`"""
    calls = []
    for m in re.finditer(r"ccall\(\(:(abcdez_\w+), LIB\),\s*(\w+),\s*\(", jl):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(jl[i], 0)
            i += 1
        calls.append((m.group(1), m.group(2), _split_top(jl[m.end():i - 1])))
    return calls


# what a C parameter type of include/abcdez_hip.h may be bound to in a ccall signature
def _c_class(ctype):
    t = re.sub(r"\b(const|restrict|__restrict__)\b", "", ctype).strip()
    t = re.sub(r"\s+", " ", t)
    t = re.sub(r"\s*\w+$", "", t) if not t.endswith("*") else t         # drop the parameter name
    t = t.replace(" *", "*").strip()
    ptr = {"abcdez_ctx*": {"Ptr{Cvoid}"}, "abcdez_ctx**": {"Ptr{Ptr{Cvoid}}", "Ref{Ptr{Cvoid}}"},
           "void*": {"Ptr{Cvoid}"}, "void**": {"Ptr{Ptr{Cvoid}}", "Ref{Ptr{Cvoid}}"},
           "double*": {"Ptr{Cvoid}", "Ptr{Float64}", "Ref{Float64}"}, "uint8_t*": {"Ptr{Cvoid}", "Ptr{UInt8}"},
           "uint32_t*": {"Ptr{Cvoid}", "Ptr{UInt32}"}, "uint64_t*": {"Ptr{Cvoid}", "Ptr{UInt64}"},
           "int64_t*": {"Ptr{Int64}", "Ref{Int64}"}, "int32_t*": {"Ptr{Int32}", "Ref{Int32}", "Ref{Cint}", "Ptr{Cint}"},
           "abz_model*": {"Ref{AbzModel}", "Ptr{AbzModel}"}, "char*": {"Cstring"}}
    val = {"int64_t": {"Int64"}, "double": {"Float64"}, "uint32_t": {"UInt32"}, "int32_t": {"Int32", "Cint"},
           "int": {"Cint", "Int32"}, "size_t": {"Csize_t"},
           # function pointers of the host transport (typedefs of the header): a @cfunction pointer or C_NULL
           "abcdez_host_allgather_fn": {"Ptr{Cvoid}"}, "abcdez_host_allreduce_fn": {"Ptr{Cvoid}"}}
    return (ptr if t.endswith("*") else val)[t]


def test_julia_shim_binds_only_exported_symbols_with_the_declared_types():
    """every `ccall((:abcdez_x, LIB), RET, (types...), args...)` of julia/ABCdeZHIP.jl names an exported function and passes
    the argument TYPES include/abcdez_hip.h declares, position by position (a Float64 where the header says int64_t, a
    Ref{Int64} where it says double*, or a missing argument is what a shim nobody can run here would get wrong)"""
    jl = open(os.path.join(ROOT, "julia", "ABCdeZHIP.jl"), encoding="utf-8").read()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "abcdez_hip.h")).read(), flags=re.S)
    calls = _julia_ccalls(jl)
    assert len(calls) >= 20
    for name, ret, types in calls:
        m = re.search(r"ABCDEZ_API\s+([\w\s\*]+?)\b%s\s*\((.*?)\)\s*;" % name, hdr, re.S)
        assert m, f"{name} is not declared in include/abcdez_hip.h"
        c_ret = m.group(1).strip()
        assert ret == {"int": "Cint", "const char*": "Cstring"}.get(c_ret, None), (name, ret, c_ret)
        params = [p.strip() for p in m.group(2).split(",") if p.strip() and p.strip() != "void"]
        assert len(types) == len(params), (name, types, params)
        for k, (jt, cp) in enumerate(zip(types, params)):
            assert jt in _c_class(cp), f"{name}: argument {k + 1} is `{cp}` in the header, bound as {jt}"
    # the callback the shim hands to abcdez_comm_init_host has the C signature of abcdez_host_allgather_fn
    assert re.search(r"typedef int \(\*abcdez_host_allgather_fn\)\(void\* user, void\* buf, int64_t piece_bytes\);", hdr)
    assert "@cfunction(host_allgather_cb, Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int64))" in jl
    for need in ("abcdez_comm_init_host", "abcdez_comm_init", "abcdez_smc_sweeps_sharded", "abcdez_mc_generation_sharded_async",
                 "abcdez_smc_prologue_packed", "abcdez_smc_swarm_packed", "abcdez_smc_resample_gather_packed",
                 "abcdez_packed_gather", "abcdez_ctx_create_user", "abcdez_blob_eval", "abcdez_dev_free", "abcdez_rng_rounds"):
        assert any(c[0] == need for c in calls), need


def test_julia_shim_keeps_the_reference_signatures():
    """src/abcdez_smc.jl:215-220, src/abcdez_mc.jl:102-104: same keyword names and defaults; `rng` takes what the reference
    takes (an AbstractRNG, default Random.default_rng()) and, as a convenience, an Integer key; `parallel` and `varexternal`
    are accepted (and ignored)"""
    jl = open(os.path.join(ROOT, "julia", "ABCdeZHIP.jl"), encoding="utf-8").read()
    smc = re.search(r"function abcdesmc!\(prior, dist!::DeviceSimulator, ϵ_target, varexternal;(.*?)\)\n", jl, re.S).group(1)
    mc = re.search(r"function abcdemc!\(prior, dist!::DeviceSimulator, ϵ_target, varexternal;(.*?)\)\n", jl, re.S).group(1)
    for sig in (smc, mc):
        assert "rng::Union{Integer,AbstractRNG}=Random.default_rng()" in sig and "parallel::Bool=false" in sig
    for kw in ("nparticles::Int=100", "α=0.95", "δess=0.5", "nsims_max::Int=10^7", "Kmcmc::Int=3", "Kmcmc_min=1.0",
               "ABCk=ABCdeZ.IndicatorStrict0toϵ", "facc_stop=0.0", "facc_min=0.0", "facc_tune=0.975", "verbose::Bool=true",
               "verboseout::Bool=true"):
        assert kw in smc, kw
    for kw in ("nparticles::Int=50", "generations::Int=20", "verbose=true"):
        assert kw in mc, kw
    assert "philox_key(rng::AbstractRNG) = rand(rng, UInt64)" in jl and "using ABCdeZ, Distributions, LinearAlgebra, Random" in jl


def test_verbose_lines_carry_the_reference_fields(oracle, caplog):
    """src/abcdez_smc.jl:238-239,372,379 and src/abcdez_mc.jl:113-114,158,164: the `@info` lines of the reference -- a header with
    every keyword, one line per generation with iteration / nsim / ϵ / range_ϵ / ess / facc / logZ, a final line; abcdemc:
    completion / nsim / range_ϵ and the `End:` line -- in the Python drivers (logger `abcdez_amd`) and, statically, in the shim"""
    import logging

    import abcdez_amd as A
    with caplog.at_level(logging.INFO, logger="abcdez_amd"):
        A.abcdesmc(A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.5, None, nparticles=300, verbose=True, rng=3, engine=oracle.oracle_engine)
        A.abcdemc(A.Normal(0, math.sqrt(10)), A.Normal1D(3.0), 0.5, None, nparticles=300, generations=6, verbose=True, rng=3,
                  engine=oracle.oracle_engine)
    text = [r.getMessage() for r in caplog.records]
    head = next(t for t in text if t.startswith("Running abcdesmc! with ϵ_target"))
    for f in ("ϵ_target", "nparticles", "α", "δess", "nsims_max", "Kmcmc", "Kmcmc_min", "ABCk", "facc_stop", "facc_min", "facc_tune", "rng",
              "parallel", "verboseout"):
        assert f + "=" in head, f
    assert any(t.startswith("Running abcdesmc! with executor") for t in text)
    gen = [t for t in text if t.startswith("Finished run: iteration=")]
    assert gen and all(f in gen[0] for f in ("iteration=", "nsim=", "ϵ=", "range_ϵ=", "ess=", "facc=", "logZ="))
    fin = next(t for t in text if t.startswith("Final run:"))
    assert all(f in fin for f in ("iteration=", "nsim=", "ϵ=", "range_ϵ=", "ess=", "facc=", "logZ="))
    mhead = next(t for t in text if t.startswith("Running abcdemc! with ϵ_target"))
    assert all(f + "=" in mhead for f in ("ϵ_target", "nparticles", "generations", "α", "rng", "parallel"))
    assert any(t.startswith("Running abcdemc! with executor") for t in text)
    assert any(t.startswith("Finished run: completion=") and "nsim=" in t and "range_ϵ=" in t for t in text)
    end = next(t for t in text if t.startswith("End:"))
    assert all(f in end for f in ("completion=", "converged=", "nsim=", "range_ϵ="))
    jl = open(os.path.join(ROOT, "julia", "ABCdeZHIP.jl"), encoding="utf-8").read()
    for line in ('@info "Running abcdesmc! with" ϵ_target nparticles α δess nsims_max Kmcmc Kmcmc_min ABCk facc_stop facc_min facc_tune rng parallel verboseout',
                 '@info "Finished run:" iteration = iters nsim = nsims ϵ = ϵ range_ϵ = extrema_dev(e) ess = ess facc = facc logZ = logZ',
                 '@info "Final run:" iteration = iters nsim = nsims ϵ = ϵ range_ϵ = ranges_ϵ[end] ess = ess facc = facc logZ = logZ',
                 '@info "Running abcdemc! with" ϵ_target nparticles generations α rng parallel',
                 '@info "Finished run:" completion = ncomplete nsim = nsims range_ϵ = (ϵ_l, ϵ_h)',
                 '@info "End:" completion = complete converged = conv nsim = nsims range_ϵ = (ϵ_l, ϵ_h)'):
        assert line in jl, line


def test_rng_argument_takes_a_key_or_a_generator():
    """`rng` of the reference signatures (src/abcdez_smc.jl:220): an int is the Philox key itself; a numpy Generator /
    RandomState -- the host-language counterpart of an AbstractRNG -- gives the key with ONE draw, so seeding it makes a run
    reproducible the way it does for the reference's CPU methods (the Julia shim: rand(rng, UInt64))"""
    from abcdez_amd.model import philox_key
    assert philox_key(5) == 5 and philox_key(-1) == 2 ** 64 - 1 and philox_key(2 ** 64 + 3) == 3
    g1, g2 = np.random.default_rng(3), np.random.default_rng(3)
    k = philox_key(g1)
    assert 0 <= k < 2 ** 64 and k == philox_key(g2) and philox_key(g1) != k          # one draw per call
    assert philox_key(np.random.RandomState(1)) == philox_key(np.random.RandomState(1))
    assert A.ModelSpec(PRIOR, SIM, seed=np.random.default_rng(3)).seed == k


def test_roctx_ranges_cost_nothing_unless_asked_for(monkeypatch):
    """abcdez_amd/_trace.py: a shared no-op object unless ABZ_ROCTX=1 / a rocprofv3 run; with ABZ_ROCTX=1 the roctx library
    loads and push / pop are balanced"""
    from abcdez_amd import _trace
    monkeypatch.delenv("ABZ_ROCTX", raising=False)
    monkeypatch.setenv("LD_PRELOAD", "")
    for k in [k for k in os.environ if k.startswith("ROCPROF")]:
        monkeypatch.delenv(k)
    monkeypatch.setattr(_trace, "_ENABLED", None)
    assert _trace.enabled() is False and _trace.rng("a") is _trace.rng("b")
    monkeypatch.setenv("ABZ_ROCTX", "1")
    monkeypatch.setattr(_trace, "_ENABLED", None)
    if _trace.enabled():               # the roctx library of the ROCm image
        with _trace.rng("outer"):
            with _trace.rng("inner"):
                pass
    monkeypatch.setattr(_trace, "_ENABLED", None)


def test_the_rule_between_rank_and_rejection_draws_is_one_function_everywhere(oracle):
    """include/abcdez_spec.h, abz_mc_draws_by_rejection: library (a pure function of the C ABI: callable without a GPU), oracle
    and the closed statement `16 (N - n_above) >= N` agree at and around the boundary for small and large populations."""
    lib = _lib.load()
    L = oracle.lib()
    for N in (5, 6, 15, 16, 17, 100, 3000, 4096, 65537, 1 << 20, (1 << 31) - 1):
        k = -(-N // 16)                                    # ceil(N / 16): the smallest number at or below eps_target that qualifies
        for n_above in {0, 1, N - k - 1, N - k, N - k + 1, N - 1, N} - {-1}:
            if not 0 <= n_above <= N:
                continue
            want = 16 * (N - n_above) >= N
            assert bool(lib.abcdez_mc_draws_by_rejection(n_above, N)) == want == bool(L.orc_mc_draws_by_rejection(n_above, N)), (N, n_above)
        assert lib.abcdez_mc_draws_by_rejection(N, N) == 0           # nobody at or below eps_target: by rank
        assert lib.abcdez_mc_draws_by_rejection(N - k, N) == 1 and lib.abcdez_mc_draws_by_rejection(N - k + 1, N) == 0
    assert lib.abcdez_mc_draws_by_rejection(-1, 10) == -1 and lib.abcdez_mc_draws_by_rejection(11, 10) == -1
    assert lib.abcdez_mc_draws_by_rejection(0, 0) == -1


def test_header_cites_the_reference_for_every_entry_point():
    text = open(os.path.join(ROOT, "include", "abcdez_hip.h")).read()
    for ref in ("src/abcdez_init.jl:2-22", "src/abcdez_smc.jl:106-153", "src/abcdez_smc.jl:59-83",
                "src/abcdez_smc.jl:15-56", "src/abcdez_smc.jl:85-104", "src/abcdez_smc.jl:301",
                "src/abcdez_mc.jl:5-61", "src/abcdez_types.jl:20-23"):
        assert ref in text, ref


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    """no CPU fallback: without a HIP device the default engine must raise, not compute"""
    with pytest.raises(_lib.AbcdezError, match="no HIP device|no CPU fallback"):
        A.abcdesmc(PRIOR, SIM, 0.3, None, nparticles=100, verbose=False)
    with pytest.raises(_lib.AbcdezError):
        A.abcdemc(PRIOR, SIM, 0.3, None, verbose=False)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "abcdez.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "liboracle" not in src and "oracle/" not in src.replace("the oracle/", ""), f


def test_public_wsample_stratified(oracle):
    """test/runtests.jl:13-19 uses ABCdeZ.wsample_stratified! directly (weighted posterior extraction)"""
    rng = np.random.default_rng(0)
    w = rng.random(1000)
    w[rng.random(1000) < 0.3] = 0
    w /= w.sum()
    spec = A.ModelSpec(PRIOR, SIM, seed=5)
    eng = oracle.oracle_engine(spec, 1000)
    inds = A.wsample_stratified(w, rng=5, engine=eng)
    assert inds.shape == (1000,) and (np.diff(inds) >= 0).all() and (w[inds] > 0).all()
    with pytest.raises(ValueError, match="Sum of weights"):
        A.wsample_stratified(w * 2, engine=eng)


def test_prior_family_ids_agree_between_header_python_and_julia():
    """include/abcdez_spec.h (ABZ_PRIOR_*), abcdez_amd/priors.py (PRIOR_*) and the descriptor() methods of julia/ABCdeZHIP.jl name
    the same family with the same number and the same discrete flag"""
    from abcdez_amd import priors

    hdr = open(os.path.join(ROOT, "include", "abcdez_spec.h"), encoding="utf-8").read()
    ids = {m.group(1): int(m.group(2)) for m in re.finditer(r"ABZ_PRIOR_(\w+) = (\d+)", hdr)}
    last = ids.pop("LAST")
    assert sorted(ids.values()) == list(range(last + 1))
    for name, n in ids.items():
        assert getattr(priors, "PRIOR_" + name) == n, name
    jl = open(os.path.join(ROOT, "julia", "ABCdeZHIP.jl"), encoding="utf-8").read()
    julia_name = {"NORMAL": "Normal", "UNIFORM": "Uniform", "DUNIFORM": "DiscreteUniform", "BETA": "Beta", "NEGBIN": "NegativeBinomial",
                  "EXPONENTIAL": "Exponential", "GAMMA": "Gamma", "LOGNORMAL": "LogNormal", "CAUCHY": "Cauchy", "LAPLACE": "Laplace",
                  "WEIBULL": "Weibull", "INVGAMMA": "InverseGamma", "TRUNCNORMAL": "Truncated{<:Normal}", "LOGISTIC": "Logistic",
                  "TDIST": "TDist", "PARETO": "Pareto", "POISSON": "Poisson", "BINOMIAL": "Binomial"}
    py_class = {"NORMAL": priors.Normal(), "UNIFORM": priors.Uniform(), "DUNIFORM": priors.DiscreteUniform(), "BETA": priors.Beta(),
                "NEGBIN": priors.NegativeBinomial(), "EXPONENTIAL": priors.Exponential(), "GAMMA": priors.Gamma(),
                "LOGNORMAL": priors.LogNormal(), "CAUCHY": priors.Cauchy(), "LAPLACE": priors.Laplace(), "WEIBULL": priors.Weibull(),
                "INVGAMMA": priors.InverseGamma(), "TRUNCNORMAL": priors.TruncatedNormal(), "LOGISTIC": priors.Logistic(),
                "TDIST": priors.TDist(), "PARETO": priors.Pareto(), "POISSON": priors.Poisson(), "BINOMIAL": priors.Binomial()}
    assert set(julia_name) == set(ids) - {"PAD", "TRUNCATED", "MIXTURE", "AFFINE"}
    af = priors.Affine(priors.TDist(4.0), 1.0, 2.0)
    assert af.descriptor_at(3)[:3] == (ids["AFFINE"], 0, 3.0) and len(af.ext_record()) == 4 + 7
    assert "AbzPriorDim(21, 0, Float64(off), 0.0, 0.0, 0.0, 0.0)" in jl and re.search(r"#define ABZ_EXT_AFFINE \(4 \+ ABZ_EXT_DESC\)", hdr)
    # the wrapper families: their descriptors point into the ext table (descriptor! in the shim, descriptor_at here)
    tr, mx = priors.truncated(priors.Gamma(2.0, 1.0), 0.5, 4.0), priors.MixtureModel([priors.Normal(), priors.Laplace()], [0.3, 0.7])
    assert tr.descriptor_at(5)[:3] == (ids["TRUNCATED"], 0, 5.0) and mx.descriptor_at(7)[:4] == (ids["MIXTURE"], 0, 2.0, 7.0)
    assert len(tr.ext_record()) == 3 + 7 and len(mx.ext_record()) == 2 * (2 + 7)
    assert "AbzPriorDim(19, par.discrete, Float64(off), 0.0, 0.0, 0.0, 0.0)" in jl and "AbzPriorDim(20, ds[1].discrete, Float64(K), Float64(off)" in jl
    assert re.search(r"#define ABZ_EXT_TRUNC \(3 \+ ABZ_EXT_DESC\)", hdr) and re.search(r"#define ABZ_EXT_MIXC \(2 \+ ABZ_EXT_DESC\)", hdr)
    for name, jn in julia_name.items():
        at = jl.index("descriptor(p::%s)" % jn)
        m = re.search(r"AbzPriorDim\((\d+), (\d),", jl[at:])
        assert m and int(m.group(1)) == ids[name], (name, m and m.group(0))
        d = py_class[name].descriptor()
        assert d[0] == ids[name] and int(m.group(2)) == d[1] == int(py_class[name].discrete), name


def test_shim_block_structure_balances():
    """no Julia here to parse julia/ABCdeZHIP.jl: at least every block opener (function, if, for, while, try, struct, let, begin, do,
    module, `abstract type`) must meet its `end`, function definitions must sit directly inside the module, and the module must close on
    the last line -- what an edit that drops or doubles an `end` breaks first"""
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "julia", "ABCdeZHIP.jl"), encoding="utf-8").read()
    stack = []
    opener = re.compile(r"(?<![\w.:@])(abstract type|function|if|for|while|try|struct|let|begin|do|quote|module|macro|end)(?![\w!])")
    for no, line in enumerate(src.splitlines(), 1):
        code = re.sub(r"#.*$", "", re.sub(r'"(?:\\.|[^"\\])*"', '""', line))
        for m in opener.finditer(code):
            pre, tok = code[:m.start()], m.group(1)
            in_brackets = pre.count("[") - pre.count("]") > 0
            in_call = pre.count("(") - pre.count(")") > 0
            if tok == "end":
                if in_brackets:
                    continue                      # a[end]
                assert stack, f"line {no}: `end` without an opener"
                stack.pop()
            elif tok in ("for", "if") and (in_brackets or in_call):
                continue                          # comprehension / generator
            else:
                if tok == "function" and not line.startswith(" "):
                    assert [t for t, _ in stack] == ["module"], f"line {no}: a top-level function inside {stack}"
                stack.append((tok, no))
    assert not stack, f"unclosed blocks: {stack}"
    assert src.rstrip().splitlines()[-1].startswith("end")


def test_differential_campaign_tool_runs_its_cases():
    """tools/fuzz_parity.py (the differential campaign of profiles/r06_fuzz_parity_summary.json) with the oracle on both sides: the case
    generator, the driver keywords it exercises and its comparison code stay runnable where there is no GPU"""
    import subprocess
    import sys

    root = os.path.join(os.path.dirname(__file__), "..")
    env = dict(os.environ, ABZ_FUZZ_SELFTEST="1")
    for extra in ([], ["--small"]):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "--cases", "12", "--first", "7000", *extra],
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        last = json.loads(r.stdout.strip().splitlines()[-1])
        assert last["summary"] and last["ran"] == 12 and last["different"] == 0
