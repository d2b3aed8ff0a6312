"""bench.py's one-line JSON contract (the driver parses it): every configuration, reduced sizes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.mark.parametrize("cfg,particles,steps", [("smc32", 65536, 3), ("mc1d", 65536, 6), ("lv", 4096, 3), ("evidence1d", 65536, 3)])
def test_bench_prints_one_json_line_with_the_contract_fields(cfg, particles, steps):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", str(steps), "--warmup", "1", "--no-whole-run",
           "--particles-per-gpu", str(particles)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["steps"] == steps and d["warmup"] == 1
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f64" and d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"]
    roof, cpu = d["roofline"], d["cpu_baseline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == {"smc32": "hbm", "lv": "valu", "mc1d": "fabric-line-fills", "evidence1d": "fabric-line-fills"}[cfg]
    assert 0 < roof["frac"] < 1.5 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    if cfg == "smc32":       # the ceiling claim as flat scalars of the driver's line
        for k in ("pattern_ceiling_read_frac", "kernel_over_pattern", "achievable_peak_gbs", "frac_of_achievable"):
            assert isinstance(roof[k], float) and roof[k] > 0, k
        assert roof["achievable_peak_gbs"] == 6290.0 and roof["timed_every_nth_step"] == 1
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] == "port" and cpu["value"] > 0


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("cfg,particles", [("smc32", 32768), ("mc1d", 65536)])
def test_bench_two_ranks_print_one_line(cfg, particles):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one JSON line from rank 0, aggregate value), rehearsed
    on the one GPU of the test box: both ranks share it and the collectives go through gloo (`--dist-backend gloo`)."""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--config", cfg, "--particles-per-gpu", str(particles), "--dist-backend", "gloo", "--no-cpu-baseline", "--no-whole-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["value"] > 0
    assert "x2" in d["config"]["parallelism"] and "host transport" in d["config"]["collectives"]
    if cfg == "smc32":
        assert d["config"]["particles_total"] == 2 * particles
        assert set(d["sharded_phases_ms"]) >= {"own_sweep", "flag_allgather", "replay"}


@pytest.mark.parametrize("cfg,extra,scaling,total", [
    ("smc32", ["--particles-per-gpu", "65536"], "weak", 131072),
    ("lv", ["--particles-total", "8192"], "strong", 8192),           # BASELINE configs[3]'s shape: a stated TOTAL split over the ranks
])
def test_bench_starts_its_own_ranks(cfg, extra, scaling, total):
    """`python bench.py --gpus 2 ...` with NO launcher (the form the driver uses for N = 1): the parent starts the ranks itself
    (it never imports torch), relays exactly one JSON line and the exit code.  Rehearsed on one GPU over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--config", cfg, "--steps", "3",
           "--warmup", "1", "--no-whole-run"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == scaling
    assert d["config"]["particles_total"] == total
    assert d["cpu_baseline"] is None and "cpu_baseline_note" in d
    assert set(d["sharded_phases_ms"]) >= {"own_sweep", "flag_allgather", "replay"}
    assert "roofline" in d and d["roofline"]["frac"] > 0


def test_bench_self_launch_reports_a_failing_rank():
    """a rank that fails makes the bare command exit non-zero (no retry, no JSON line)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "1", "--warmup", "0",
           "--particles-per-gpu", "4096", "--lanes", "3"]    # not a power of two: abcdez_ctx_set_lanes refuses on every rank, before any collective
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_bench_whole_run_reports_both_halves_of_the_metric():
    """BASELINE.json's metric: particle-updates/s AND posterior-mean / log-Z error against the closed forms
    (test/runtests.jl:159-162 compares with 30/11 within one posterior std; here the exact finite-eps values)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "evidence1d", "--steps", "2", "--warmup", "1",
           "--particles-per-gpu", "262144", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
    w = d["whole_run"]
    m1, m2 = w["model1_prior_N(0,sqrt10)"], w["model2_prior_N(0,sqrt100)"]
    assert m1["posterior_mean_exact"] == pytest.approx(2.7198461287877933, abs=1e-12)
    assert m2["posterior_mean_exact"] == pytest.approx(2.9694148727465426, abs=1e-12)
    # 2^18 particles: posterior std 0.94 / sqrt(n_alive ~ 1e5) ~ 3e-3; logZ std ~ 5e-3 (profiles/r03_logz_seeds.json scaled)
    assert abs(m1["posterior_mean_err"]) < 0.02 and abs(m2["posterior_mean_err"]) < 0.02
    assert abs(m1["logZ_err"]) < 0.03 and abs(m2["logZ_err"]) < 0.03
    assert abs(w["bayes_factor_rel_err"]) < 0.05
    assert set(d["errors_vs_exact"]) >= {"model1_prior_N(0,sqrt10)", "model2_prior_N(0,sqrt100)", "bayes_factor_rel_err"}


def test_bench_sharded_path_in_a_one_rank_rccl_group():
    """the sharded code path of the bench (flag all-gather + replay + grouped sweeps) over RCCL itself, in a group of one rank"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--particles-per-gpu", "65536",
           "--force-collectives", "--no-cpu-baseline", "--no-whole-run", "--no-pattern"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "sharded_phases_ms" in d and "RCCL" in d["config"]["collectives"]
