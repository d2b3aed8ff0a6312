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
    assert roof["bound"] in ("hbm", "valu") and 0 < roof["frac"] < 1.5 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] == "port" and cpu["value"] > 0
