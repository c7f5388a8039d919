"""The line `bench.py` prints is what the driver reads: one JSON object with the contract's keys, the BASELINE metric and
workload, a `roofline` whose numbers follow from each other, and a `cpu_baseline` from the oracle (the only place outside
tests/ and smoke() where oracle/ runs -- never inside the timed region)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=e, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _bench("--gpus", "1", "--steps", "40", "--warmup", "5", "--no-learn-loop", "--no-success")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert "grad-steps" in d["metric"] and "grad-steps" in json.dumps(base)      # BASELINE.json's metric
    assert "workload" in d["config"] and "batch 256" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 0.01                 # value = steps / time
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) < 0.05 * r["achieved"]
    assert abs(r["step_frac"] - r["step_flops"] / (d["ms_per_step"] * 1e-3) / 1e12 / r["peak"]) < 2e-3
    assert r["step_flops"] <= r["step_flops_executed"] * (1 + 1e-9)
    assert r["traffic"] is None or r["traffic"] > 0
    for m in r["memory"]:
        assert m["bound"] == "hbm" and m["peak"] == 8000.0 and 0 < m["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert d["value"] > 20 * c["value"]          # a GPU path that routed through the CPU oracle could not be this far ahead


def test_bench_refuses_a_substituted_library():
    e = dict(os.environ, GRL_LIBRARY="/nonexistent/libgrl.so")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"], cwd=ROOT, env=e,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "GRL_LIBRARY" in (out.stderr + out.stdout)
