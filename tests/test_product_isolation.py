"""The product never routes through the checker: nothing under deep-rl-grasping_amd/ imports, loads or names a file of
oracle/ or tests/hostemu/ (except the emulation include switch of the kernel headers, which only -DGRL_HOSTEMU activates),
and bench.py touches oracle/ only inside its cpu_baseline functions."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deep-rl-grasping_amd")


def _py_files(top):
    for d, _, fs in os.walk(top):
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_package_never_imports_the_oracle():
    bad = []
    for path in _py_files(PKG):
        tree = ast.parse(open(path).read(), path)
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            for n in names:
                if n == "oracle" or n.startswith("oracle.") or n.startswith("tests"):
                    bad.append((os.path.relpath(path, ROOT), n))
    assert not bad, bad


def test_native_sources_reach_the_emulation_loops_only_behind_the_hostemu_switch():
    csrc = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".h", ".hip", ".inl")):
            continue
        src = open(os.path.join(csrc, f)).read()
        code = re.sub(r"//.*", "", src)          # (comments cite the oracle files a kernel is checked against)
        assert "oracle" not in "".join(ln for ln in code.splitlines() if "#include" in ln or "dlopen" in ln or "fopen" in ln), f
        # every include of a tests/hostemu reference header sits directly under `#ifdef GRL_HOSTEMU`
        lines = src.splitlines()
        for i, ln in enumerate(lines):
            if re.match(r'\s*#include "[a-z0-9_]+_ref\d\.h"', ln):
                assert any("GRL_HOSTEMU" in lines[j] for j in range(max(0, i - 3), i)), (f, i + 1)


def test_bench_uses_the_oracle_only_for_the_cpu_baseline():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                assert fn.name.startswith("cpu_baseline"), fn.name
            if isinstance(node, ast.Import):
                assert all(a.name.split(".")[0] != "oracle" for a in node.names) or fn.name.startswith("cpu_baseline"), fn.name
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    for n in top:
        mod = n.module if isinstance(n, ast.ImportFrom) else n.names[0].name
        assert (mod or "").split(".")[0] != "oracle"


def test_bench_refuses_a_profile_summary_of_a_different_kernel_generation(tmp_path, monkeypatch):
    """`frac_graph` quotes a COMMITTED rocprofv3 summary; one recorded at a throughput more than 5 % from the live run's
    describes other kernels and must not be quoted."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "r09_rocprofv3_summary_sac_depth.txt").write_text(
        '{"metric": "m", "value": 5000.0, "unit": "grad-steps/s"}\n'
        "launch wgrad_conv       calls   110 avg    30.00 us  min   28.84  total    3300.0 us\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    ms, src = bench.graph_trace_avg("wgrad_conv", "sac_depth", live_value=5100.0)
    assert abs(ms - 0.030) < 1e-9 and src.endswith("r09_rocprofv3_summary_sac_depth.txt")
    ms, why = bench.graph_trace_avg("wgrad_conv", "sac_depth", live_value=5600.0)
    assert ms is None and "refused" in why
    assert bench.graph_trace_avg("wgrad_conv", "no_such_workload", live_value=5000.0) == (None, None)
