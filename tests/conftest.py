import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deep-rl-grasping_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle is small PyTorch-CPU work (batch 8 .. 256 convolutions): on a 256-core GPU box the default of one intra-op
    # thread per core made the 200-update trajectory test take 88 s instead of 6.
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    except Exception:
        pass


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def hostemu_lib():
    """TEST-ONLY g++ build of the library's translation units (csrc/*.hip) with the HIP runtime stubbed (tests/hostemu/hostemu.h
    + the reference loops tests/hostemu/*_ref*.h that the kernel headers include only under -DGRL_HOSTEMU): validates the
    host-side launch plan against the oracle without a GPU.  Never used by the product."""
    out = os.path.join(ROOT, "tests", "_build", "libgrl_hostemu.so")
    csrc = os.path.join(PKG, "csrc")
    emu = os.path.join(ROOT, "tests", "hostemu")
    deps = [os.path.join(d, f) for d in (csrc, emu) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h", ".inl"))]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        units = [f[:-4] for f in sorted(os.listdir(csrc)) if f.endswith(".hip")]
        objs = [os.path.join(ROOT, "tests", "_build", u + ".emu.o") for u in units]
        procs = [subprocess.Popen(["g++", "-O2", "-std=c++17", "-fPIC", "-DGRL_HOSTEMU", "-I", emu, "-x", "c++", "-c",
                                   os.path.join(csrc, u + ".hip"), "-o", o]) for u, o in zip(units, objs)]
        if any(p.wait() != 0 for p in procs):
            raise RuntimeError("g++ emulation build failed")
        subprocess.check_call(["g++", "-shared", "-fPIC"] + objs + ["-o", out])
    return out
