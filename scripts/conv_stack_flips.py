"""Diagnostic for the sample-local convolution stack at the headline batch: is a gradient deviation from the oracle a ReLU
whose pre-activation sits within rounding of zero (two fp32-faithful summation orders disagree about its sign), or an error?
Runs one update of the B = 256 depth case with the stack and with one launch per layer, and compares activations, masks and
gradients with each other and with the oracle.   python scripts/conv_stack_flips.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

torch.set_num_threads(16)
import parity_util as pu  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N_REPLAY = int(sys.argv[2]) if len(sys.argv) > 2 else 600        # (600 / 3: the case of tests/test_gpu_parity.py::test_headline_config_b256)
N_STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
case = pu.make_case(extractor="augmented", kind="depth", B=B, n_replay=N_REPLAY, n_steps=N_STEPS)
ref, orc = pu.oracle_run(case)
d0 = ref[0]
out = {}
for mode in ("1", "0"):
    os.environ["GRL_TUNE"] = "conv_stack=" + mode
    eng = pu.engine_setup(case)
    eng.train(1, case["idx"][:1], case["eps"][:1])
    o = {"a1": eng.fetch("a1_pi", (B * 225 // 2, 64)), "a2": eng.fetch("a2_pi", (B * 36, 64)), "a3": eng.fetch("a3_pi", (B * 16, 64)),
         "g3": eng.fetch("g3_vf", (B * 16, 64)), "g2": eng.fetch("g2_vf", (B * 36, 64)),
         "feat_pi": eng.fetch("feat_pi"), "feat_vf": eng.fetch("feat_vf"), "G": eng.get_gradients()}
    out[mode] = o
    eng.close()
s, l = out["1"], out["0"]
for k in ("a1", "a2", "a3", "feat_pi", "feat_vf", "g3", "g2"):
    a, b = s[k], l[k]
    flips = int(np.sum((a > 0) != (b > 0)))
    print("%-8s stack vs per-layer: max |d| %.3e  (max |x| %.3e)  sign / mask differences %d of %d" % (k, np.abs(a - b).max(), np.abs(b).max(), flips, a.size))
    if flips:
        idx = np.argwhere((a > 0) != (b > 0))[:6]
        for i in idx:
            print("          at %s: stack %.3e  per-layer %.3e" % (tuple(i), a[tuple(i)], b[tuple(i)]))
print("gradients: deviation / (1e-3 max|ref|) against the oracle, stack | per-layer | stack vs per-layer")
for n, g in d0["grads"].items():
    g = np.asarray(g, np.float64)
    sc = max(np.abs(g).max(), 1e-12)
    ds, dl, dd = (np.abs(x - y).max() for x, y in ((s["G"][n], g), (l["G"][n], g), (s["G"][n], l["G"][n])))
    flag = "  <-- over" if ds > 1e-3 * sc + 1e-9 or dl > 1e-3 * sc + 1e-9 else ""
    print("  %-40s %8.4f | %8.4f | %8.4f%s" % (n, ds / (1e-3 * sc), dl / (1e-3 * sc), dd / (1e-3 * sc), flag))
