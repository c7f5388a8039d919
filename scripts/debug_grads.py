"""Development aid: per-tensor gradient error of the engine vs the oracle (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import parity_util as pu

def run(B, n_replay, verbose=True):
    case = pu.make_case(extractor="augmented", kind="depth", B=B, n_replay=n_replay, n_steps=1)
    ref, orc = pu.oracle_run(case)
    eng = pu.engine_setup(case)
    eng.train(1, case["idx"][:1], case["eps"][:1])
    G = eng.get_gradients()
    print("== B", B, "n_replay", n_replay)
    for n, g in ref[0]["grads"].items():
        a = np.asarray(G[n], np.float64); r = np.asarray(g, np.float64)
        d = np.abs(a - r)
        rel = d.max() / max(np.abs(r).max(), 1e-12)
        if rel > 1e-5 or verbose:
            print("%-40s max|ref| %.3e  max|d| %.3e  rel %.2e" % (n, np.abs(r).max(), d.max(), rel))
            if rel > 1e-5:
                bad = np.argwhere(d / np.abs(r).max() > 1e-5)
                print("   bad count", len(bad), "of", d.size, "first", bad[:16].tolist())
    for name, key in (("feat_pi", "h_pi"), ("feat_vf", "h_c")):
        ldf = 516
        f = eng.fetch(name, (B, ldf))[:, :513]
        dd = np.abs(f - ref[0][key].numpy() if hasattr(ref[0][key], "numpy") else f - np.asarray(ref[0][key]))
        print(name, "max|d|", dd.max())
    eng.close()

for arg in sys.argv[1:]:
    B, n = arg.split(":")
    run(int(B), int(n), verbose=False)
