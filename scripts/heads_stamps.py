#!/usr/bin/env python
"""Development aid: wall-clock stamps (100 MHz) of the phases of heads_fused_kernel's row block 0, per chain type.
python scripts/heads_stamps.py   (sets GRL_TUNE=heads_stamps=1)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["GRL_TUNE"] = "heads_stamps=1"
import numpy as np
import parity_util as pu

case = pu.make_case(extractor="augmented", kind="depth", B=256, n_replay=300, n_steps=1)
eng = pu.engine_setup(case)
for it in range(6):
    eng.train(1)
    eng.synchronize()
    st = eng.fetch("heads_stamps", (4, 32, 2)).view(np.uint32).astype(np.uint64)
    st = (st[..., 0] | (st[..., 1] << np.uint64(32))).astype(np.int64)
    if it >= 3:
        for ty in range(4):
            v = st[ty][st[ty] > 0]
            print("iter %d type %d: phases (us): %s  total %.2f" % (it, ty, np.round(np.diff(v) / 100.0, 2).tolist(), (v[-1] - v[0]) / 100.0))
eng.close()
