#!/usr/bin/env python
"""Development aid (GPU): long multi-update calls of every prefetching sequence against their reference runs, bit for bit --
150-update uniform DQN / BDQ calls (plan_q "q_pf") and 400 SAC updates in calls of 37 / 16 / 3 / 11 and as one call with the
image gather riding on the head launch (plan_sac "gather_ride"), depth and RGB-D byte-colour rings.  The suite's tests run the
same checks at 7 - 37 updates; device-only races (workgroups of one launch reading what another of them advances) need
the real scheduler and more repetitions.      python scripts/soak_prefetch.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import parity_util as pu
import q_parity_util as qu

class MP:
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k): os.environ.pop(k, None)

# uniform Q prefetch: one call of 300 updates vs 300 single calls (uniform_multi_update_check does splits too)
for name in ("bdq_baseline_config3_uniform", "dqn_reference_shape"):
    qu.uniform_multi_update_check(MP(), name, 3000, n=150)
    print("q_pf soak ok", name, flush=True)

# SAC ride: 400 updates in calls of 37 / 16 / 3 vs gather_ride=0 in one pattern
for kw in (dict(kind="depth"), dict(kind="rgbd", rgb_u8=True)):
    case = pu.make_case(extractor="augmented", B=256, n_replay=2000, n_steps=1, **kw)
    outs = []
    for tune, split in (("gather_ride=0", [100] * 4), ("gather_ride=1", [37] * 10 + [16, 3, 11]), ("gather_ride=1", [400])):
        os.environ["GRL_TUNE"] = tune
        eng = pu.engine_setup(case)
        for n in split:
            eng.train(n)
        outs.append((eng.get_parameters(), eng.fetch("adam_v").copy(), eng.fetch("idx_raw").copy(), eng.metrics()))
        eng.close()
    os.environ.pop("GRL_TUNE")
    for o in outs[1:]:
        assert all(np.array_equal(outs[0][0][k], o[0][k]) for k in outs[0][0])
        assert np.array_equal(outs[0][1], o[1]) and np.array_equal(outs[0][2], o[2]) and outs[0][3] == o[3]
    print("ride soak ok", kw, flush=True)
