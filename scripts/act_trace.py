#!/usr/bin/env python
"""Where the latency of grl_act on uploaded observations goes: 400 calls under `rocprofv3 --kernel-trace`; this script prints
the host-side time per call, scripts/act_trace.sh adds the kernels' durations and the gaps between them from the trace."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd")):
    sys.path.insert(0, p)
import numpy as np
from grasp_rl import _capi
from grasp_rl.engine import SacEngine
from grasp_rl.init import init_parameters

n = 16
cfg = _capi.make_config("augmented", obs_channels=2, n_direct=1, act_dim=5, layers=(64, 64), batch_size=256, replay_capacity=1024,
                        normalize=True, act_batch=n, seed=1)
eng = SacEngine(cfg)
eng.set_parameters(init_parameters(eng.table, seed=0))
obs = np.random.default_rng(0).normal(size=(n, 64, 64, 2)).astype(np.float32)
eng.observe(obs, update_stats=True)
for _ in range(200):
    eng.act(n, True, raw=True, observed=True)
t0 = time.perf_counter()
for _ in range(400):
    eng.act(n, True, raw=True, observed=True)
print("act on observed observations, n = 16: %.1f us per call (host clock)" % (1e6 * (time.perf_counter() - t0) / 400))
eng.close()
