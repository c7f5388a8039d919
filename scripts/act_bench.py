#!/usr/bin/env python
"""Latency of the two per-env-step inference calls: `grl_act` (actor forward for
N envs, host obs in, host actions out) and `grl_encode` (auto-encoder features for N depth images)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from grasp_rl import _capi
from grasp_rl.engine import SacEngine
from grasp_rl.init import init_parameters

for n in (1, 16, 64):
    cfg = _capi.make_config("augmented", obs_channels=2, n_direct=1, act_dim=5, layers=(64, 64), batch_size=256,
                            replay_capacity=1024, normalize=True, act_batch=n, seed=1)
    eng = SacEngine(cfg)
    eng.set_parameters(init_parameters(eng.table, seed=0))
    obs = np.random.default_rng(0).normal(size=(n, 64, 64, 2)).astype(np.float32)
    for _ in range(2000 if n == 1 else 200):      # (the first engine of the process also wakes the GPU's clocks up)
        eng.act(obs, True)
    t0 = time.perf_counter()
    for _ in range(200):
        eng.act(obs, True)
    dt = (time.perf_counter() - t0) / 200
    print("act   n=%2d: %.1f us per call (%.1f us per env)" % (n, 1e6 * dt, 1e6 * dt / n))
    a = np.zeros((n, 5), np.float32); r = np.zeros(n, np.float32)
    for _ in range(10):
        eng.replay_add(obs, a, r, obs, r)
    t0 = time.perf_counter()
    for _ in range(200):
        eng.replay_add(obs, a, r, obs, r)
    dt = (time.perf_counter() - t0) / 200
    print("replay_add n=%2d: %.1f us per call" % (n, 1e6 * dt))
    # the same three consumers fed by ONE upload per env step (grl_observe): statistics, action, replay rows
    eps = np.zeros((n, 5), np.float32)
    def once():
        eng.observe(obs, update_stats=True)
        eng.replay_add_observed(a, r, r)
        return eng.act(n, False, eps, raw=True, observed=True)
    def separate():
        eng.norm_update(obs)
        eng.replay_add(obs, a, r, obs, r)
        return eng.act(obs, False, eps, raw=True)
    eng.observe(obs)
    for name, fn in (("uploaded once", once), ("separate uploads", separate)):
        for _ in range(20):
            fn()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        dt = (time.perf_counter() - t0) / 200
        print("env step n=%2d, %s (statistics + replay rows + stochastic action): %.1f us" % (n, name, 1e6 * dt))
    for _ in range(20):
        eng.act(n, True, raw=True, observed=True)
    t0 = time.perf_counter()
    for _ in range(200):
        eng.act(n, True, raw=True, observed=True)
    dt = (time.perf_counter() - t0) / 200
    print("act   n=%2d on observed observations: %.1f us per call" % (n, 1e6 * dt))
    t0 = time.perf_counter()
    for _ in range(200):
        eng.observe(obs, update_stats=True)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print("observe n=%2d (+ statistics): %.1f us per call" % (n, 1e6 * dt))
    eng.close()
W = np.load(os.path.join(ROOT, "tests", "golden", "ae_new_gripper_encoder.npz"))
order = ["encoder/conv2d_1/kernel", "encoder/conv2d_1/bias", "encoder/conv2d_2/kernel", "encoder/conv2d_2/bias",
         "encoder/conv2d_3/kernel", "encoder/conv2d_3/bias", "encoder/dense_1/kernel", "encoder/dense_1/bias"]
for n in (1, 16):
    cfg = _capi.make_config("mlp", obs_dim=101, act_dim=5, batch_size=2, replay_capacity=4, act_batch=n)
    eng = SacEngine(cfg)
    eng.load_encoder([W[k] for k in order])
    x = np.random.default_rng(0).uniform(0, 0.5, size=(n, 64, 64, 1)).astype(np.float32)
    for _ in range(20):
        eng.encode(x)
    t0 = time.perf_counter()
    for _ in range(200):
        eng.encode(x)
    dt = (time.perf_counter() - t0) / 200
    print("encode n=%2d: %.1f us per call" % (n, 1e6 * dt))
    eng.close()
