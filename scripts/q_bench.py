#!/usr/bin/env python
"""DQN / BDQ update throughput with prioritised replay on the device (SURVEY.md 8d config 3:
gripper_grasp.yaml --algo BDQ: auto-encoder features (101-d), 5 branches x 33 bins, layers [[64,64],[32],[32]],
batch 64, prioritized_replay True) -- development / documentation aid, not the headline bench.

    python scripts/q_bench.py [--replay 1000000] [--steps 2000]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch


def run(algo, replay, steps, prioritized):
    from grasp_rl import _capi
    from grasp_rl.engine import QEngine
    if algo == "bdq":
        cfg = _capi.make_q_config("bdq", 101, 5, 33, common=(64, 64), branch_hidden=(32,), value_hidden=(32,),
                                  batch_size=64, replay_capacity=replay, lr=1e-4, prioritized=prioritized)
        act_dim = 5
    else:
        cfg = _capi.make_q_config("dqn", 101, 1, 12, branch_hidden=(64, 64), value_hidden=(64, 64),
                                  batch_size=32, replay_capacity=replay, lr=5e-4, prioritized=prioritized)
        act_dim = 1
    eng = QEngine(cfg)
    rng = np.random.default_rng(0)
    P = {}
    for name, _, _, shape, _ in eng.table:
        if "/target_q_func/" not in name:
            P[name] = (rng.normal(0.0, 0.1, shape) if len(shape) >= 2 else np.zeros(shape)).astype(np.float32)
    for name, _, _, shape, _ in eng.table:      # target network = online network
        if "/target_q_func/" in name:
            P[name] = P[name.replace("/target_q_func", "")].copy()
    eng.set_parameters(P)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    chunk = 65536
    for k0 in range(0, replay, chunk):
        m = min(chunk, replay - k0)
        with torch.cuda.stream(eng.be.stream):
            obs = torch.randn((m, 101), generator=g, device=dev)
            nxt = torch.randn((m, 101), generator=g, device=dev)
            act = torch.randint(0, 33 if algo == "bdq" else 12, (m, act_dim), generator=g, device=dev).float()
            rew = torch.randn(m, generator=g, device=dev)
            done = (torch.rand(m, generator=g, device=dev) < 1.0 / 15.0).float()
            eng.replay_add_device(obs.contiguous(), act.contiguous(), rew.contiguous(), nxt.contiguous(), done.contiguous())
        eng.be.stream.synchronize()

    def go(n):
        if prioritized:
            eng.train_per(n, beta=0.4)
        else:
            eng.train_device(n)
    go(50)
    eng.synchronize()
    t0 = time.perf_counter()
    go(steps)
    eng.synchronize()
    dt = time.perf_counter() - t0
    eng.profile(True)
    go(50)
    eng.synchronize()
    prof = eng.profile_dump()
    eng.profile(False)
    out = {"algo": algo, "prioritized": prioritized, "replay": replay, "updates_per_s": round(steps / dt, 1),
           "us_per_update": round(1e6 * dt / steps, 2),
           "launch_us": {k: round(1e3 * v["avg_ms"] * v["launches"] / 50.0, 2) for k, v in sorted(prof.items())}}
    eng.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--replay", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=2000)
    a = ap.parse_args()
    for algo in ("bdq", "dqn"):
        for per in (True, False):
            print(json.dumps(run(algo, a.replay, a.steps, per)))
