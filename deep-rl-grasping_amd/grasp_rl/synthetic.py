"""Seeded synthetic replay contents with the statistics of the reference's real runs.

Per-pixel observation statistics come from the VecNormalize pickles the reference ships
(``trained_models/SAC_depth_1mbuffer/best_model/vecnormalize.pkl`` for depth,
``trained_models/SAC_full_rgbd/vecnormalize.pkl`` for RGB-D); ``scripts/make_golden.py`` extracted
them into ``data/obs_stats_{depth,rgbd}.npz``.  Rewards mimic the shaped reward of
``config/gripper_grasp.yaml:39-47``.
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_obs_stats(kind="depth"):
    """dict(mean[64,64,C], var[64,64,C] float64, ret_var, count) in the env's HWC layout."""
    z = np.load(os.path.join(_DATA, "obs_stats_%s.npz" % kind))
    return {"mean": z["mean"], "var": z["var"], "ret_var": float(z["ret_var"]), "count": float(z["count"])}


def make_transitions(n, kind="depth", act_dim=5, seed=0, stats=None):
    """n raw (un-normalised) transitions in env layout: obs [n,64,64,C] float32 etc."""
    stats = stats or load_obs_stats(kind)
    rng = np.random.default_rng(seed)
    mean, var = stats["mean"], stats["var"]
    C = mean.shape[-1]

    def draw():
        o = rng.normal(mean, np.sqrt(var), (n,) + mean.shape).astype(np.float32)
        if kind == "rgbd":
            o[..., :3] = np.clip(np.round(o[..., :3]), 0, 255)
            o[..., 3] = np.clip(o[..., 3], 0.02, 2.0)
        else:
            o[..., 0] = np.clip(o[..., 0], 0.02, 2.0)
        o[..., C - 1] = 0.0                                   # sensor pad channel (robot.py:199-204)
        o[:, 0, 0, C - 1] = rng.uniform(0.0, 1.0, n)          # gripper width / 0.1 in pixel [0,0]
        return o

    obs, nxt = draw(), draw()
    act = rng.uniform(-1, 1, (n, act_dim)).astype(np.float32)
    u = rng.random(n)
    rew = np.where(u < 0.80, -200.0,
                   np.where(u < 0.99, -100.0 + 1000.0 * rng.uniform(-0.03, 0.03, n), 10000.0)).astype(np.float32)
    done = (rng.random(n) < 1.0 / 15.0).astype(np.float32)
    return {"obs": obs, "act": act, "rew": rew, "next_obs": nxt, "done": done}


def make_vector_transitions(n, mean, var, act_dim=5, seed=0):
    """Vector-observation variant (auto-encoder features, sacMlp)."""
    rng = np.random.default_rng(seed)
    obs = rng.normal(mean, np.sqrt(var), (n,) + mean.shape).astype(np.float32)
    nxt = rng.normal(mean, np.sqrt(var), (n,) + mean.shape).astype(np.float32)
    act = rng.uniform(-1, 1, (n, act_dim)).astype(np.float32)
    rew = rng.normal(-150.0, 300.0, n).astype(np.float32)
    done = (rng.random(n) < 1.0 / 15.0).astype(np.float32)
    return {"obs": obs, "act": act, "rew": rew, "next_obs": nxt, "done": done}


def make_noise(n_steps, batch, act_dim, replay_size, seed=1):
    """Explicit minibatch indices (with replacement, A.1 step 5) and policy noise streams."""
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, replay_size, (n_steps, batch), dtype=np.int64)
    eps = rng.standard_normal((n_steps, batch, act_dim)).astype(np.float32)
    return idx, eps
