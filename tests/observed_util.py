"""Shared body of the "observations uploaded once" checks (grl_observe / grl_act(GRL_ACT_OBSERVED) /
grl_replay_add_observed): run on the emulation build by the CPU suite and on the MI355X by the GPU suite.

The reference's learn loop (stable-baselines SAC.learn, entered at manipulation_main/training/sb_helper.py:175-177) hands the
same observation to VecNormalize's statistics, to the policy and to two replay rows; the calls below must leave the engine
in exactly the state the separate uploads (grl_norm_update, grl_act, grl_replay_add) leave it in."""
import numpy as np

from grasp_rl import _capi
from grasp_rl.init import init_parameters

RING = ("rp_obs", "rp_next", "rp_dobs", "rp_dnext", "rp_act", "rp_rew", "rp_done")


def _observations(rng, n, shape):
    x = rng.uniform(0.0, 1.0, (n,) + shape).astype(np.float32)
    if len(shape) == 3:
        x[..., -1] = 0.0
        x[:, 0, 0, -1] = rng.uniform(0, 1, n)        # the direct feature rides in the last channel
        if shape[-1] == 5:
            x[..., :3] = np.rint(x[..., :3] * 255.0)
    return x


def check_observed_path(make_engine, extractor="augmented", channels=2, n=6, steps=5, rgb_u8=False):
    if extractor == "mlp":
        shape = (37,)
        kw = dict(obs_dim=37)
    else:
        shape = (64, 64, channels)
        kw = dict(obs_channels=channels, n_direct=1)
    if rgb_u8:
        kw["replay_rgb_u8"] = True
    cfg = lambda: _capi.make_config(extractor, act_dim=5, layers=(64, 64), batch_size=4, replay_capacity=16, normalize=True,
                                    act_batch=n, **kw)
    a, b = make_engine(cfg()), make_engine(cfg())
    try:
        p = init_parameters(a.table, seed=2)
        a.set_parameters(p)
        b.set_parameters(p)
        rng = np.random.default_rng(11)
        obs = _observations(rng, n, shape)
        a.norm_update(obs)
        s0 = b.observe(obs, update_stats=True)
        for step in range(steps):
            eps = rng.standard_normal((n, 5)).astype(np.float32)
            det = step % 2 == 0
            act_a = a.act(obs, deterministic=det, eps=None if det else eps, raw=True)
            act_b = b.act(n, deterministic=det, eps=None if det else eps, raw=True, observed=True)
            assert np.array_equal(act_a, act_b), step
            new = _observations(rng, n, shape)
            rew = rng.normal(size=n).astype(np.float32)
            done = (rng.uniform(size=n) < 0.4).astype(np.float32)
            if step == 1:
                done[:] = 0.0                           # a step without terminal rows
            if step == 2:
                done[:] = 1.0                           # every row terminal
            rows = [i for i in range(n) if done[i] > 0]
            term = _observations(rng, len(rows), shape) if rows else None
            store = new.copy()
            for j, i in enumerate(rows):
                store[i] = term[j]
            a.norm_update(new)
            a.replay_add(obs, act_a, rew, store, done)
            s1 = b.observe(new, update_stats=True)
            assert s1 == s0 + 1
            s0 = s1
            b.replay_add_observed(act_b, rew, done, rows, term)
            assert a.replay_size() == b.replay_size()
            obs = new
        for name in RING:       # (16 slots, 30 transitions: the ring wrapped)
            ra, rb = a.fetch(name), b.fetch(name)
            assert ra.shape == rb.shape and np.array_equal(ra.view(np.uint32), rb.view(np.uint32)), name
        ma, mb = a.get_obs_stats(shape), b.get_obs_stats(shape)
        assert ma[2] == mb[2] and np.array_equal(ma[0], mb[0]) and np.array_equal(ma[1], mb[1])
        # state errors: a different number of observations, and a replay row without two consecutive observe calls
        c = make_engine(cfg())
        try:
            for bad in (lambda: c.replay_add_observed(act_a, rew, done),
                        lambda: c.act(n, deterministic=True, raw=True, observed=True)):
                try:
                    bad()
                except _capi.GrlError:
                    pass
                else:
                    raise AssertionError("expected GrlError")
            c.observe(obs)
            c.observe(obs[:n - 1])
            try:
                c.replay_add_observed(act_a, rew, done)
            except _capi.GrlError:
                pass
            else:
                raise AssertionError("expected GrlError")
            try:
                check = _capi.check
                check(c.lib, c.lib.grl_act(c.h, None, n - 1, 8, None, np.empty((n, 5), np.float32).ctypes.data))
            except _capi.GrlError:
                pass
            else:
                raise AssertionError("unknown flag bits must be refused")
        finally:
            c.close()
    finally:
        a.close()
        b.close()
