"""Shared engine-vs-oracle comparison used by the CPU (host-emulation plan check) and GPU parity tests.

Tolerances: forward tensors |d| <= 1e-5 + 1e-4*|ref|; losses rel 1e-4;
gradients and post-Adam weights: 1e-3 of the per-tensor max-abs (summation order differs);
normalised minibatch tensors and integer paths: bit exact.
"""
from collections import OrderedDict

import numpy as np
import torch

from grasp_rl import _capi, synthetic
from grasp_rl.engine import SacEngine
from oracle import sac as osac

FWD_ATOL, FWD_RTOL = 1e-5, 1e-4
GRAD_REL = 1e-3


def make_case(extractor="augmented", kind="depth", B=8, n_replay=40, act_dim=5, layers=(64, 64), seed=0,
              normalize=True, n_steps=2, obs_dim=101):
    case = {"B": B, "n_steps": n_steps, "extractor": extractor, "normalize": normalize}
    if extractor == "mlp":
        rng = np.random.default_rng(seed + 7)
        mean = rng.uniform(0.3, 0.8, obs_dim)
        var = rng.uniform(0.02, 0.11, obs_dim)
        stats = {"mean": mean, "var": var, "ret_var": 1.7e7}
        tr = synthetic.make_vector_transitions(n_replay, mean, var, act_dim, seed)
        spec = osac.SacSpec(extractor="mlp", obs_dim=obs_dim, act_dim=act_dim, layers=list(layers))
        cfg = _capi.make_config("mlp", obs_dim=obs_dim, act_dim=act_dim, layers=layers, batch_size=B,
                                replay_capacity=n_replay, normalize=normalize, act_batch=4)
    else:
        stats = synthetic.load_obs_stats(kind)
        tr = synthetic.make_transitions(n_replay, kind, act_dim, seed, stats)
        C = stats["mean"].shape[-1]
        if extractor == "augmented":
            spec = osac.SacSpec(extractor="augmented", img_channels=C - 1, n_direct=1, act_dim=act_dim,
                                layers=list(layers))
        else:
            spec = osac.SacSpec(extractor="nature", img_channels=C, n_direct=0, act_dim=act_dim, layers=list(layers))
        cfg = _capi.make_config(extractor, obs_channels=C, n_direct=1 if extractor == "augmented" else 0,
                                act_dim=act_dim, layers=layers, batch_size=B, replay_capacity=n_replay,
                                normalize=normalize, act_batch=4)
    idx, eps = synthetic.make_noise(n_steps, B, act_dim, n_replay, seed + 1)
    case.update(spec=spec, cfg=cfg, stats=stats, tr=tr, idx=idx, eps=eps,
                params=osac.init_params(spec, seed))
    return case


def oracle_run(case):
    """List of per-step diagnostics + final parameters from the CPU oracle."""
    spec = case["spec"]
    orc = osac.SacOracle(spec, case["params"])
    tr, out = case["tr"], []
    for s in range(case["n_steps"]):
        ii = case["idx"][s]
        raw = {k: tr[k][ii] for k in ("obs", "act", "rew", "next_obs", "done")}
        batch = osac.prepare_batch(spec, raw, case["stats"] if case["normalize"] else None)
        d = orc.step(batch, case["eps"][s])
        d["batch"] = batch
        out.append(d)
    return out, orc


def engine_setup(case, backend=None, lib_path=None):
    eng = SacEngine(case["cfg"], backend=backend, lib_path=lib_path)
    eng.set_parameters(case["params"])
    st = case["stats"]
    eng.set_obs_stats(st["mean"], st["var"], st["ret_var"])
    tr = case["tr"]
    eng.replay_add(tr["obs"], tr["act"], tr["rew"], tr["next_obs"], tr["done"])
    return eng


def close(a, ref, atol=FWD_ATOL, rtol=FWD_RTOL, what=""):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    err = np.abs(a - ref) - (atol + rtol * np.abs(ref))
    assert err.max() <= 0, "%s: max excess %.3e (max |d| %.3e, max |ref| %.3e)" % (
        what, err.max(), np.abs(a - ref).max(), np.abs(ref).max())


def close_rel_max(a, ref, rel=GRAD_REL, what="", floor=1e-12):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    scale = max(np.abs(ref).max(), floor)
    d = np.abs(a - ref).max()
    assert d <= rel * scale + 1e-9, "%s: max |d| %.3e vs %.1e * max|ref| %.3e" % (what, d, rel, scale)


def compare_first_step(eng, case, d0):
    """After exactly one engine update: forward tensors, losses and gradients vs the oracle."""
    spec, B, A = case["spec"], case["B"], case["spec"].act_dim
    batch = d0["batch"]
    ldf = (spec.feat_dim + 3) // 4 * 4
    if spec.extractor != "mlp":
        Ci = spec.img_channels if spec.extractor == "augmented" else spec.obs_channels
        x = eng.fetch("x_obs", (B, 64, 64, Ci))
        xn = eng.fetch("x_next", (B, 64, 64, Ci))
        ref = batch["obs"].numpy()[..., :Ci]
        refn = batch["next_obs"].numpy()[..., :Ci]
        assert np.array_equal(x, ref), "normalised obs minibatch is not bit-exact"
        assert np.array_equal(xn, refn), "normalised next_obs minibatch is not bit-exact"
    assert np.array_equal(eng.fetch("rew", (B,)), batch["rew"].numpy()), "normalised reward not bit-exact"
    assert np.array_equal(eng.fetch("done", (B,)), batch["done"].numpy())
    assert np.array_equal(eng.fetch("act", (B, A)), batch["act"].numpy())
    F = spec.feat_dim
    for name, key in (("feat_pi", "h_pi"), ("feat_vf", "h_c"), ("feat_tgt", "h_tgt")):
        close(eng.fetch(name, (B, ldf))[:, :F], d0[key], what=name)
    for name in ("mu", "pi"):
        close(eng.fetch(name, (B, A)), d0[name], what=name)
    # the engine keeps the raw `dense_1` output; the oracle reports it after clip_by_value(-20, 2)
    close(np.clip(eng.fetch("log_std", (B, A)), osac.LOG_STD_MIN, osac.LOG_STD_MAX), d0["log_std"], what="log_std")
    for name in ("logp", "qf1", "qf2", "v", "v_tgt", "qf1_pi", "qf2_pi"):
        close(eng.fetch(name, (B,)), d0[name], atol=2e-5, rtol=2e-4, what=name)
    m = eng.metrics()
    for k_e, k_o in (("policy_loss", "policy_loss"), ("qf1_loss", "qf1_loss"), ("qf2_loss", "qf2_loss"),
                     ("value_loss", "value_loss"), ("ent_coef_loss", "ent_loss"), ("ent_coef", "ent_coef")):
        ref = float(d0[k_o])
        assert abs(m[k_e] - ref) <= 1e-4 * abs(ref) + 1e-6, (k_e, m[k_e], ref)
    G = eng.get_gradients()
    for n, g in d0["grads"].items():
        close_rel_max(G[n], g, what="grad " + n)


def compare_params(eng, orc, lr, n_steps):
    """Post-update weights.  Adam's first steps move every weight by ~lr*sign(g), so the comparison
    is made on the scale of the update, not of the weight: max |d| <= 0.3*lr*n_steps and
    mean |d| <= 0.02*lr*n_steps per tensor (elements whose gradient is ~0 are sign-sensitive)."""
    P = eng.get_parameters()
    for n, ref in orc.P.items():
        d = np.abs(np.asarray(P[n], np.float64) - np.asarray(ref, np.float64))
        assert P[n].shape == ref.shape, n
        assert d.max() <= 0.3 * lr * n_steps + 1e-7, "param %s: max |d| %.3e (lr %.1e)" % (n, d.max(), lr)
        assert d.mean() <= 0.02 * lr * n_steps + 1e-9, "param %s: mean |d| %.3e (lr %.1e)" % (n, d.mean(), lr)
