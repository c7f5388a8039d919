"""DQN / BDQ engine-vs-oracle comparison shared by the CPU plan test and the GPU parity test."""
import numpy as np
import torch

import parity_util as pu
from grasp_rl import _capi
from grasp_rl.engine import QEngine
from oracle import dqn as od

CASES = {
    "dqn": dict(algo="dqn", obs_dim=20, D=1, bins=6, common=(), branch=(16, 16), value=(16, 16), B=8),
    "dqn_reference_shape": dict(algo="dqn", obs_dim=100, D=1, bins=12, common=(), branch=(64, 64), value=(64, 64), B=32),
    "bdq": dict(algo="bdq", obs_dim=20, D=3, bins=5, common=(16, 16), branch=(8,), value=(8,), B=8),
    "bdq_5_branches": dict(algo="bdq", obs_dim=24, D=5, bins=7, common=(32, 16), branch=(8,), value=(12,), B=9),
    "bdq_reference_shape": dict(algo="bdq", obs_dim=100, D=3, bins=33, common=(64, 64), branch=(32,), value=(32,), B=64),
    # BASELINE configs[2] exactly: gripper_grasp.yaml:104-118 (layers [[64,64],[32],[32]], num_actions_pad 33,
    # batch 64) on the 101-d auto-encoder observation (100 features + gripper width), 5 action dimensions
    "bdq_baseline_config3": dict(algo="bdq", obs_dim=101, D=5, bins=33, common=(64, 64), branch=(32,), value=(32,), B=64, lr=1e-4),
    # ... and as gripper_grasp.yaml:104-118 actually selects it: `prioritized_replay: False` (:106) -- uniform replay, every
    # importance weight 1
    "bdq_baseline_config3_uniform": dict(algo="bdq", obs_dim=101, D=5, bins=33, common=(64, 64), branch=(32,), value=(32,), B=64,
                                         lr=1e-4, uniform=True),
    # the alternative readings of the unavailable bdq_sb fork (oracle/dqn.py switches): TD loss summed over the
    # branches; no 1/(D+1) rescaling of the trunk gradient
    "bdq_loss_sum": dict(algo="bdq", obs_dim=20, D=3, bins=5, common=(16, 16), branch=(8,), value=(8,), B=8, loss_sum=True),
    "bdq_no_trunk_rescale": dict(algo="bdq", obs_dim=20, D=3, bins=5, common=(16, 16), branch=(8,), value=(8,), B=8,
                                 trunk_rescale=False),
}


def make_q_case(algo, obs_dim, D, bins, common, branch, value, B, n_replay=40, n_steps=3, seed=0, lr=1e-3,
                normalize=False, loss_sum=False, trunk_rescale=True, uniform=False):
    rng = np.random.default_rng(seed)
    spec = od.QSpec(algo=algo, obs_dim=obs_dim, n_branches=D, n_bins=bins, common=list(common),
                    branch_hidden=list(branch), value_hidden=list(value), gamma=0.97, lr=lr,
                    loss_sum_branches=loss_sum, trunk_rescale=trunk_rescale)
    cfg = _capi.make_q_config(algo, obs_dim, D, bins, common, branch, value, batch_size=B, act_batch=4,
                              replay_capacity=n_replay, gamma=0.97, lr=lr, normalize=normalize,
                              loss_sum_branches=loss_sum, trunk_rescale=trunk_rescale)
    mean, var = rng.uniform(0.2, 0.8, obs_dim), rng.uniform(0.05, 0.2, obs_dim)
    tr = {"obs": rng.normal(mean, np.sqrt(var), (n_replay, obs_dim)).astype(np.float32),
          "next_obs": rng.normal(mean, np.sqrt(var), (n_replay, obs_dim)).astype(np.float32),
          "act": rng.integers(0, bins, (n_replay, D)).astype(np.float32),
          "rew": rng.normal(0, 2.0, n_replay).astype(np.float32),
          "done": (rng.random(n_replay) < 0.2).astype(np.float32)}
    idx = rng.integers(0, n_replay, (n_steps, B), dtype=np.int64)
    weights = rng.uniform(0.3, 1.0, (n_steps, B)).astype(np.float32)
    if uniform:                   # uniform replay (prioritized_replay False): stable-baselines feeds weights of one
        weights = np.ones((n_steps, B), np.float32)
    return dict(spec=spec, cfg=cfg, tr=tr, idx=idx, weights=weights, params=od.init_params(spec, seed), B=B,
                n_steps=n_steps, stats={"mean": mean, "var": var, "ret_var": 9.0}, normalize=normalize)


def q_engine_setup(case, backend=None, lib_path=None):
    eng = QEngine(case["cfg"], backend=backend, lib_path=lib_path)
    eng.set_parameters(case["params"])
    st = case["stats"]
    eng.set_obs_stats(st["mean"], st["var"], st["ret_var"])
    tr = case["tr"]
    eng.replay_add(tr["obs"], tr["act"], tr["rew"], tr["next_obs"], tr["done"])
    return eng


def _batch(case, s):
    from oracle import sac as osac
    tr, ii = case["tr"], case["idx"][s]
    obs, nxt, rew = tr["obs"][ii], tr["next_obs"][ii], tr["rew"][ii]
    if case["normalize"]:
        st = case["stats"]
        obs = osac.normalize_obs(obs, st["mean"], st["var"])
        nxt = osac.normalize_obs(nxt, st["mean"], st["var"])
        rew = osac.normalize_reward(rew, st["ret_var"])
    f = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    return {"obs": f(obs), "next_obs": f(nxt), "act": f(tr["act"][ii]), "rew": f(rew), "done": f(tr["done"][ii])}


def run_and_compare(case, backend=None, lib_path=None):
    spec = case["spec"]
    orc = od.QOracle(spec, case["params"])
    eng = q_engine_setup(case, backend, lib_path)
    B, D = case["B"], spec.n_branches
    # act path
    obs4 = case["tr"]["obs"][:3]
    pu.close(eng.q_values(obs4), orc.q_values(obs4), atol=2e-5, rtol=2e-4, what="Q-values (act path)")
    for s in range(case["n_steps"]):
        ref = orc.step(_batch(case, s), case["weights"][s])
        eng.compute_grads(case["idx"][s:s + 1], case["weights"][s:s + 1])
        pu.close(eng.td_errors(), ref["td"], atol=3e-5, rtol=2e-4, what="td step %d" % s)
        pu.close(eng.priorities(), ref["priority"], atol=1e-4, rtol=2e-4, what="priority")
        if s == 0:
            G = eng.get_gradients()
            for n, g in ref["grads"].items():
                pu.close_rel_max(G[n], g, what="grad " + n)
            assert abs(eng.metrics()["policy_loss"] - ref["loss"]) <= 1e-4 * abs(ref["loss"]) + 1e-6
        eng.apply_grads(1.0)
        if s == 1:
            eng.update_target()
            orc.update_target()
    pu.compare_params(eng, orc, spec.lr, case["n_steps"])
    P = eng.get_parameters()
    sc = spec.scope
    for n in P:                                             # hard copy happened after step 1 only
        if "/target_q_func/" in n:
            assert not np.array_equal(P[n], P[n.replace("/target_q_func", "")]) or P[n].size <= 1, n
    assert P[sc + "/eps:0"] == case["params"][sc + "/eps:0"]
    # fused call == split calls
    eng2 = q_engine_setup(case, backend, lib_path)
    eng2.train(2, case["idx"][:2], case["weights"][:2])
    eng3 = q_engine_setup(case, backend, lib_path)
    for s in range(2):
        eng3.compute_grads(case["idx"][s:s + 1], case["weights"][s:s + 1])
        eng3.apply_grads(1.0)
    Pa, Pb = eng2.get_parameters(), eng3.get_parameters()
    for n in Pa:
        assert np.array_equal(Pa[n], Pb[n]), n
    for e in (eng, eng2, eng3):
        e.close()


# ---------------------------------------------------------------------------------------------------
# prioritised replay on the device (csrc/per_kernels.h) against oracle/per.py, draw for draw
def _leaves_equal(dev, ref, exact, what):
    """float64 leaves: bit-equal; `exact=False` (GPU, leaves written by `add` after max_priority left 1.0) allows the
    last bit of a FLOAT64 pow to differ between the device's math library and libm."""
    if exact:
        assert np.array_equal(dev, ref), "%s: %d leaves differ, max rel %.3e" % (
            what, int((dev != ref).sum()), float(np.max(np.abs(dev - ref) / np.maximum(np.abs(ref), 1e-300))))
    else:
        assert np.all(np.abs(dev - ref) <= np.spacing(np.abs(ref))), what


def per_check(backend=None, lib_path=None, cap=3000, n_store=2500, B=16, n_steps=5, seed=3, case_name="dqn",
              stratified=False):
    """Fill, then alternate (sample with explicit float64 uniforms -> update -> leaves written back) for n_steps
    rounds, then add across the ring wrap and sample again.  The oracle (stable-baselines' segment trees, restated in
    oracle/per.py) and the device evolve INDEPENDENTLY: the only values handed from one to the other are the
    uniforms and, as in DQN.learn, the TD errors of the minibatch.  After every round the drawn indices must be
    EQUAL and the float64 leaves bit-identical; the importance weights agree to float32 rounding."""
    from oracle.per import PerOracle
    rng = np.random.default_rng(seed)
    c = dict(CASES[case_name])
    c["B"] = B
    case = make_q_case(n_replay=n_store, n_steps=n_steps, **c)
    case["cfg"].replay_capacity = cap
    case["cfg"].q_per, case["cfg"].q_per_alpha, case["cfg"].q_per_eps = 1, 0.6, 1e-6
    case["cfg"].q_per_alpha64, case["cfg"].q_per_stratified = 0.6, int(stratified)
    eng = q_engine_setup(case, backend=backend, lib_path=lib_path)
    on_cpu = backend is not None
    orc = PerOracle(cap, 0.6, 1e-6, stratified=stratified)
    orc.add(n_store)
    p0 = eng.stored_priorities()
    assert p0.dtype == np.float64 and np.array_equal(p0, orc.leaves) and not p0[n_store:].any()

    def one_round(s, beta):
        u = rng.random(B)                               # float64, like np.random.random
        eng.train_per(1, beta, u[None])
        idx = eng.sampled_indices()
        ref_idx, ref_w = orc.sample(u, beta)
        assert np.array_equal(idx, ref_idx), (s, idx, ref_idx)
        assert idx.max() <= orc.size - 2                # the published sum(0, len - 1) never draws the newest transition
        w = eng.importance_weights()
        assert np.allclose(w, ref_w.astype(np.float32), rtol=1e-6, atol=0), (w, ref_w)
        assert w.max() <= 1.0 + 1e-6 and w.min() > 0
        orc.update(idx, eng.priorities())               # |td| of the minibatch just trained on (sum over branches)
        return idx

    for s in range(n_steps):
        one_round(s, 0.4 + 0.1 * s)
        _leaves_equal(eng.stored_priorities(), orc.leaves, True, "leaves after update %d" % s)
    assert float(orc._max_priority) > 1.0              # the TD errors of these cases exceed 1: `add` now takes a float64 power
    # new transitions enter with max_priority ** alpha, ring wrap included; then sampling goes on over the mixed leaves
    tr = case["tr"]
    k = cap - n_store + 7
    eng.replay_add(tr["obs"][:k], tr["act"][:k], tr["rew"][:k], tr["next_obs"][:k], tr["done"][:k])
    orc.add(k)
    _leaves_equal(eng.stored_priorities(), orc.leaves, on_cpu, "leaves after add")
    if on_cpu:
        for s in range(2):
            one_round(n_steps + s, 0.9 + 0.05 * s)
            _leaves_equal(eng.stored_priorities(), orc.leaves, True, "leaves after update %d" % (n_steps + s))
    # device RNG path: runs, indices in range
    eng.train_per(2, 1.0)
    idx = eng.sampled_indices()
    assert idx.min() >= 0 and idx.max() < cap
    tot = float(orc._it_sum.sum())
    eng.close()
    return tot


def uniform_multi_update_check(monkeypatch, name, n_store, backend=None, lib_path=None, n=7):
    """Uniform replay on the device RNG (what gripper_grasp.yaml:106 selects for BDQ): ONE call of n updates -- the index draw and
    the gather of update t + 1 ride on the apply launch of update t, the forward launch opens each update (plan_q "q_pf", four
    launches per update) -- == n calls of one update == the same with GRL_TUNE q_pf=0: parameters, Adam moments, the last drawn
    indices and the metrics bit for bit."""
    def run(split, env=None):
        if env:
            monkeypatch.setenv("GRL_TUNE", env)
        case = make_q_case(n_replay=n_store, n_steps=1, **dict(CASES[name]))
        eng = q_engine_setup(case, backend, lib_path)
        for k in split:
            eng.train_device(k)
        out = (eng.get_parameters(), eng.fetch("adam_m").copy(), eng.fetch("adam_v").copy(), eng.sampled_indices(), eng.metrics())
        eng.close()
        if env:
            monkeypatch.delenv("GRL_TUNE")
        return out
    ref = run([1] * n)
    for got in (run([n]), run([2, n - 2]), run([n - 3, 1, 2]), run([n], env="q_pf=0")):
        for k in ref[0]:
            assert np.array_equal(ref[0][k], got[0][k]), k
        assert np.array_equal(ref[1], got[1]) and np.array_equal(ref[2], got[2])
        assert np.array_equal(ref[3], got[3]) and ref[4] == got[4]


def per_multi_update_check(monkeypatch, name, cap, n_store, backend=None, lib_path=None, n=9):
    """Prioritised replay on the device RNG: ONE call of n updates (the first sums every block of the ring, every apply
    launch rebuilds the blocks its priority write-back touched, the later samplers start from those -- per_refresh_body) ==
    n calls of one update (a block-sum pass over the whole ring in front of every sampler) == the same with GRL_TUNE
    per_inc=0: drawn indices, float64 leaves and parameters bit for bit.  (Several samples per 1024-leaf block, repeated
    indices, the block that holds leaf size - 2.)"""
    def run(split, env=None):
        if env:
            monkeypatch.setenv("GRL_TUNE", env)
        case = make_q_case(n_replay=n_store, n_steps=1, **dict(CASES[name]))
        case["cfg"].replay_capacity = cap
        case["cfg"].q_per, case["cfg"].q_per_alpha, case["cfg"].q_per_eps = 1, 0.6, 1e-6
        case["cfg"].q_per_alpha64 = 0.6
        eng = q_engine_setup(case, backend, lib_path)
        idx = []
        for k in split:
            eng.train_per(k, 0.7)
            idx.append(eng.sampled_indices())
        out = (eng.get_parameters(), eng.stored_priorities(), idx[-1], eng.metrics())
        eng.close()
        if env:
            monkeypatch.delenv("GRL_TUNE")
        return out
    ref = run([1] * n)
    assert np.count_nonzero(ref[1] != ref[1][0]) > min(100, n * 4)            # the priorities moved
    splits = ([n], [2, n - 2], [n - 4, 1, 3])
    for got in [run(list(sp)) for sp in splits] + [run([n], env="per_inc=0")]:
        for k in ref[0]:
            assert np.array_equal(ref[0][k], got[0][k]), k
        assert np.array_equal(ref[1], got[1]) and np.array_equal(ref[2], got[2]) and ref[3] == got[3]
