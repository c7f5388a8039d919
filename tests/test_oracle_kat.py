"""Known-answer tests that pin the oracle to INDEPENDENT restatements of the published algorithms.

The reference's own tests do not touch this path (tests_gripper/test_sim.py checks the simulator only) and
its arithmetic lives in un-vendored TF 1.14 / stable-baselines 2.10.1 / Keras 2.2.4, so the oracle cannot
be run against the reference here ("parity unpinned" by reference tests).  Next to
the fixture-derived pins of test_oracle_golden.py, every building block of oracle/ is cross-checked here
against a second implementation that shares no code with it: plain NumPy loops, SciPy distributions, a
sum-tree sampler written from the stable-baselines algorithm, hand-derived closed forms, and float64
finite differences of the scalar losses for the gradients.
"""
import numpy as np
import pytest
import torch
from scipy import stats

from oracle import autoencoder as oae
from oracle import dqn as odqn
from oracle import per as oper
from oracle import sac as osac


# ----------------------------------------------------------------------------------------------- convolutions
def _conv_loops(x, w, b, stride, pad_lo=0, pad_hi=0):
    """Direct NHWC / HWIO cross-correlation with explicit zero padding, written as tap loops."""
    n, H, W, C = x.shape
    kh, kw, _, co = w.shape
    xp = np.zeros((n, H + pad_lo + pad_hi, W + pad_lo + pad_hi, C), np.float64)
    xp[:, pad_lo:pad_lo + H, pad_lo:pad_lo + W] = x
    oh = (xp.shape[1] - kh) // stride + 1
    ow = (xp.shape[2] - kw) // stride + 1
    y = np.zeros((n, oh, ow, co), np.float64)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + stride * (oh - 1) + 1:stride, j:j + stride * (ow - 1) + 1:stride, :]
            y += patch @ w[i, j].astype(np.float64)
    return y + np.asarray(b, np.float64).reshape(-1)


@pytest.mark.parametrize("k,s,cin,cout,hw", [(8, 4, 1, 32, 64), (4, 2, 32, 64, 15), (3, 1, 64, 64, 6), (8, 4, 4, 32, 64)])
def test_valid_convolution_layers_of_the_extractor(k, s, cin, cout, hw):
    """custom_obs_policy.py:34-36: conv(8,4) -> conv(4,2) -> conv(3,1), VALID, NHWC x HWIO."""
    rng = np.random.default_rng(k * 100 + cin)
    x = rng.normal(size=(2, hw, hw, cin)).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32)
    got = osac._conv_nhwc(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), s).numpy()
    want = _conv_loops(x, w, b, s)
    assert got.shape == want.shape
    assert np.allclose(got, want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("k,s,hw,lo,hi", [(7, 2, 64, 2, 3), (5, 2, 32, 1, 2), (3, 2, 16, 0, 1), (3, 1, 16, 1, 1), (7, 1, 64, 3, 3)])
def test_tf_same_convolution_of_the_autoencoder(k, s, hw, lo, hi):
    """TF 'SAME': out = ceil(in / s), total padding (out-1) s + k - in, the extra pixel goes AFTER (high side)."""
    assert oae.tf_same_pad(hw, k, s) == (lo, hi)
    rng = np.random.default_rng(k + s + hw)
    x = rng.normal(size=(2, hw, hw, 3)).astype(np.float32)
    w = rng.normal(size=(k, k, 3, 5)).astype(np.float32) / k
    b = rng.normal(size=(5,)).astype(np.float32)
    got = oae._conv_same(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), s).numpy()
    want = _conv_loops(x, w, b, s, lo, hi)
    assert got.shape == want.shape == (2, -(-hw // s), -(-hw // s), 5)
    assert np.allclose(got, want, atol=2e-5, rtol=1e-5)


def test_flatten_order_and_direct_feature_of_the_augmented_extractor():
    """conv_to_fc flattens NHWC row-major ((h*4 + w)*64 + c); the direct feature is pad[0, 0] (robot.py:199-204)."""
    spec = osac.SacSpec()
    P = {k: torch.as_tensor(np.asarray(v)) for k, v in osac.init_params(spec, seed=1).items()}
    rng = np.random.default_rng(2)
    x = rng.uniform(0, 1, size=(3, 64, 64, 2)).astype(np.float32)
    keep = {}
    h = osac.extractor_fwd(spec, P, "model/pi", torch.from_numpy(x), keep).numpy()
    n1, n2, n3, nf = osac.cnn_names(spec)
    g = lambda n, s: P["model/pi/%s/%s:0" % (n, s)].numpy()
    a = np.maximum(_conv_loops(x[..., :1], g(n1, "w"), g(n1, "b"), 4), 0)
    a = np.maximum(_conv_loops(a, g(n2, "w"), g(n2, "b"), 2), 0)
    a = np.maximum(_conv_loops(a, g(n3, "w"), g(n3, "b"), 1), 0)
    flat = np.stack([a[i].reshape(-1) for i in range(3)])              # C-order over (h, w, c)
    fc = np.maximum(flat @ g(nf, "w").astype(np.float64) + g(nf, "b"), 0)
    assert h.shape == (3, 513)
    assert np.allclose(h[:, :512], fc, atol=1e-4, rtol=1e-4)
    assert np.array_equal(h[:, 512], x[:, 0, 0, 1])


# ----------------------------------------------------------------------------------------------- policy
def test_squashed_gaussian_log_prob_and_entropy():
    """SB sac/policies.py: gaussian_likelihood with EPS in the std, entropy of the un-squashed Gaussian,
    squash correction - sum log(1 - tanh(u)^2 + 1e-6)."""
    spec = osac.SacSpec(extractor="mlp", obs_dim=7, act_dim=3, layers=(8,))
    P = {k: torch.as_tensor(np.asarray(v)) for k, v in osac.init_params(spec, seed=3).items()}
    rng = np.random.default_rng(4)
    obs = rng.normal(size=(5, 7)).astype(np.float32)
    eps = rng.normal(size=(5, 3)).astype(np.float32)
    out = osac.actor_fwd(spec, P, torch.from_numpy(obs), torch.from_numpy(eps))
    mu, ls = out["mu"].numpy().astype(np.float64), out["log_std"].numpy().astype(np.float64)
    assert ls.min() >= -20 and ls.max() <= 2
    std = np.exp(ls)
    u = mu + eps * std
    lp = stats.norm.logpdf(u, loc=mu, scale=std).sum(1)                 # EPS = 1e-6 in the std is below f32 resolution here
    lp -= np.log(1 - np.tanh(u) ** 2 + 1e-6).sum(1)
    assert np.allclose(out["logp"].numpy(), lp, atol=2e-4, rtol=1e-5)
    assert np.allclose(out["entropy"].numpy(), stats.norm.entropy(loc=mu, scale=std).sum(1), atol=1e-5)
    assert np.allclose(out["pi"].numpy(), np.tanh(u), atol=1e-6) and np.allclose(out["det"].numpy(), np.tanh(mu), atol=1e-6)


def _tiny_batch(spec, B, seed):
    rng = np.random.default_rng(seed)
    batch = {"obs": rng.normal(size=(B, spec.obs_dim)).astype(np.float32),
             "next_obs": rng.normal(size=(B, spec.obs_dim)).astype(np.float32),
             "act": rng.uniform(-1, 1, size=(B, spec.act_dim)).astype(np.float32),
             "rew": rng.normal(size=(B,)).astype(np.float32), "done": (rng.uniform(size=B) < 0.3).astype(np.float32)}
    return {k: torch.from_numpy(v) for k, v in batch.items()}, rng.normal(size=(B, spec.act_dim)).astype(np.float32)


def test_sac_loss_values_against_numpy_closed_forms():
    """A.4: qf loss 0.5 mean (q - (r + (1-d) gamma V_target(s')))^2, v loss 0.5 mean (v - (min q_pi - alpha logp))^2,
    policy loss mean(alpha logp - qf1_pi), entropy-coefficient loss -mean(log_alpha (logp + target_entropy))."""
    spec = osac.SacSpec(extractor="mlp", obs_dim=6, act_dim=2, layers=(8, 8))
    orc = osac.SacOracle(spec, seed=5)
    batch, eps = _tiny_batch(spec, 9, 6)
    out = orc.forward(batch, torch.from_numpy(eps))
    g = lambda k: out[k].detach().numpy().astype(np.float64)
    alpha = float(np.exp(orc.P["model/log_ent_coef:0"]))
    r, d = batch["rew"].numpy().astype(np.float64), batch["done"].numpy().astype(np.float64)
    qb = r + (1 - d) * spec.gamma * g("v_tgt")
    vb = np.minimum(g("qf1_pi"), g("qf2_pi")) - alpha * g("logp")
    assert np.isclose(float(out["qf1_loss"]), 0.5 * np.mean((g("qf1") - qb) ** 2), rtol=1e-5)
    assert np.isclose(float(out["qf2_loss"]), 0.5 * np.mean((g("qf2") - qb) ** 2), rtol=1e-5)
    assert np.isclose(float(out["value_loss"]), 0.5 * np.mean((g("v") - vb) ** 2), rtol=1e-5)
    assert np.isclose(float(out["policy_loss"]), np.mean(alpha * g("logp") - g("qf1_pi")), rtol=1e-5)
    assert np.isclose(float(out["ent_loss"]), -np.mean(np.log(alpha) * (g("logp") + spec.target_entropy)), rtol=1e-5, atol=1e-7)
    assert spec.target_entropy == -2.0                                   # 'auto': -prod(action_space.shape)


def test_sac_gradient_sets_by_finite_differences():
    """Each of the three optimisers differentiates ITS loss w.r.t. ITS variables with the other terms held
    constant (stop-gradients of A.4): float64 central differences of the scalar losses, parameter by parameter."""
    spec = osac.SacSpec(extractor="mlp", obs_dim=4, act_dim=2, layers=(5,))
    orc = osac.SacOracle(spec, seed=7)
    batch, eps = _tiny_batch(spec, 6, 8)
    eps_t = torch.from_numpy(eps)
    _, G = orc.grads(batch, eps_t)
    P0 = {k: v.copy() for k, v in orc.P.items()}
    b64 = {k: v.double() for k, v in batch.items()}

    def terms(P):
        T = {k: torch.tensor(v, dtype=torch.float64) for k, v in P.items()}
        a = osac.actor_fwd(spec, T, b64["obs"], eps_t.double())
        c = osac.critic_fwd(spec, T, "model/values_fn", b64["obs"], b64["act"], a["pi"])
        ct = osac.critic_fwd(spec, T, "target/values_fn", b64["next_obs"])
        return T, a, c, ct

    # the quantities stable-baselines wraps in tf.stop_gradient are frozen at the base point: finite
    # differences see the forward value of a detached term, so it has to be a constant here
    T0, a0, c0, ct0 = terms({k: v.astype(np.float64) for k, v in P0.items()})
    alpha0 = torch.exp(T0["model/log_ent_coef:0"])
    qb0 = b64["rew"] + (1 - b64["done"]) * spec.gamma * ct0["v"]
    vb0 = torch.minimum(c0["qf1_pi"], c0["qf2_pi"]) - alpha0 * a0["logp"]
    lp0 = a0["logp"] + spec.target_entropy

    def losses(P):
        """float64 restatement of A.4 (three scalar losses, one per optimiser)."""
        T, a, c, _ = terms(P)
        policy = torch.mean(alpha0 * a["logp"] - c["qf1_pi"])
        values = 0.5 * torch.mean((c["qf1"] - qb0) ** 2) + 0.5 * torch.mean((c["qf2"] - qb0) ** 2) + 0.5 * torch.mean((c["v"] - vb0) ** 2)
        ent = -torch.mean(T["model/log_ent_coef:0"] * lp0)
        return {"policy": float(policy), "values": float(values), "ent": float(ent)}

    g_pi, g_vf, g_ent = osac.trainable_groups(spec)
    assert g_ent == ["model/log_ent_coef:0"] and all(n.startswith("model/pi/") for n in g_pi)
    rng = np.random.default_rng(9)
    h = 1e-5
    for tag, names in (("policy", g_pi), ("values", g_vf), ("ent", g_ent)):
        for name in names:
            flat = P0[name].reshape(-1)
            for j in rng.choice(flat.size, size=min(3, flat.size), replace=False):
                Pp = {k: v.astype(np.float64) for k, v in P0.items()}
                Pm = {k: v.astype(np.float64) for k, v in P0.items()}
                Pp[name].reshape(-1)[j] += h
                Pm[name].reshape(-1)[j] -= h
                fd = (losses(Pp)[tag] - losses(Pm)[tag]) / (2 * h)
                got = float(np.asarray(G[name]).reshape(-1)[j])
                assert abs(got - fd) <= 2e-4 * max(1.0, abs(fd)) + 2e-5, (tag, name, int(j), got, fd)


# ----------------------------------------------------------------------------------------------- optimisers
def test_tf_adam_two_steps_closed_form():
    """TF-1.x AdamOptimizer: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); theta -= lr_t m / (sqrt(v) + eps)."""
    P = {"w:0": np.array([1.0, -2.0, 0.5], np.float32)}
    st = osac.adam_init(P, ["w:0"])
    g1, g2 = np.array([0.1, -0.2, 0.0], np.float32), np.array([0.3, 0.1, -0.4], np.float32)
    lr, b1, b2, eps = 3e-4, 0.9, 0.999, 1e-8
    th = P["w:0"].astype(np.float64)
    m = np.zeros(3)
    v = np.zeros(3)
    for t, g in enumerate((g1, g2), 1):
        osac.adam_apply(P, {"w:0": g}, st, lr)
        m = b1 * m + (1 - b1) * g.astype(np.float64)
        v = b2 * v + (1 - b2) * g.astype(np.float64) ** 2
        th = th - lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (np.sqrt(v) + eps)
        assert np.allclose(P["w:0"], th, atol=1e-7, rtol=1e-6), t


def test_keras_adam_differs_only_in_epsilon():
    """Keras 2.2.4 Adam: same recursion with epsilon 1e-7 (K.epsilon()) -- visible when sqrt(v) is tiny."""
    W = {k: v for k, v in oae.init_params(seed=0).items()}
    orc = oae.AeOracle(W, lr=1e-3)
    assert orc.eps == 1e-7
    x = np.zeros((2, 64, 64, 1), np.float32)
    x[:, 20:40, 20:40] = 0.3
    before = {k: v.copy() for k, v in orc.params().items()}
    orc.step(x)
    after = orc.params()
    # first step of Adam moves every coordinate with a non-zero gradient by lr * g / (|g| + eps sqrt(1-b2)/(1-b1)...) ~ lr
    d = np.abs(after["decoder/conv2d_6/bias"] - before["decoder/conv2d_6/bias"])
    assert np.all(d <= 1e-3 * 1.0001) and np.all(d >= 1e-3 * 0.99)


# ----------------------------------------------------------------------------------------------- VecNormalize
def test_running_statistics_merge_is_chans_formula():
    rng = np.random.default_rng(10)
    a, b = rng.normal(1.0, 2.0, size=(40, 3)), rng.normal(-1.0, 0.5, size=(7, 3))
    mean, var, count = a.mean(0), a.var(0), float(len(a))
    m2, v2, c2 = osac.rms_update(mean, var, count, b)
    both = np.concatenate([a, b])
    assert np.allclose(m2, both.mean(0)) and np.allclose(v2, both.var(0)) and c2 == 47.0
    obs = rng.normal(size=(5, 3)) * 50
    z = osac.normalize_obs(obs, m2, v2)
    assert np.array_equal(z, np.clip((obs - m2) / np.sqrt(v2 + 1e-8), -10, 10))


# ----------------------------------------------------------------------------------------------- DQN / BDQ
def test_dqn_target_and_huber_loss_by_hand():
    """Double-DQN target r + gamma (1-d) Q_target(s', argmax_a Q_online(s', a)); Huber delta 1; importance weights."""
    spec = odqn.QSpec(algo="dqn", obs_dim=3, n_branches=1, n_bins=4, branch_hidden=(6,), value_hidden=(6,))
    orc = odqn.QOracle(spec, seed=11)
    rng = np.random.default_rng(12)
    B = 7
    batch = {"obs": torch.from_numpy(rng.normal(size=(B, 3)).astype(np.float32)),
             "next_obs": torch.from_numpy(rng.normal(size=(B, 3)).astype(np.float32)),
             "act": torch.from_numpy(rng.integers(0, 4, size=(B, 1)).astype(np.float32)),
             "rew": torch.from_numpy((rng.normal(size=B) * 3).astype(np.float32)),
             "done": torch.from_numpy((rng.uniform(size=B) < 0.4).astype(np.float32))}
    w = rng.uniform(0.2, 1.0, size=B).astype(np.float32)
    out, _ = orc.grads(batch, w)
    q = orc.q_values(batch["obs"].numpy())[:, 0, :].astype(np.float64)
    qn_online = orc.q_values(batch["next_obs"].numpy())[:, 0, :]
    T = orc.tensors()
    qn_target = odqn.q_forward(spec, T, "%s/target_q_func/model" % spec.scope, batch["next_obs"])[0].numpy()[:, 0, :]
    a = batch["act"].numpy().astype(int)[:, 0]
    y = batch["rew"].numpy() + spec.gamma * (1 - batch["done"].numpy()) * qn_target[np.arange(B), qn_online.argmax(1)]
    td = q[np.arange(B), a] - y
    hub = np.where(np.abs(td) < 1, 0.5 * td ** 2, np.abs(td) - 0.5)
    assert np.allclose(out["td"][:, 0], td, atol=1e-5)
    assert np.isclose(out["loss"], np.mean(w * hub), rtol=1e-5)
    assert np.allclose(out["priority"], np.abs(td), atol=1e-5)


def test_dueling_aggregation_is_value_plus_centred_advantage():
    spec = odqn.bdq_spec(5, 3, 6, [[8], [4], [4]])
    orc = odqn.QOracle(spec, seed=13)
    obs = torch.from_numpy(np.random.default_rng(14).normal(size=(4, 5)).astype(np.float32))
    q, adv, v = odqn.q_forward(spec, orc.tensors(), "%s/model" % spec.scope, obs)
    q, v, adv = q.numpy(), v.numpy(), adv.numpy()
    assert q.shape == (4, 3, 6)
    assert np.allclose(q, v.reshape(4, 1, 1) + adv - adv.mean(axis=2, keepdims=True), atol=1e-6)


# ----------------------------------------------------------------------------------------------- prioritised replay
@pytest.mark.parametrize("stratified", [False, True])
def test_proportional_sampler_against_cumulative_sums(stratified):
    """oracle/per.py restates stable-baselines' segment trees; an implementation that shares nothing with it -- a
    float64 cumulative sum + binary search over the sampled range [0, size - 2] -- must name the same transitions
    for every mass that is not within rounding of an interval boundary, and the closed-form weights must agree."""
    rng = np.random.default_rng(15 + int(stratified))
    n, B = 300, 32
    o = oper.PerOracle(512, alpha=0.6, stratified=stratified)
    o.add(n)
    assert np.array_equal(o.leaves[:n], np.ones(n)) and not o.leaves[n:].any()
    td = rng.uniform(0.01, 5.0, size=n).astype(np.float32)
    o.update(np.arange(n), td)
    leaves = o.leaves[:n].copy()
    want = ((td + np.float32(1e-6)).astype(np.float64) ** float(np.float32(0.6))).astype(np.float32).astype(np.float64)
    assert np.array_equal(leaves, want)                       # float32 power, stored as float64
    cum = np.cumsum(leaves[:n - 1])                            # sum(0, len - 1): the newest transition is left out
    u = rng.random(B)
    mass = o.masses(u)
    assert np.isclose(o._it_sum.sum(0, n - 1), cum[-1], rtol=1e-13)
    if stratified:
        assert np.allclose(mass, (u + np.arange(B)) * cum[-1] / B, rtol=1e-13)
    else:
        assert np.allclose(mass, u * cum[-1], rtol=1e-13)
    idx, w = o.sample(u, beta=0.4)
    ref = np.searchsorted(cum, mass, side="right")
    clear = np.abs(cum[np.minimum(ref, n - 2)] - mass) > 1e-9 * cum[-1]
    assert clear.sum() >= B - 1 and np.array_equal(idx[clear], ref[clear])
    assert idx.max() <= n - 2
    total = leaves.sum()
    assert np.allclose(w, (n * leaves[idx] / total) ** -0.4 / (n * leaves.min() / total) ** -0.4, rtol=1e-12)
    assert float(o._max_priority) == float(np.max(td + np.float32(1e-6)))
    # a repeated index keeps the value of its last occurrence (NumPy fancy assignment), and add() after the update
    # takes the float64 power of the float32 running maximum
    o.update(np.array([5, 7, 5]), np.array([0.5, 0.25, 2.0], np.float32))
    assert o.leaves[5] == np.float64(np.float32(float(np.float32(2.0) + np.float32(1e-6)) ** float(np.float32(0.6))))
    o.add(3)
    assert np.array_equal(o.leaves[n:n + 3], np.full(3, np.float64(o._max_priority) ** 0.6))


def test_segment_tree_prefix_reduce_is_right_nested():
    """`sum(0, end)` of the published SegmentTree adds the left siblings along the path to its last leaf as
    left + (left' + (...)): checked against that closed form written out by hand on a 16-leaf tree."""
    rng = np.random.default_rng(3)
    t = oper.SumSegmentTree(16)
    vals = rng.uniform(0.1, 3.0, 11)
    t[np.arange(11)] = vals
    v = t._value
    # reduce over [0, 9]: node(0..7) + (node(8..9))
    assert t.sum(0, 10) == v[2] + v[6 * 2]
    # reduce over [0, 10]: node(0..7) + (node(8..9) + leaf 10)
    assert t.sum(0, 11) == v[2] + (v[12] + v[16 + 10])
    # reduce over [0, 6]: node(0..3) + (node(4..5) + leaf 6)
    assert t.sum(0, 7) == v[4] + (v[10] + v[16 + 6])
    assert t.sum() == v[1]


def test_relu_hints_move_only_sign_ambiguous_units(monkeypatch):
    """oracle/sac.py RELU_HINTS (used by parity_util.compare_first_step when a gradient deviates): a hint decides the sign of
    a ReLU unit ONLY where the pre-activation is within AMBIG_TOL of zero; everywhere else the oracle keeps its own sign."""
    import parity_util as pu
    from oracle import sac as osac
    case = pu.make_case(extractor="augmented", kind="depth", B=2, n_replay=6, n_steps=1)
    ref, _ = pu.oracle_run(case)
    batch, eps = ref[0]["batch"], case["eps"][0]
    keep = {}
    T = osac.SacOracle(case["spec"], case["params"]).tensors()
    osac.critic_fwd(case["spec"], T, "model/values_fn", batch["obs"], batch["act"], None, keep)
    l1 = keep["l1"].numpy()
    # all-False hints with the real tolerance: nothing is ambiguous in this tiny case -> identical gradients
    hints = {("model/values_fn", 1): np.zeros(l1.shape, bool), ("model/pi", 1): np.zeros(l1.shape, bool)}
    monkeypatch.setattr(osac, "RELU_HINTS", hints)
    osac.RELU_ALIGNED[:] = []
    d1 = osac.SacOracle(case["spec"], case["params"]).step(batch, eps)
    assert osac.RELU_ALIGNED == []
    for n, g in ref[0]["grads"].items():
        assert np.array_equal(g, d1["grads"][n]), n
    # a tolerance wide enough to catch the smallest positive unit: exactly the units below it follow the hint
    pos = l1[l1 > 0]
    tol = float(np.sort(pos)[2])
    monkeypatch.setattr(osac, "AMBIG_TOL", tol)
    osac.RELU_ALIGNED[:] = []
    d2 = osac.SacOracle(case["spec"], case["params"]).step(batch, eps)
    moved = [a for a in osac.RELU_ALIGNED if a[0] == "model/values_fn" and a[1] == 1]
    assert len(moved) == 3 and all(0 < a[3] <= tol for a in moved)
    assert any(not np.array_equal(ref[0]["grads"][n], d2["grads"][n]) for n in d2["grads"] if "values_fn/cnn1" in n or "values_fn/c1" in n)


def test_sign_ambiguity_bounds_are_frozen():
    """the tolerance within which `compare_first_step` may tell the oracle the engine's side of a ReLU, and the
    number of units per case that may need it, are test constants -- widening either must fail here first."""
    import parity_util as pu
    from oracle import sac as osac
    assert osac.AMBIG_TOL == pu.AMBIG_TOL_FROZEN == 2e-8
    assert pu.MAX_HINTED_UNITS == 2
