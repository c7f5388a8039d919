"""CPU restatement (PyTorch-CPU fp32 + NumPy f64) of the SAC update the reference trains with.

TEST INFRASTRUCTURE -- see oracle/__init__.py for who may import this.

The reference never implements this arithmetic itself; it calls
``stable_baselines.SAC`` (``/root/reference/manipulation_main/training/sb_helper.py:104-128``)
with the policy selected at ``sb_helper.py:85-96`` and the feature extractor of
``custom_obs_policy.py:15-43``.  stable-baselines 2.10.1 / TensorFlow 1.14 are absent, so
every function below restates their published algorithm and cites
the reference line that selects it.  Backward passes use torch autograd, which makes the
oracle independent of the hand-derived gradients in the HIP kernels.

PARITY UNPINNED by reference tests; pinned by fixture relationships only
(tests/test_oracle_golden.py, scripts/make_golden.py).
"""
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-6            # stable_baselines.sac.policies.EPS
LOG_STD_MAX = 2.0     # stable_baselines.sac.policies.LOG_STD_MAX
LOG_STD_MIN = -20.0   # stable_baselines.sac.policies.LOG_STD_MIN


@dataclass
class SacSpec:
    """What ``SBPolicy.learn`` (sb_helper.py:85-96) selects.

    extractor:
      'augmented' -- sacCnn + custom_obs_policy.create_augmented_nature_cnn(n_direct)
                     (sb_helper.py:86-90): last obs channel carries the direct features.
      'nature'    -- sacCnn with the default nature_cnn over all channels (sb_helper.py:91-93).
      'mlp'       -- sacMlp on vector observations (sb_helper.py:94-96); no /255 scaling.
    """
    extractor: str = "augmented"
    img_hw: int = 64                 # config/camera_info.yaml:1-2
    img_channels: int = 1            # channels fed to the CNN (depth: 1, RGB-D: 4; robot.py:223-228)
    n_direct: int = 1                # create_augmented_nature_cnn(1), sb_helper.py:89
    obs_dim: int = 101               # only for extractor == 'mlp'
    act_dim: int = 5                 # actuator.py:72-73
    layers: List[int] = field(default_factory=lambda: [64, 64])   # gripper_grasp.yaml:81
    gamma: float = 0.99              # gripper_grasp.yaml:73
    lr: float = 3e-4                 # gripper_grasp.yaml:83
    tau: float = 0.005               # SB default, confirmed in the shipped zip JSON
    clip_obs: float = 10.0           # sb_helper.py:118-119
    clip_reward: float = 10.0        # VecNormalize default
    norm_eps: float = 1e-8           # VecNormalize default (pickles)

    @property
    def obs_channels(self):
        return self.img_channels + (1 if self.extractor == "augmented" else 0)

    @property
    def feat_dim(self):
        if self.extractor == "mlp":
            return self.obs_dim
        return 512 + (self.n_direct if self.extractor == "augmented" else 0)

    @property
    def target_entropy(self):
        return -float(self.act_dim)   # SB: target_entropy='auto' -> -prod(action_space.shape)


# ----------------------------------------------------------------------------- names
def cnn_names(spec):
    """TF variable names created by the extractor under one scope (Appendix A.2)."""
    if spec.extractor == "augmented":      # custom_obs_policy.py:34-40
        return ["cnn1", "cnn2", "cnn3", "cnn_fc1"]
    if spec.extractor == "nature":         # stable_baselines.common.policies.nature_cnn
        return ["c1", "c2", "c3", "fc1"]
    return []


def param_shapes(spec):
    """OrderedDict TF-name -> shape, in TF creation order (matches the shipped zips, B.1)."""
    out = OrderedDict()
    C = spec.img_channels if spec.extractor == "augmented" else spec.obs_channels
    conv_shapes = [(8, 8, C, 32), (4, 4, 32, 64), (3, 3, 64, 64)]

    def extractor(scope):
        names = cnn_names(spec)
        if not names:
            return
        for n, s in zip(names[:3], conv_shapes):
            out["%s/%s/w:0" % (scope, n)] = s
            out["%s/%s/b:0" % (scope, n)] = (1, s[3], 1, 1)
        out["%s/%s/w:0" % (scope, names[3])] = (1024, 512)
        out["%s/%s/b:0" % (scope, names[3])] = (512,)

    def mlp(scope, in_dim, out_name, out_dim):
        d = in_dim
        for i, h in enumerate(spec.layers):
            out["%s/fc%d/kernel:0" % (scope, i)] = (d, h)
            out["%s/fc%d/bias:0" % (scope, i)] = (h,)
            d = h
        if isinstance(out_name, str):
            out_name = [out_name]
        for on in out_name:
            out["%s/%s/kernel:0" % (scope, on)] = (d, out_dim)
            out["%s/%s/bias:0" % (scope, on)] = (out_dim,)

    F_, A = spec.feat_dim, spec.act_dim
    extractor("model/pi")
    mlp("model/pi", F_, ["dense", "dense_1"], A)
    extractor("model/values_fn")
    mlp("model/values_fn/vf", F_, "vf", 1)
    mlp("model/values_fn/qf1", F_ + A, "qf1", 1)
    mlp("model/values_fn/qf2", F_ + A, "qf2", 1)
    out["model/log_ent_coef:0"] = ()
    extractor("target/values_fn")
    mlp("target/values_fn/vf", F_, "vf", 1)
    return out


def ortho_init(shape, scale, rng):
    """stable_baselines.a2c.utils.ortho_init (Appendix A.2)."""
    shape = tuple(shape)
    flat = shape if len(shape) == 2 else (int(np.prod(shape[:-1])), shape[-1])
    a = rng.normal(0.0, 1.0, flat)
    u, _, v = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == flat else v
    q = q.reshape(shape)
    return (scale * q[:shape[0], :shape[1]]).astype(np.float32)


def init_params(spec, seed=0):
    """Seeded init: orthogonal(sqrt 2) conv/cnn_fc1, Glorot-uniform dense heads, zero bias."""
    rng = np.random.default_rng(seed)
    P = OrderedDict()
    shapes = param_shapes(spec)
    for name, shp in shapes.items():
        if name.startswith("target/"):
            continue
        if name.endswith("/w:0"):
            P[name] = ortho_init(shp, np.sqrt(2.0), rng)
        elif name.endswith("/kernel:0"):
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            P[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
        elif name == "model/log_ent_coef:0":
            P[name] = np.float32(0.0).reshape(())     # log(1.0)
        else:
            P[name] = np.zeros(shp, np.float32)
    for name in shapes:                               # SB: target_init_op copies source
        if name.startswith("target/"):
            P[name] = P["model" + name[len("target"):]].copy()
    return OrderedDict((n, P[n]) for n in shapes)


# ----------------------------------------------------------------------------- A.1
def normalize_obs(obs, mean, var, clip=10.0, eps=1e-8):
    """VecNormalize.normalize_obs at sample time (float64, A.1 step 3)."""
    z = (np.asarray(obs, np.float64) - mean) / np.sqrt(var + eps)
    return np.clip(z, -clip, clip)


def normalize_reward(r, ret_var, clip=10.0, eps=1e-8):
    return np.clip(np.asarray(r, np.float64) / np.sqrt(ret_var + eps), -clip, clip)


def rms_update(mean, var, count, x):
    """stable-baselines 2.10.1 RunningMeanStd.update + update_from_moments (A.1 step 2): batch moments with
    np.mean / np.var IN THE DTYPE OF x (float32 for the observations DummyVecEnv hands to VecNormalize, float64 for
    the returns), float64 parallel (Chan et al.) merge -- expression by expression as published."""
    x = np.asarray(x)
    batch_mean, batch_var, batch_count = np.mean(x, axis=0), np.var(x, axis=0), x.shape[0]
    delta = batch_mean - mean
    tot_count = count + batch_count
    new_mean = mean + delta * batch_count / tot_count
    m_a = var * count
    m_b = batch_var * batch_count
    m_2 = m_a + m_b + np.square(delta) * count * batch_count / (count + batch_count)
    new_var = m_2 / (count + batch_count)
    return new_mean, new_var, tot_count


def prepare_batch(spec, raw, stats, norm_obs=True, norm_reward=True):
    """raw: dict obs[B,H,W,Cobs]|[B,D], act, rew, next_obs, done.  stats: obs mean/var, ret_var
    (or None for ``normalize: False`` configs); norm_obs / norm_reward are the VecNormalize flags of the same
    names (both on in the reference, sb_helper.py:117-119).  Returns float32 torch tensors as fed to TF."""
    obs, nxt, rew = raw["obs"], raw["next_obs"], raw["rew"]
    if stats is not None and norm_obs:
        obs = normalize_obs(obs, stats["mean"], stats["var"], spec.clip_obs, spec.norm_eps)
        nxt = normalize_obs(nxt, stats["mean"], stats["var"], spec.clip_obs, spec.norm_eps)
    if stats is not None and norm_reward:
        rew = normalize_reward(rew, stats["ret_var"], spec.clip_reward, spec.norm_eps)
    obs = torch.from_numpy(np.asarray(obs, np.float32))
    nxt = torch.from_numpy(np.asarray(nxt, np.float32))
    if spec.extractor != "mlp":          # observation_input(scale=True): Box(0,255) -> /255
        obs = obs / 255.0
        nxt = nxt / 255.0
    return {"obs": obs, "next_obs": nxt,
            "act": torch.from_numpy(np.asarray(raw["act"], np.float32)),
            "rew": torch.from_numpy(np.asarray(rew, np.float32)).reshape(-1),
            "done": torch.from_numpy(np.asarray(raw["done"], np.float32)).reshape(-1)}


# ----------------------------------------------------------------------------- A.2
def _conv_nhwc(x, w, b, stride):
    """SB tf_layers.conv: NHWC input, HWIO kernel, VALID, cross-correlation, + bias."""
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=stride)
    return y.permute(0, 2, 3, 1) + b.reshape(1, 1, 1, -1)


# ReLU units whose pre-activation lies within rounding of zero have no defined derivative in float32: two fp32-faithful
# summation orders of the same convolution disagree about the sign of a 1e-10 where the activations are 1e-2 (measured:
# scripts/conv_stack_flips.py).  A comparison may therefore tell the oracle which side the OTHER implementation took for such
# units -- RELU_HINTS[(scope, layer)] = boolean array "output > 0" -- and ONLY units with |pre-activation| <= AMBIG_TOL follow it.
RELU_HINTS = None
AMBIG_TOL = 2e-8
RELU_ALIGNED = []        # (scope, layer, index, pre-activation) of every unit a hint decided, for the test's report


def _relu(z, scope, layer):
    h = RELU_HINTS.get((scope, layer)) if RELU_HINTS else None
    if h is None:
        return F.relu(z)
    zd = z.detach()
    amb = zd.abs() <= AMBIG_TOL
    hint = torch.as_tensor(np.asarray(h, bool)).reshape(zd.shape)
    for i in torch.nonzero(amb & (hint != (zd > 0))):
        RELU_ALIGNED.append((scope, layer, tuple(int(v) for v in i), float(zd[tuple(i)])))
    return z * torch.where(amb, hint, zd > 0).to(z.dtype)


def extractor_fwd(spec, P, scope, x, keep=None):
    """custom_obs_policy.py:15-43 ('augmented') / nature_cnn ('nature') / identity ('mlp')."""
    if spec.extractor == "mlp":
        return x
    B = x.shape[0]
    n1, n2, n3, nf = cnn_names(spec)
    if spec.extractor == "augmented":
        direct = x[..., -1].reshape(B, -1)[:, :spec.n_direct]      # :28-30
        img = x[..., :-1]                                          # :32
    else:
        direct, img = None, x
    g = lambda n, s: P["%s/%s/%s:0" % (scope, n, s)]
    l1 = _relu(_conv_nhwc(img, g(n1, "w"), g(n1, "b"), 4), scope, 1)     # :34
    l2 = _relu(_conv_nhwc(l1, g(n2, "w"), g(n2, "b"), 2), scope, 2)      # :35
    l3 = _relu(_conv_nhwc(l2, g(n3, "w"), g(n3, "b"), 1), scope, 3)      # :36
    flat = l3.reshape(B, -1)                                       # conv_to_fc, NHWC order :37
    h = F.relu(flat @ g(nf, "w") + g(nf, "b"))                     # :40
    if keep is not None:
        keep.update({"l1": l1, "l2": l2, "l3": l3, "fc": h})
    if direct is not None:
        h = torch.cat([h, direct], dim=1)                          # :41
    return h


def mlp_fwd(spec, P, scope, x):
    for i in range(len(spec.layers)):
        x = F.relu(x @ P["%s/fc%d/kernel:0" % (scope, i)] + P["%s/fc%d/bias:0" % (scope, i)])
    return x


def dense(P, scope, name, x):
    return x @ P["%s/%s/kernel:0" % (scope, name)] + P["%s/%s/bias:0" % (scope, name)]


# ----------------------------------------------------------------------------- A.3
def actor_fwd(spec, P, obs, eps):
    """SB sac.policies.FeedForwardPolicy.make_actor (A.3)."""
    h = extractor_fwd(spec, P, "model/pi", obs)
    z = mlp_fwd(spec, P, "model/pi", h)
    mu = dense(P, "model/pi", "dense", z)
    log_std = torch.clamp(dense(P, "model/pi", "dense_1", z), LOG_STD_MIN, LOG_STD_MAX)
    std = torch.exp(log_std)
    u = mu + eps * std
    logp = (-0.5 * (((u - mu) / (std + EPS)) ** 2 + 2 * log_std + np.log(2 * np.pi))).sum(1)
    entropy = (log_std + 0.5 * np.log(2.0 * np.pi * np.e)).sum(1)
    det = torch.tanh(mu)
    pi = torch.tanh(u)
    logp = logp - torch.log(1 - pi ** 2 + EPS).sum(1)
    return {"h_pi": h, "mu": mu, "log_std": log_std, "pi": pi, "det": det, "logp": logp,
            "entropy": entropy}


def critic_fwd(spec, P, scope, obs, action=None, pi_action=None, keep=None):
    """SB make_critics: shared extractor; vf on h; qf1/qf2 on concat(h, action) (A.3)."""
    h = extractor_fwd(spec, P, scope, obs, keep)
    out = {"h": h}
    out["v"] = dense(P, scope + "/vf", "vf", mlp_fwd(spec, P, scope + "/vf", h)).reshape(-1)
    for tag, a in (("", action), ("_pi", pi_action)):
        if a is None:
            continue
        qin = torch.cat([h, a], dim=1)
        for q in ("qf1", "qf2"):
            out[q + tag] = dense(P, "%s/%s" % (scope, q), q,
                                 mlp_fwd(spec, P, "%s/%s" % (scope, q), qin)).reshape(-1)
    return out


# ----------------------------------------------------------------------------- A.5
def adam_init(P, names):
    return {"m": {n: np.zeros_like(P[n]) for n in names},
            "v": {n: np.zeros_like(P[n]) for n in names},
            "beta1_power": np.float32(0.9), "beta2_power": np.float32(0.999)}


def adam_apply(P, grads, st, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """TF 1.x ApplyAdam (training_ops.cc), float32 arithmetic throughout (A.5)."""
    f = np.float32
    alpha = f(lr) * np.sqrt(f(1) - st["beta2_power"], dtype=np.float32) / (f(1) - st["beta1_power"])
    for n, g in grads.items():
        g = g.astype(np.float32)
        st["m"][n] = st["m"][n] + (g - st["m"][n]) * (f(1) - f(beta1))
        st["v"][n] = st["v"][n] + (g * g - st["v"][n]) * (f(1) - f(beta2))
        P[n] = (P[n] - (st["m"][n] * alpha) / (np.sqrt(st["v"][n]) + f(eps))).astype(np.float32)
    st["beta1_power"] = f(st["beta1_power"] * f(beta1))
    st["beta2_power"] = f(st["beta2_power"] * f(beta2))


def trainable_groups(spec):
    names = list(param_shapes(spec).keys())
    pi = [n for n in names if n.startswith("model/pi/")]
    vf = [n for n in names if n.startswith("model/values_fn/")]
    return pi, vf, ["model/log_ent_coef:0"]


def polyak_pairs(spec):
    """zip(get_vars('target/values_fn'), get_vars('model/values_fn')) -- extractor + vf (A.4)."""
    names = list(param_shapes(spec).keys())
    tgt = [n for n in names if n.startswith("target/values_fn/")]
    return [(t, "model" + t[len("target"):]) for t in tgt]


class SacOracle:
    """Holds parameters + the three Adam states; ``step`` = SB ``SAC._train_step`` followed by
    ``target_update_op`` (A.4), on an already prepared batch."""

    def __init__(self, spec, params=None, seed=0):
        self.spec = spec
        self.P = OrderedDict((k, np.array(v, np.float32)) for k, v in
                             (params if params is not None else init_params(spec, seed)).items())
        g_pi, g_vf, g_ent = trainable_groups(spec)
        self.groups = (g_pi, g_vf, g_ent)
        self.opt = [adam_init(self.P, g) for g in self.groups]

    def tensors(self, requires_grad=False):
        T = OrderedDict()
        for k, v in self.P.items():
            t = torch.from_numpy(np.array(v, np.float32))
            if requires_grad and not k.startswith("target/"):
                t.requires_grad_(True)
            T[k] = t
        return T

    def forward(self, batch, eps, T=None):
        spec = self.spec
        T = T if T is not None else self.tensors()
        eps = torch.as_tensor(np.asarray(eps, np.float32))
        a = actor_fwd(spec, T, batch["obs"], eps)
        keep = {}
        c = critic_fwd(spec, T, "model/values_fn", batch["obs"], batch["act"], a["pi"], keep)
        t = critic_fwd(spec, T, "target/values_fn", batch["next_obs"])
        log_ent = T["model/log_ent_coef:0"]
        ent_coef = torch.exp(log_ent)
        q_backup = (batch["rew"] + (1 - batch["done"]) * spec.gamma * t["v"]).detach()
        qf1_loss = 0.5 * torch.mean((q_backup - c["qf1"]) ** 2)
        qf2_loss = 0.5 * torch.mean((q_backup - c["qf2"]) ** 2)
        ent_loss = -torch.mean(log_ent * (a["logp"] + spec.target_entropy).detach())
        policy_loss = torch.mean(ent_coef * a["logp"] - c["qf1_pi"])
        min_q = torch.minimum(c["qf1_pi"], c["qf2_pi"])
        v_backup = (min_q - ent_coef * a["logp"]).detach()
        value_loss = 0.5 * torch.mean((c["v"] - v_backup) ** 2)
        values_loss = qf1_loss + qf2_loss + value_loss
        out = dict(a)
        out.update({"h_c": c["h"], "h_tgt": t["h"], "v": c["v"], "v_tgt": t["v"],
                    "qf1": c["qf1"], "qf2": c["qf2"], "qf1_pi": c["qf1_pi"],
                    "qf2_pi": c["qf2_pi"], "q_backup": q_backup, "v_backup": v_backup,
                    "policy_loss": policy_loss, "qf1_loss": qf1_loss, "qf2_loss": qf2_loss,
                    "value_loss": value_loss, "values_loss": values_loss, "ent_loss": ent_loss,
                    "ent_coef": ent_coef, "cnn_c": keep})
        return out

    def grads(self, batch, eps):
        """One forward, three gradient sets at the pre-update parameters (A.4 ordering note)."""
        T = self.tensors(requires_grad=True)
        out = self.forward(batch, eps, T)
        g_pi, g_vf, g_ent = self.groups
        G = {}
        for loss, names in ((out["policy_loss"], g_pi), (out["values_loss"], g_vf),
                            (out["ent_loss"], g_ent)):
            gs = torch.autograd.grad(loss, [T[n] for n in names], retain_graph=True,
                                     allow_unused=True)
            for n, g in zip(names, gs):
                G[n] = (torch.zeros_like(T[n]) if g is None else g).numpy().copy()
        return out, G

    def step(self, batch, eps):
        out, G = self.grads(batch, eps)
        for names, st in zip(self.groups, self.opt):       # pi -> values -> alpha
            adam_apply(self.P, {n: G[n] for n in names}, st, self.spec.lr)
        tau = np.float32(self.spec.tau)
        for t, s in polyak_pairs(self.spec):               # target_update_op
            self.P[t] = ((np.float32(1) - tau) * self.P[t] + tau * self.P[s]).astype(np.float32)
        diag = {k: (v.detach().numpy().copy() if torch.is_tensor(v) else v)
                for k, v in out.items() if k != "cnn_c"}
        diag["grads"] = G
        return diag

    def act(self, obs, deterministic=True, eps=None):
        """SB ``policy_tf.step`` (A.3 / call site utils.py:71): obs already normalised."""
        T = self.tensors()
        obs = torch.as_tensor(np.asarray(obs, np.float32))
        if self.spec.extractor != "mlp":
            obs = obs / 255.0
        if eps is None:
            eps = np.zeros((obs.shape[0], self.spec.act_dim), np.float32)
        a = actor_fwd(self.spec, T, obs, torch.as_tensor(np.asarray(eps, np.float32)))
        return (a["det"] if deterministic else a["pi"]).numpy()
