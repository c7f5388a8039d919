"""CPU restatement of the DQN and BDQ updates.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

DQN: stock stable-baselines 2.10.1 ``deepq`` as constructed at
/root/reference/manipulation_main/training/sb_helper.py:159-165 (``DQNMlpPolicy``, dueling, double-Q,
Huber loss, per-variable clip_by_norm(10), Adam 5e-4, hard target copy).  Topology and TF variable
names are pinned by the shipped ``trained_models/DQN_4pads/*.zip`` (tests/golden/q_zip_meta.json).

BDQ: the ``bdq_sb`` fork (``.gitmodules:1-3``) is NOT vendored and has no pinned commit; its source is
unavailable.  What follows is restated from the Action-Branching paper (Tavakoli et al. 2018,
acknowledged at README.md:135) constrained by the topology / hyper-parameter names of the shipped
``trained_models/BDQ_*/*.zip``: shared ``common_net`` -> per-dimension ``action_value`` branches +
``state_value``; dueling aggregation per branch with the mean advantage; double-Q argmax per branch;
target averaged over branches; mean squared TD over branches; trunk gradient rescaled by 1/(D+1);
priorities = sum_d |td_d|.  EVERY ONE OF THESE IS A DOCUMENTED DECISION, NOT A CITATION:
**BDQ parity is unpinned** (SURVEY.md 8c).
"""
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch
import torch.nn.functional as F

from . import sac as osac


@dataclass
class QSpec:
    algo: str = "dqn"                 # 'dqn' | 'bdq'
    obs_dim: int = 100
    n_branches: int = 1               # D: 1 for DQN, action dims for BDQ
    n_bins: int = 12                  # actions per branch (num_actions_pad)
    common: List[int] = field(default_factory=list)                 # BDQ layers[0]
    branch_hidden: List[int] = field(default_factory=lambda: [64, 64])   # DQN tower / BDQ layers[1]
    value_hidden: List[int] = field(default_factory=lambda: [64, 64])    # DQN tower / BDQ layers[2]
    gamma: float = 0.99
    lr: float = 5e-4
    grad_clip: float = 10.0
    double_q: bool = True
    # BDQ: the two aggregation choices with a defensible alternative reading of the (unavailable) fork -- switches, not
    # citations.  Nothing the reference ships discriminates them: trained_models/BDQ_8pads/logs.full.csv logs mean_loss
    # next to mean_td_errors over 25 022 rows, but their ratio loss / td^2 ranges from 0.0017 to 1.8 (importance
    # weights, heavy-tailed TD errors and unknown logging windows), compatible with either aggregation; the zips store
    # hyper-parameters only (note prioritized_replay_eps = 1e8 there: priorities are effectively uniform).
    loss_sum_branches: bool = False   # True: sum_d td_d^2 instead of mean_d (gradients D times larger before clipping)
    trunk_rescale: bool = True        # False: no 1/(D+1) scaling of the gradient entering the shared trunk

    @property
    def scope(self):
        return "deepq" if self.algo == "dqn" else "bdq"

    @property
    def huber(self):
        return self.algo == "dqn"

    @property
    def trunk_scale(self):
        return 1.0 / (self.n_branches + 1) if (self.algo == "bdq" and self.common and self.trunk_rescale) else 1.0


def bdq_spec(obs_dim, act_dim, num_actions_pad, layers, **kw):
    """``policy_kwargs={'layers': [[common...], [branch...], [value...]]}`` (sb_helper.py:213, zips)."""
    return QSpec(algo="bdq", obs_dim=obs_dim, n_branches=act_dim, n_bins=num_actions_pad, common=list(layers[0]),
                 branch_hidden=list(layers[1]), value_hidden=list(layers[2]), **kw)


def _fc(k):
    return "fully_connected" if k == 0 else "fully_connected_%d" % k


def net_shapes(spec, prefix):
    """tf.contrib.layers.fully_connected variables in creation order under one q_func scope."""
    out = OrderedDict()
    d = spec.obs_dim
    if spec.common:
        for k, h in enumerate(spec.common):
            out["%s/common_net/%s/weights:0" % (prefix, _fc(k))] = (d, h)
            out["%s/common_net/%s/biases:0" % (prefix, _fc(k))] = (h,)
            d = h
    k = 0
    for _ in range(spec.n_branches):
        dd = d
        for h in spec.branch_hidden:
            out["%s/action_value/%s/weights:0" % (prefix, _fc(k))] = (dd, h)
            out["%s/action_value/%s/biases:0" % (prefix, _fc(k))] = (h,)
            dd = h
            k += 1
        out["%s/action_value/%s/weights:0" % (prefix, _fc(k))] = (dd, spec.n_bins)
        out["%s/action_value/%s/biases:0" % (prefix, _fc(k))] = (spec.n_bins,)
        k += 1
    dd, k = d, 0
    for h in spec.value_hidden:
        out["%s/state_value/%s/weights:0" % (prefix, _fc(k))] = (dd, h)
        out["%s/state_value/%s/biases:0" % (prefix, _fc(k))] = (h,)
        dd = h
        k += 1
    out["%s/state_value/%s/weights:0" % (prefix, _fc(k))] = (dd, 1)
    out["%s/state_value/%s/biases:0" % (prefix, _fc(k))] = (1,)
    return out


def param_shapes(spec):
    out = OrderedDict()
    out["%s/eps:0" % spec.scope] = ()
    out.update(net_shapes(spec, "%s/model" % spec.scope))
    out.update(net_shapes(spec, "%s/target_q_func/model" % spec.scope))
    return out


def init_params(spec, seed=0):
    """tf.contrib fully_connected default: Xavier-uniform weights, zero biases; target = copy."""
    rng = np.random.default_rng(seed)
    P = OrderedDict()
    for name, shp in param_shapes(spec).items():
        if "/target_q_func/" in name:
            P[name] = P[name.replace("/target_q_func", "")].copy()
        elif name.endswith("eps:0"):
            P[name] = np.float32(0.0).reshape(())
        elif name.endswith("weights:0"):
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            P[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
        else:
            P[name] = np.zeros(shp, np.float32)
    return P


def q_forward(spec, T, prefix, obs, trunk_scale=1.0):
    """Returns q [B, D, n] (dueling-aggregated), adv [B, D, n], v [B]."""
    def lin(scope, k, x):
        return x @ T["%s/%s/%s/weights:0" % (prefix, scope, _fc(k))] + T["%s/%s/%s/biases:0" % (prefix, scope, _fc(k))]
    h = obs
    for k in range(len(spec.common)):
        h = F.relu(lin("common_net", k, h))
    if trunk_scale != 1.0:      # gradient entering the shared trunk rescaled, forward value unchanged
        h = h * trunk_scale + (h * (1.0 - trunk_scale)).detach()
    advs, k = [], 0
    for _ in range(spec.n_branches):
        z = h
        for _h in spec.branch_hidden:
            z = F.relu(lin("action_value", k, z))
            k += 1
        advs.append(lin("action_value", k, z))
        k += 1
    adv = torch.stack(advs, dim=1)
    z, k = h, 0
    for _h in spec.value_hidden:
        z = F.relu(lin("state_value", k, z))
        k += 1
    v = lin("state_value", k, z).reshape(-1)
    q = v[:, None, None] + adv - adv.mean(dim=2, keepdim=True)
    return q, adv, v


class QOracle:
    def __init__(self, spec, params=None, seed=0):
        self.spec = spec
        self.P = OrderedDict((k, np.array(v, np.float32)) for k, v in
                             (params if params is not None else init_params(spec, seed)).items())
        self.train_names = [n for n in self.P if "/target_q_func/" not in n and not n.endswith("eps:0")]
        self.opt = osac.adam_init(self.P, self.train_names)

    def tensors(self, grad=False):
        T = OrderedDict()
        for k, v in self.P.items():
            t = torch.from_numpy(np.array(v, np.float32))
            if grad and k in self.train_names:
                t.requires_grad_(True)
            T[k] = t
        return T

    def q_values(self, obs):
        q, _, _ = q_forward(self.spec, self.tensors(), "%s/model" % self.spec.scope,
                            torch.as_tensor(np.asarray(obs, np.float32)))
        return q.numpy()

    def grads(self, batch, weights=None):
        """batch: obs, next_obs [B,obs_dim]; act [B,D] integer bins; rew, done [B] (torch float32)."""
        spec = self.spec
        T = self.tensors(grad=True)
        B = batch["obs"].shape[0]
        w = torch.ones(B) if weights is None else torch.as_tensor(np.asarray(weights, np.float32))
        pre = "%s/model" % spec.scope
        q, _, _ = q_forward(spec, T, pre, batch["obs"], spec.trunk_scale)
        a = batch["act"].long()
        q_sel = torch.gather(q, 2, a[:, :, None]).squeeze(2)                    # [B, D]
        with torch.no_grad():
            q_tp1_online, _, _ = q_forward(spec, T, pre, batch["next_obs"])
            q_tp1_tgt, _, _ = q_forward(spec, T, "%s/target_q_func/model" % spec.scope, batch["next_obs"])
            sel = (q_tp1_online if spec.double_q else q_tp1_tgt).argmax(dim=2)    # [B, D]
            q_best = torch.gather(q_tp1_tgt, 2, sel[:, :, None]).squeeze(2).mean(dim=1)
            y = batch["rew"] + spec.gamma * (1.0 - batch["done"]) * q_best
        td = q_sel - y[:, None]                                                 # [B, D]
        if spec.huber:
            err = torch.where(td.abs() < 1.0, 0.5 * td ** 2, td.abs() - 0.5)
        else:
            err = td ** 2
        loss = torch.mean(w * (err.sum(dim=1) if (spec.algo == "bdq" and spec.loss_sum_branches) else err.mean(dim=1)))
        gs = torch.autograd.grad(loss, [T[n] for n in self.train_names])
        G = {n: g.numpy().copy() for n, g in zip(self.train_names, gs)}
        return {"loss": float(loss.detach()), "td": td.detach().numpy(), "q_sel": q_sel.detach().numpy(),
                "y": y.numpy(), "q": q.detach().numpy(), "priority": td.detach().abs().sum(dim=1).numpy()}, G

    def step(self, batch, weights=None):
        out, G = self.grads(batch, weights)
        clip = np.float32(self.spec.grad_clip)
        Gc = {}
        for n, g in G.items():                          # tf.clip_by_norm per variable
            norm = np.sqrt(np.sum(g.astype(np.float32) ** 2, dtype=np.float32))
            Gc[n] = (g * clip / max(norm, clip)).astype(np.float32)
        osac.adam_apply(self.P, Gc, self.opt, self.spec.lr)
        out["grads"], out["clipped"] = G, Gc
        return out

    def update_target(self):
        for n in self.P:
            if "/target_q_func/" in n:
                self.P[n] = self.P[n.replace("/target_q_func", "")].copy()
