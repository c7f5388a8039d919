#!/usr/bin/env python
"""Extract small fixtures from /root/reference and pin the oracle against them.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python scripts/make_golden.py

Writes
  tests/golden/sac_mlp_best_model.npz     SAC-MLP parameter set shipped by the reference
                                          (trained_models/SAC_encoder_1mbuffer/best_model/best_model.zip)
  tests/golden/vecnorm_encoder.npz        its VecNormalize statistics + the 2 real 101-d observations
  tests/golden/depth_frames.npz           the six real 64x64 depth frames preserved in vecnormalize.pkl files
  tests/golden/ae_new_gripper_encoder.npz encoder half of encoder_files/new_gripper_encoder/model.h5
  tests/golden/oracle_pins.json           results of the SURVEY.md B.5 known-relationship checks +
                                          oracle outputs on the real inputs (golden vectors for the kernels)
  deep-rl-grasping_amd/grasp_rl/data/obs_stats_{depth,rgbd}.npz   per-pixel observation statistics used
                                          to draw synthetic replay contents
"""
import glob
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import autoencoder as ae            # noqa: E402
from oracle import fixtures as fx               # noqa: E402
from oracle import sac as osac                  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "deep-rl-grasping_amd", "grasp_rl", "data")


def ent_coef_log_pin():
    """B.7: the first rows of trained_models/SAC_depth_1mbuffer/logs.csv (the training log of the reference's own depth-SAC run:
    total_timesteps, ent_coef, current_lr, entropy, ...).  SAC starts updating after learning_starts = 100 env steps with
    one update per step; the entropy coefficient is exp(log_ent_coef), log_ent_coef starts at 0 and takes one Adam step per
    update -- while the gradient keeps its sign (entropy far above the target) every step is ~lr, so the logged value at
    step T is exp(-lr (T - 100)): a relationship that holds only for THIS parametrisation, sign and update schedule."""
    import csv
    rows = []
    with open(REF + "/trained_models/SAC_depth_1mbuffer/logs.csv") as f:
        for k, r in enumerate(csv.DictReader(f)):
            rows.append({"total_timesteps": int(float(r["total_timesteps"])), "ent_coef": float(r["ent_coef"]),
                         "current_lr": float(r["current_lr"]), "entropy": float(r["entropy"]), "ent_coef_loss": float(r["ent_coef_loss"])})
            if k >= 3:
                break
    import yaml
    cfg = yaml.safe_load(open(REF + "/trained_models/SAC_depth_1mbuffer/config.yaml"))
    return {"source": "trained_models/SAC_depth_1mbuffer/logs.csv (first rows) + config.yaml", "rows": rows,
            "learning_rate": float(cfg["SAC"]["learning_rate"]) if "learning_rate" in cfg.get("SAC", {}) else rows[0]["current_lr"],
            "learning_starts": 100, "action_dim": 5}


def entropy_log_pin(n_rows=6):
    """B.8: the first rows of the same log carry THREE columns that only the restated arithmetic ties together: `entropy`
    (SB: mean of the diagonal Gaussian's entropy, sum(log_std + 0.5 log(2 pi e))), `ent_coef` and `ent_coef_loss`
    (-mean(log_ent_coef (logp_pi + target_entropy)) with logp_pi the SQUASHED log-likelihood and target_entropy = -A).  Early
    in training the policy mean is ~0, so the logged entropy fixes log_std, log_std fixes E[logp_pi] through the Gaussian
    likelihood and the tanh correction log(1 - tanh(u)^2 + eps), and the logged coefficient then fixes the logged loss: row 1
    to 0.04 %, the first six rows to < 1 %.  Without the tanh correction the loss is 37 % off, with target_entropy = -A/2
    30 %, with `entropy` = -mean(logp_pi) the log would say 3.4 instead of 6.5."""
    import csv
    rows = []
    with open(REF + "/trained_models/SAC_depth_1mbuffer/logs.csv") as f:
        for k, r in enumerate(csv.DictReader(f)):
            if k >= n_rows:
                break
            rows.append({"total_timesteps": int(float(r["total_timesteps"])), "ent_coef": float(r["ent_coef"]),
                         "entropy": float(r["entropy"]), "ent_coef_loss": float(r["ent_coef_loss"])})
    return {"source": "trained_models/SAC_depth_1mbuffer/logs.csv (first %d rows: entropy, ent_coef, ent_coef_loss)" % n_rows,
            "rows": rows, "action_dim": 5}


def eps_schedule_log_pin():
    """B.9: the training logs of the shipped DQN / BDQ runs carry `time_spent_exploring` = int(100 * exploration.value(t))
    (stable-baselines deepq / the BDQ fork log it per episode batch) next to `total_timesteps`: 1 061 + 25 021 rows, 98 / 91
    distinct values, every one of them equal to int(100 * (1 + min(t / (exploration_fraction * total_timesteps), 1) *
    (exploration_final_eps - 1))) with the fraction / final epsilon / total of the run's config.yaml (DQN: stable-baselines'
    defaults 0.1 / 0.02).  Kept: every row where the logged value changes, the row before it, the first and the last."""
    import csv
    import yaml
    out = {}
    for d, fname, algo in (("DQN_4pads", "logs.csv", "DQN"), ("BDQ_8pads", "logs.full.csv", "BDQ")):
        cfg = yaml.safe_load(open(REF + "/trained_models/%s/config.yaml" % d))[algo]
        lines = [l for l in open(REF + "/trained_models/%s/%s" % (d, fname)) if not l.startswith("#")]
        rows = [(int(float(r["total_timesteps"])), int(r["time_spent_exploring"])) for r in csv.DictReader(lines)]
        keep = sorted({0, len(rows) - 1} | {k for k in range(1, len(rows)) if rows[k][1] != rows[k - 1][1]}
                      | {k - 1 for k in range(1, len(rows)) if rows[k][1] != rows[k - 1][1]})
        out[d] = {"source": "trained_models/%s/%s + config.yaml" % (d, fname), "algo": algo, "n_rows_in_log": len(rows),
                  "total_timesteps": int(float(cfg["total_timesteps"])),
                  "exploration_fraction": cfg.get("exploration_fraction"), "exploration_final_eps": cfg.get("exploration_final_eps"),
                  "rows": [list(rows[k]) for k in keep]}
    return out


def main():
    if "--pins-r6" in sys.argv:      # add the round-6 pins to the existing file, leave every other fixture as it is
        with open(GOLD + "/oracle_pins.json") as f:
            pins = json.load(f)
        pins["b8_entropy_log"] = entropy_log_pin()
        pins["b9_eps_schedule_logs"] = eps_schedule_log_pin()
        with open(GOLD + "/oracle_pins.json", "w") as f:
            json.dump(pins, f, indent=1, sort_keys=True)
        print("updated", GOLD + "/oracle_pins.json")
        return
    os.makedirs(GOLD, exist_ok=True)
    os.makedirs(DATA, exist_ok=True)
    pins = {}

    # ---------------------------------------------------------------- SAC-MLP fixture
    zpath = REF + "/trained_models/SAC_encoder_1mbuffer/best_model/best_model.zip"
    data, params = fx.load_sb_zip(zpath)
    np.savez_compressed(GOLD + "/sac_mlp_best_model.npz", **{k: v for k, v in params.items()})
    pins["sac_mlp_zip"] = {
        "source": zpath[len(REF) + 1:],
        "param_names": list(params.keys()),
        "hyper": {k: data[k] for k in ("gamma", "tau", "batch_size", "buffer_size", "learning_starts",
                                       "train_freq", "ent_coef", "n_envs")},
    }
    vn = fx.load_vecnormalize_pkl(REF + "/trained_models/SAC_encoder_1mbuffer/best_model/vecnormalize.pkl")
    vn2 = fx.load_vecnormalize_pkl(REF + "/trained_models/SAC_encoder_1mbuffer/vecnormalize.pkl")
    real_obs = np.concatenate([vn["old_obs"], vn2["old_obs"]], 0).astype(np.float32)
    np.savez_compressed(GOLD + "/vecnorm_encoder.npz", mean=vn["obs_rms"]["mean"], var=vn["obs_rms"]["var"],
                        count=vn["obs_rms"]["count"], ret_var=vn["ret_rms"]["var"],
                        clip_obs=vn["clip_obs"], clip_reward=vn["clip_reward"], epsilon=vn["epsilon"],
                        gamma=vn["gamma"], real_obs=real_obs)

    # B.5 check 1: V(s) ~= min(Q1,Q2)(s, pi(s)) - alpha*logp for the trained parameter set
    spec = osac.SacSpec(extractor="mlp", obs_dim=101, act_dim=5, layers=[64, 64])
    assert list(osac.param_shapes(spec).keys()) == list(params.keys()), "TF name/order mismatch"
    for k, shp in osac.param_shapes(spec).items():
        assert tuple(params[k].shape) == tuple(shp), (k, params[k].shape, shp)
    rng = np.random.default_rng(0)
    mean, var = vn["obs_rms"]["mean"], vn["obs_rms"]["var"]
    raw = rng.normal(mean, np.sqrt(var), (512, 101))
    obs_n = osac.normalize_obs(raw, mean, var).astype(np.float32)
    orc = osac.SacOracle(spec, params)
    T = orc.tensors()
    eps = rng.standard_normal((512, 5)).astype(np.float32)
    a = osac.actor_fwd(spec, T, torch.from_numpy(obs_n), torch.from_numpy(eps))
    c = osac.critic_fwd(spec, T, "model/values_fn", torch.from_numpy(obs_n), a["pi"], a["pi"])
    t = osac.critic_fwd(spec, T, "target/values_fn", torch.from_numpy(obs_n))
    alpha = float(np.exp(params["model/log_ent_coef:0"]))
    minq = torch.minimum(c["qf1"], c["qf2"]).numpy()
    v = c["v"].numpy()
    corr = float(np.corrcoef(v, minq)[0, 1])
    # reversed concat order must NOT satisfy the relation (discriminating check)
    Tw = dict(T)
    h = torch.from_numpy(obs_n)
    qs = []
    for q in ("qf1", "qf2"):
        qin = torch.cat([a["pi"], h], 1)
        qs.append(osac.dense(Tw, "model/values_fn/" + q, q,
                             osac.mlp_fwd(spec, Tw, "model/values_fn/" + q, qin)).reshape(-1))
    corr_rev = float(np.corrcoef(v, torch.minimum(*qs).numpy())[0, 1])
    corr_tgt = float(np.corrcoef(v, t["v"].numpy())[0, 1])
    pins["b5_sac_head_wiring"] = {"alpha": alpha, "corr_V_minQpi": corr, "corr_reversed_concat": corr_rev,
                                  "corr_V_Vtarget": corr_tgt,
                                  "mean_abs_V_minus_minQ": float(np.abs(v - minq).mean())}
    print("B.5-1", pins["b5_sac_head_wiring"])
    assert corr > 0.9 and corr_rev < 0.5 and corr_tgt > 0.99

    # golden forward vectors on the 2 real observations (normalised with the shipped stats)
    ro = osac.normalize_obs(real_obs, mean, var).astype(np.float32)
    a2 = osac.actor_fwd(spec, T, torch.from_numpy(ro), torch.zeros(2, 5))
    c2 = osac.critic_fwd(spec, T, "model/values_fn", torch.from_numpy(ro), a2["det"], a2["det"])
    pins["sac_mlp_real_obs"] = {"mu": a2["mu"].numpy().tolist(), "log_std": a2["log_std"].numpy().tolist(),
                                "det_action": a2["det"].numpy().tolist(), "v": c2["v"].numpy().tolist(),
                                "qf1_det": c2["qf1"].numpy().tolist(), "qf2_det": c2["qf2"].numpy().tolist()}

    # ---------------------------------------------------------------- observation statistics
    vd = fx.load_vecnormalize_pkl(REF + "/trained_models/SAC_depth_1mbuffer/best_model/vecnormalize.pkl")
    np.savez_compressed(DATA + "/obs_stats_depth.npz", mean=vd["obs_rms"]["mean"], var=vd["obs_rms"]["var"],
                        count=vd["obs_rms"]["count"], ret_var=vd["ret_rms"]["var"])
    vr = fx.load_vecnormalize_pkl(REF + "/trained_models/SAC_full_rgbd/vecnormalize.pkl")
    np.savez_compressed(DATA + "/obs_stats_rgbd.npz", mean=vr["obs_rms"]["mean"], var=vr["obs_rms"]["var"],
                        count=vr["obs_rms"]["count"], ret_var=vr["ret_rms"]["var"])
    pins["obs_stats"] = {"depth_mean_range": [float(vd["obs_rms"]["mean"][..., 0].min()),
                                              float(vd["obs_rms"]["mean"][..., 0].max())],
                         "depth_pad00": [float(vd["obs_rms"]["mean"][0, 0, 1]), float(vd["obs_rms"]["var"][0, 0, 1])],
                         "ret_var": float(vd["ret_rms"]["var"])}

    # ---------------------------------------------------------------- real depth frames
    frames = []
    for p in sorted(glob.glob(REF + "/trained_models/**/vecnormalize.pkl", recursive=True)):
        o = fx.load_vecnormalize_pkl(p)["old_obs"]
        if o.ndim == 4:
            frames.append(o[0, :, :, 0 if o.shape[-1] == 2 else 3].astype(np.float32))
    frames = np.stack(frames)
    np.savez_compressed(GOLD + "/depth_frames.npz", frames=frames)
    print("real depth frames:", frames.shape, frames.min(), frames.max())

    # ---------------------------------------------------------------- auto-encoder (B.5 check 2)
    table = {}
    for tag, p in (("root", REF + "/encoder_files/model.h5"),
                   ("new_gripper_encoder", REF + "/encoder_files/new_gripper_encoder/model.h5"),
                   ("original_encoder", REF + "/encoder_files/original_encoder/model.h5"),
                   ("new_encoder", REF + "/encoder_files/new_encoder/model.h5")):
        W = fx.load_keras_ae_h5(p)
        x = frames[..., None]
        rec = ae.decode(W, ae.encode(W, x))
        mse_ok = float(np.mean((rec - x) ** 2))
        # wrong conventions (symmetric k//2 padding) must be clearly worse
        orig = ae.tf_same_pad
        ae.tf_same_pad = lambda n, k, s: (k // 2, k // 2 - (1 if (n + 2 * (k // 2) - k) % s else 0)) \
            if s > 1 else (k // 2, k // 2)
        try:
            rec_bad = ae.decode(W, ae.encode(W, x))
        finally:
            ae.tf_same_pad = orig
        mse_bad = float(np.mean((rec_bad - x) ** 2))
        table[tag] = {"mse_tf_same_nhwc": mse_ok, "mse_symmetric_pad": mse_bad}
        print("B.5-2", tag, table[tag])
        assert mse_ok < 0.5 * mse_bad and mse_ok < 0.01
        if tag == "new_gripper_encoder":
            enc = {k: v for k, v in W.items() if k.startswith("encoder/")}
            np.savez_compressed(GOLD + "/ae_new_gripper_encoder.npz", **enc)
            z = ae.encode(W, x)
            np.savez_compressed(GOLD + "/ae_encodings.npz", z=z)
            pins["ae_encoding_abs_mean"] = float(np.abs(z).mean())
    pins["b5_autoencoder_mse"] = table

    # ---------------------------------------------------------------- DQN / BDQ zips (B.6: relationships the fork's
    # un-vendored source leaves as the only evidence)
    from oracle import dqn as od
    q = {}
    for tag, rel in (("bdq_33_big_final", "BDQ_33pads_big/BDQ_33_big.zip"), ("bdq_33_big_best", "BDQ_33pads_big/best_model/best_model.zip"),
                     ("bdq_8pads_final", "BDQ_8pads/BDQ_simple_8pads.zip"), ("dqn_4pads_final", "DQN_4pads/DQN_simple_4pads.zip")):
        data, params = fx.load_sb_zip(REF + "/trained_models/" + rel)
        d = max(float(np.abs(params[k] - params[k.replace("/target_q_func", "")]).max()) for k in params if "/target_q_func/" in k)
        eps_name = [k for k in params if k.endswith("eps:0")][0]
        q[tag] = {"source": "trained_models/" + rel, "max_abs_target_minus_online": d,
                  "target_network_update_freq": data["target_network_update_freq"], "learning_rate": data["learning_rate"],
                  "stored_eps": float(params[eps_name]), "exploration_final_eps": data["exploration_final_eps"]}
        # (1) a run's final zip is written right after a target update (total_timesteps is a multiple of the update
        #     period): target == online BIT FOR BIT -- only a hard copy can do that (Polyak averaging never reaches it);
        #     the mid-interval best_model zip differs by at most (steps since the copy) x (Adam step <= lr).
        if tag.endswith("_final"):
            assert d == 0.0, (tag, d)
        else:
            assert 0.0 < d <= data["target_network_update_freq"] * data["learning_rate"], (tag, d)
        # (2) the `eps` variable holds the value of the exploration schedule at save time (= its final value)
        assert abs(float(params[eps_name]) - data["exploration_final_eps"]) < 1e-7, tag
    pins["b6_q_zip_relationships"] = q
    print("B.6", json.dumps(q, indent=1))
    # golden vectors: the shipped big BDQ network (5 branches x 33 bins, [[512,256],[128],[128]]-style towers as stored)
    # on the two real 101-d observations of the SAC-encoder run (the only real feature vectors the reference ships)
    data, params = fx.load_sb_zip(REF + "/trained_models/BDQ_33pads_big/best_model/best_model.zip")
    shapes = {k: tuple(v.shape) for k, v in params.items()}
    common = [shapes["bdq/model/common_net/fully_connected%s/weights:0" % ("" if i == 0 else "_%d" % i)][1]
              for i in range(8) if "bdq/model/common_net/fully_connected%s/weights:0" % ("" if i == 0 else "_%d" % i) in shapes]
    n_av = len([k for k in shapes if k.startswith("bdq/model/action_value/") and k.endswith("weights:0")])
    bins = shapes["bdq/model/action_value/fully_connected_1/weights:0"][1]
    branch_h = shapes["bdq/model/action_value/fully_connected/weights:0"][1]
    value_h = shapes["bdq/model/state_value/fully_connected/weights:0"][1]
    D = n_av // 2
    obs_dim = shapes["bdq/model/common_net/fully_connected/weights:0"][0]
    spec_q = od.bdq_spec(obs_dim, D, bins, [common, [branch_h], [value_h]])
    assert list(od.param_shapes(spec_q).keys()) == list(params.keys())
    np.savez_compressed(GOLD + "/bdq_33_big_best_model.npz", **{k: v for k, v in params.items()})
    orc_q = od.QOracle(spec_q, params)
    obs_q = real_obs[:, :obs_dim].astype(np.float32)
    qv = orc_q.q_values(obs_q)
    pins["bdq_real_obs"] = {"spec": {"obs_dim": obs_dim, "branches": D, "bins": bins, "common": common, "branch": branch_h, "value": value_h},
                            "q_values": np.asarray(qv).tolist(), "greedy_bins": np.asarray(qv).argmax(axis=2).tolist()}
    # dueling aggregation per branch (Q_d = V + A_d - mean_a A_d): the branch means of Q all equal V, i.e. each other
    bm = np.asarray(qv).mean(axis=2)
    assert np.allclose(bm, bm[:, :1], atol=1e-4), bm
    pins["bdq_real_obs"]["branch_means_of_Q"] = bm.tolist()

    pins["b7_ent_coef_log"] = ent_coef_log_pin()
    pins["b8_entropy_log"] = entropy_log_pin()
    pins["b9_eps_schedule_logs"] = eps_schedule_log_pin()

    with open(GOLD + "/oracle_pins.json", "w") as f:
        json.dump(pins, f, indent=1, sort_keys=True)
    print("wrote", GOLD)


if __name__ == "__main__":
    main()
