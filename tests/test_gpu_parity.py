"""GPU parity tests proper: the HIP path, called through the C ABI of libgrl.so, against the CPU
oracle on identical seeded minibatches (tolerances: tests/parity_util.py)."""
import json
import os

import numpy as np
import pytest

import parity_util as pu
from grasp_rl import _capi, synthetic
from grasp_rl.engine import SacEngine
from oracle import sac as osac

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    "depth_augmented": dict(extractor="augmented", kind="depth", B=6, n_replay=24),
    "depth_augmented_b64": dict(extractor="augmented", kind="depth", B=64, n_replay=200),
    "rgbd_augmented": dict(extractor="augmented", kind="rgbd", B=5, n_replay=12),
    "depth_nature": dict(extractor="nature", kind="depth", B=7, n_replay=16, act_dim=3),
    "mlp_features": dict(extractor="mlp", B=64, n_replay=256),
    "mlp_wide_3layer": dict(extractor="mlp", B=33, n_replay=64, layers=(128, 128, 32), obs_dim=37),
    "depth_no_normalize": dict(extractor="augmented", kind="depth", B=4, n_replay=16, normalize=False),
    "batch_1": dict(extractor="augmented", kind="depth", B=1, n_replay=3),
    "depth_norm_obs_only": dict(extractor="augmented", kind="depth", B=8, n_replay=16, normalize="obs"),
    "mlp_norm_reward_only": dict(extractor="mlp", B=16, n_replay=32, normalize="reward"),
    "rgbd_u8_replay": dict(extractor="augmented", kind="rgbd", B=9, n_replay=20, rgb_u8=True),
    "mlp_layers_128_64": dict(extractor="mlp", B=20, n_replay=64, layers=(128, 64), obs_dim=37),     # two column blocks per wave, ragged
    "mlp_layers_96_72": dict(extractor="mlp", B=17, n_replay=64, layers=(96, 72), obs_dim=29, act_dim=7),
    "depth_layers128_b64": dict(extractor="augmented", kind="depth", B=64, n_replay=200, layers=(128, 128)),   # SAC_real_2m_buffer_128/config.yaml
    "depth_augmented_b128": dict(extractor="augmented", kind="depth", B=128, n_replay=300),   # per-rank shape of configs[4]
}


@pytest.mark.parametrize("name", list(CASES))
def test_update_matches_oracle(name):
    case = pu.make_case(n_steps=2, **CASES[name])
    ref, orc = pu.oracle_run(case)
    eng = pu.engine_setup(case)
    eng.train(1, case["idx"][:1], case["eps"][:1])
    pu.compare_first_step(eng, case, ref[0])
    eng.train(1, case["idx"][1:2], case["eps"][1:2])
    pu.compare_params(eng, orc, case["spec"].lr, 2)
    eng.close()


@pytest.mark.parametrize("var", ["GRL_NO_FUSED_HEADS=1", "GRL_NO_HEADS_MFMA=1", "GRL_NO_V2=1", "GRL_TUNE=fused_adam=0",
                                 "GRL_TUNE=conv_stack=0", "GRL_TUNE=conv_stack_bwd=1"])
def test_fallback_paths_match_oracle(monkeypatch, var):
    """The per-layer GEMM heads, the two-launch VALU head chains (heads_kernels.h), the scalar-gather igemm_kernel, the
    separate Adam launch and one implicit-GEMM launch per convolution instead of the sample-local stack (conv_stack.h) --
    the kernels other shapes fall back to -- stay correct."""
    monkeypatch.setenv(*var.split("=", 1))
    case = pu.make_case(n_steps=2, extractor="augmented", kind="depth", B=16, n_replay=48)
    ref, orc = pu.oracle_run(case)
    eng = pu.engine_setup(case)
    eng.train(1, case["idx"][:1], case["eps"][:1])
    pu.compare_first_step(eng, case, ref[0])
    eng.train(1, case["idx"][1:2], case["eps"][1:2])
    pu.compare_params(eng, orc, case["spec"].lr, 2)
    eng.close()


def test_headline_config_b256():
    """BASELINE config 2: depth 64x64x2, batch 256, layers [64,64], A=5 -- three updates."""
    case = pu.make_case(extractor="augmented", kind="depth", B=256, n_replay=600, n_steps=3)
    ref, orc = pu.oracle_run(case)
    eng = pu.engine_setup(case)
    eng.train(1, case["idx"][:1], case["eps"][:1])
    pu.compare_first_step(eng, case, ref[0])
    eng.train(2, case["idx"][1:3], case["eps"][1:3])
    pu.compare_params(eng, orc, case["spec"].lr, 3)
    m = eng.metrics()
    assert abs(m["policy_loss"] - float(ref[2]["policy_loss"])) <= 2e-3 * abs(float(ref[2]["policy_loss"])) + 1e-4
    eng.close()


def test_rgbd_config_b256():
    """BASELINE configs[3]: RGB-D observation 64x64x5 (4 image channels + the direct-feature channel), batch 256,
    augmented extractor, layers [64,64], A=5 -- three updates (trained_models/SAC_full_rgbd/config.yaml:25-33,47-50)."""
    case = pu.make_case(extractor="augmented", kind="rgbd", B=256, n_replay=400, n_steps=3)
    ref, orc = pu.oracle_run(case)
    eng = pu.engine_setup(case)
    eng.train(1, case["idx"][:1], case["eps"][:1])
    pu.compare_first_step(eng, case, ref[0])
    eng.train(2, case["idx"][1:3], case["eps"][1:3])
    pu.compare_params(eng, orc, case["spec"].lr, 3)
    eng.close()


def test_rgbd_u8_replay_config_b256():
    """The configuration `bench.py --workload sac_rgbd` runs (configs[3] with the byte-colour ring, grl_config.replay_rgb_u8)
    at its batch size: the camera's integer colours survive the packed ring bit-exactly (compare_first_step asserts the
    normalised minibatch) and the update matches the oracle."""
    case = pu.make_case(extractor="augmented", kind="rgbd", B=256, n_replay=400, n_steps=2, rgb_u8=True)
    ref, orc = pu.oracle_run(case)
    eng = pu.engine_setup(case)
    eng.train(1, case["idx"][:1], case["eps"][:1])
    pu.compare_first_step(eng, case, ref[0])
    eng.train(1, case["idx"][1:2], case["eps"][1:2])
    pu.compare_params(eng, orc, case["spec"].lr, 2)
    eng.close()


def test_cpu_reference_config_nature_b64():
    """BASELINE configs[0]: simplified_object_picking.yaml with depth observations -> sacCnn with the default
    nature_cnn over both channels, default layers [64,64], A=3, batch 64, no VecNormalize
    (sb_helper.py:91-93; simplified_object_picking.yaml:67,80)."""
    case = pu.make_case(extractor="nature", kind="depth", B=64, n_replay=200, act_dim=3, normalize=False, n_steps=3)
    ref, orc = pu.oracle_run(case)
    eng = pu.engine_setup(case)
    eng.train(1, case["idx"][:1], case["eps"][:1])
    pu.compare_first_step(eng, case, ref[0])
    eng.train(2, case["idx"][1:3], case["eps"][1:3])
    pu.compare_params(eng, orc, case["spec"].lr, 3)
    eng.close()


@pytest.mark.parametrize("kw", [dict(extractor="augmented", kind="depth", B=256, n_replay=300),
                                dict(extractor="mlp", B=64, n_replay=128)], ids=["depth_b256", "mlp_b64"])
def test_optimiser_step_is_tf_adam_on_identical_inputs(kw):
    """Adam / Polyak kernels in isolation: the device's own gradients, moments and weights are handed to the
    oracle's TF-Adam; weights must agree to 1e-3 of ONE step, moments to 1e-6 (parity_util.check_apply_step)."""
    pu.check_optimiser_steps(pu.make_case(n_steps=3, **kw), n=4)


def test_graph_replay_equals_eager(monkeypatch):
    case = pu.make_case(extractor="augmented", kind="depth", B=16, n_replay=64, n_steps=4)
    engs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("GRL_NO_GRAPH", flag)
        eng = pu.engine_setup(case)
        eng.train(4, case["idx"], case["eps"])
        engs.append(eng.get_parameters())
        eng.close()
    for n in engs[0]:
        assert np.array_equal(engs[0][n], engs[1][n]), n


@pytest.mark.parametrize("var", ["fused_adam", "gather_prefetch"])
def test_launch_plan_switches_do_not_touch_arithmetic(monkeypatch, var):
    """Adam fused into the slab reduction or as a launch of its own, the next minibatch gathered by the update's last
    launch or by one of its own: the switches only regroup work -- parameters after three updates (explicit minibatches)
    and after further calls of several updates on the device RNG are bit-identical."""
    case = pu.make_case(extractor="augmented", kind="depth", B=32, n_replay=96, n_steps=3)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("GRL_TUNE", "%s=%s" % (var, flag))
        eng = pu.engine_setup(case)
        eng.train(3, case["idx"], case["eps"])
        eng.train_device(5)
        eng.train_device(2)
        outs.append(eng.get_parameters())
        eng.close()
    for n in outs[0]:
        assert np.array_equal(outs[0][n], outs[1][n]), n


@pytest.mark.parametrize("kind", ["sac_cnn", "sac_mlp"])
def test_updates_grouped_into_one_graph_are_bit_identical(monkeypatch, kind):
    """Calls of several updates on the device RNG send their identical updates out in groups of up to 16 per hipGraph
    (engine.hip: run_repeated; GRL_TUNE graph_updates=1 keeps one graph per update): same kernels in the same order --
    parameters after 37 updates (1 + 32 + 2 + 1 + 1 as first / 16+16 / 2 / 1 / last) are bit-identical."""
    case = (pu.make_case(extractor="augmented", kind="depth", B=32, n_replay=200, n_steps=1) if kind == "sac_cnn"
            else pu.make_case(extractor="mlp", B=64, n_replay=256, n_steps=1))
    outs = []
    for flag in ("1", "16"):
        monkeypatch.setenv("GRL_TUNE", "graph_updates=" + flag)
        eng = pu.engine_setup(case)
        eng.train_device(37)
        eng.train_device(3)
        outs.append(eng.get_parameters())
        eng.close()
    for n in outs[0]:
        assert np.array_equal(outs[0][n], outs[1][n]), n


def test_split_api_equals_fused_and_is_deterministic():
    case = pu.make_case(extractor="augmented", kind="depth", B=16, n_replay=64, n_steps=3)
    outs = []
    for mode in ("fused", "split", "fused"):
        eng = pu.engine_setup(case)
        if mode == "fused":
            eng.train(3, case["idx"], case["eps"])
        else:
            for s in range(3):
                eng.compute_grads(case["idx"][s:s + 1], case["eps"][s:s + 1])
                eng.apply_grads(1.0)
        outs.append(eng.get_parameters())
        eng.close()
    for n in outs[0]:
        assert np.array_equal(outs[0][n], outs[1][n]), n
        assert np.array_equal(outs[0][n], outs[2][n]), n


def test_device_rng_mode_trains():
    case = pu.make_case(extractor="augmented", kind="depth", B=32, n_replay=100)
    eng = pu.engine_setup(case)
    p0 = eng.get_parameters()
    eng.train(5)
    m = eng.metrics()
    assert all(np.isfinite(v) for v in m.values()), m
    p1 = eng.get_parameters()
    assert not np.array_equal(p0["model/pi/cnn1/w:0"], p1["model/pi/cnn1/w:0"])
    # same seed -> same trajectory
    eng2 = pu.engine_setup(case)
    eng2.train(5)
    p2 = eng2.get_parameters()
    for n in p1:
        assert np.array_equal(p1[n], p2[n]), n
    eng.close(); eng2.close()


def test_replay_ring_wraps():
    case = pu.make_case(extractor="mlp", B=8, n_replay=10, n_steps=1)
    eng = pu.engine_setup(case)
    tr = case["tr"]
    assert eng.replay_size() == 10
    eng.replay_add(tr["obs"][:4] + 1.0, tr["act"][:4], tr["rew"][:4], tr["next_obs"][:4], tr["done"][:4])
    assert eng.replay_size() == 10
    # slots 0..3 were overwritten: sampling index 0 must now see obs+1
    idx = np.zeros((1, 8), np.int64)
    eng.train(1, idx, case["eps"][:1])
    st = case["stats"]
    want = osac.normalize_obs(tr["obs"][:1] + 1.0, st["mean"], st["var"]).astype(np.float32)
    got = eng.fetch("feat_pi", (8, 104))[:1, :101]
    assert np.array_equal(got, want)
    eng.close()


def test_act_matches_oracle():
    case = pu.make_case(extractor="augmented", kind="depth", B=2, n_replay=4)
    eng = pu.engine_setup(case)
    orc = osac.SacOracle(case["spec"], case["params"])
    st = case["stats"]
    obs = osac.normalize_obs(case["tr"]["obs"][:3], st["mean"], st["var"]).astype(np.float32)
    eps = np.random.default_rng(3).standard_normal((3, 5)).astype(np.float32)
    pu.close(eng.act(obs, True), orc.act(obs, True), what="deterministic action")
    pu.close(eng.act(obs, False, eps), orc.act(obs, False, eps), what="stochastic action")
    eng.close()


def test_shipped_sac_mlp_weights_on_real_observations():
    """Golden vectors: the reference's trained SAC-MLP parameters on the two real 101-d observations
    preserved in its vecnormalize.pkl (tests/golden/oracle_pins.json, made by scripts/make_golden.py)."""
    pins = json.load(open(os.path.join(GOLD, "oracle_pins.json")))
    z = np.load(os.path.join(GOLD, "sac_mlp_best_model.npz"))
    vn = np.load(os.path.join(GOLD, "vecnorm_encoder.npz"))
    cfg = _capi.make_config("mlp", obs_dim=101, act_dim=5, layers=(64, 64), batch_size=2, replay_capacity=4,
                            act_batch=2)
    eng = SacEngine(cfg)
    eng.set_parameters({k: z[k] for k in z.files})
    obs = osac.normalize_obs(vn["real_obs"], vn["mean"], vn["var"]).astype(np.float32)
    a = eng.act(obs, True)
    pu.close(a, np.asarray(pins["sac_mlp_real_obs"]["det_action"], np.float32), what="golden action")
    eng.close()


def test_autoencoder_features_match_golden():
    """encoder_files/new_gripper_encoder weights on the six real depth frames."""
    W = np.load(os.path.join(GOLD, "ae_new_gripper_encoder.npz"))
    frames = np.load(os.path.join(GOLD, "depth_frames.npz"))["frames"]
    want = np.load(os.path.join(GOLD, "ae_encodings.npz"))["z"]
    cfg = _capi.make_config("mlp", obs_dim=101, act_dim=5, batch_size=2, replay_capacity=4, act_batch=6)
    eng = SacEngine(cfg)
    order = ["encoder/conv2d_1/kernel", "encoder/conv2d_1/bias", "encoder/conv2d_2/kernel", "encoder/conv2d_2/bias",
             "encoder/conv2d_3/kernel", "encoder/conv2d_3/bias", "encoder/dense_1/kernel", "encoder/dense_1/bias"]
    eng.load_encoder([W[k] for k in order])
    got = eng.encode(frames[..., None])
    pu.close(got, want, atol=2e-5, rtol=2e-4, what="auto-encoder features")
    # a single frame through the batch path gives the same row
    pu.close(eng.encode(frames[2:3, ..., None]), want[2:3], atol=2e-5, rtol=2e-4, what="single frame")
    eng.close()


def test_error_paths():
    cfg = _capi.make_config("mlp", obs_dim=11, act_dim=3, batch_size=4, replay_capacity=8)
    eng = SacEngine(cfg)
    with pytest.raises(_capi.GrlError):
        eng.train(1)                      # empty replay
    with pytest.raises(_capi.GrlError):
        eng.encode(np.zeros((1, 64, 64, 1), np.float32))   # encoder not loaded
    with pytest.raises(_capi.GrlError):
        eng.set_parameters({"nope": np.zeros(3)})
    eng.close()


def test_multi_update_call_prefetch_is_bit_identical_b256(monkeypatch):
    """Headline shape, device RNG: one call of 6 updates (the gather of update t+1 rides on the last launch of update t,
    the head launch opens each update) == 6 calls of one update == the same with GRL_TUNE gather_prefetch=0: parameters,
    Adam moments and the last drawn indices bit for bit."""
    def run(split, off=False):
        if off:
            monkeypatch.setenv("GRL_TUNE", "gather_prefetch=0")
        case = pu.make_case(extractor="augmented", kind="depth", B=256, n_replay=512, n_steps=1)
        eng = pu.engine_setup(case)
        for n in split:
            eng.train(n)
        out = (eng.get_parameters(), eng.fetch("adam_m").copy(), eng.fetch("adam_v").copy(), eng.fetch("idx_raw").copy(),
               eng.metrics())
        eng.close()
        if off:
            monkeypatch.delenv("GRL_TUNE")
        return out
    ref = run([1] * 6)
    for got in (run([6]), run([2, 4]), run([6], off=True)):
        assert all(np.array_equal(ref[0][n], got[0][n]) for n in ref[0])
        assert all(np.array_equal(a, b) for a, b in zip(ref[1:4], got[1:4]))
        assert ref[4] == got[4]


@pytest.mark.parametrize("kw", [dict(kind="depth"), dict(kind="rgbd", rgb_u8=True)], ids=["depth", "rgbd_u8"])
def test_image_gather_riding_on_the_head_launch_is_bit_identical_b256(monkeypatch, kw):
    """plan_sac "gather_ride" at the bench shapes (depth float32 ring, RGB-D byte-colour ring): the images of minibatch t+1 are
    gathered by extra workgroups of update t's head launch into the second image buffer.  Calls of 3 / 4 updates (both buffer
    parities), a call longer than one graph (37 = 1 + 32 + 2 + 1 + 1: grouped sequences of both parities) and the same with the
    rider switched off must all leave the bits of single-update calls."""
    case = pu.make_case(extractor="augmented", B=256, n_replay=512, n_steps=1, **kw)

    def run(split, tune=None):
        if tune:
            monkeypatch.setenv("GRL_TUNE", tune)
        eng = pu.engine_setup(case)
        for n in split:
            eng.train(n)
        out = (eng.get_parameters(), eng.fetch("adam_m").copy(), eng.fetch("adam_v").copy(), eng.fetch("idx_raw").copy(),
               eng.fetch("x_obs").copy(), eng.metrics())
        eng.close()
        if tune:
            monkeypatch.delenv("GRL_TUNE")
        return out
    ref = run([1] * 7)
    for got in (run([3, 4], tune="gather_ride=1"), run([7], tune="gather_ride=0")):     # (the byte-colour ring's default is off)
        assert all(np.array_equal(ref[0][n], got[0][n]) for n in ref[0])
        assert all(np.array_equal(a, b) for a, b in zip(ref[1:4], got[1:4]))
        assert ref[5] == got[5]
    long_ref = run([37], tune="gather_ride=0")
    got = run([37], tune="gather_ride=1")
    assert all(np.array_equal(long_ref[0][n], got[0][n]) for n in long_ref[0])
    assert all(np.array_equal(a, b) for a, b in zip(long_ref[1:4], got[1:4]))
    assert long_ref[5] == got[5]


def test_conv_stack_is_bit_identical_to_the_per_layer_launches_where_the_order_is_the_same(monkeypatch):
    """csrc/conv_stack.h sums every output in increasing k, one fmaf per step -- the order of the per-layer implicit-GEMM
    launches whose reduction stays in one wave (conv1, conv2; conv3's per-layer launch adds the partial sums of wave pairs).
    Layer-1 and layer-2 activations of both trained networks must therefore be the SAME BITS with and without the stack at
    the headline shape, layer 3 and everything downstream equal to rounding -- and no ReLU unit may change sides
    (scripts/conv_stack_flips.py tells the story of the k-permuted first cut)."""
    case = pu.make_case(extractor="augmented", kind="depth", B=256, n_replay=600, n_steps=1)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GRL_TUNE", "conv_stack=" + mode)
        eng = pu.engine_setup(case)
        eng.train(1, case["idx"][:1], case["eps"][:1])
        out[mode] = {k: eng.fetch(k) for k in ("a1_pair", "a2_pi", "a2_vf", "a3_pi", "a3_vf", "feat_pi", "feat_vf", "feat_tgt")}
        out[mode]["G"] = eng.get_gradients()
        eng.close()
    s, l = out["1"], out["0"]
    for k in ("a1_pair", "a2_pi", "a2_vf"):
        assert np.array_equal(s[k], l[k]), k
    for k in ("a3_pi", "a3_vf", "feat_pi", "feat_vf", "feat_tgt"):
        assert np.array_equal(s[k] > 0, l[k] > 0), "a ReLU unit changed sides: " + k
        assert np.abs(s[k] - l[k]).max() <= 1e-7, k
    for n, g in l["G"].items():
        pu.close_rel_max(s["G"][n], g, rel=1e-5, what="stack vs per-layer grad " + n)
