"""CPU check of the HOST logic of the engine (parameter layout, addressing tables, problem
descriptors, launch order, Adam/Polyak bookkeeping) against the oracle.

engine.hip is compiled with g++ -DGRL_HOSTEMU (csrc/hostemu.h): the HIP runtime is stubbed and the
MFMA kernel is replaced by a plain reference loop over the same descriptors, so this validates every
table and descriptor the GPU kernels consume -- not the GPU kernels themselves (tests/test_gpu_parity.py
does that on the MI355X).  The emulation build is test infrastructure, never loaded by the product.
"""
import numpy as np
import pytest

import parity_util as pu
from grasp_rl import _capi
from hostemu_backend import NumpyHostBackend
from oracle import sac as osac

CASES = {
    "depth_augmented": dict(extractor="augmented", kind="depth", B=6, n_replay=24),
    "rgbd_augmented": dict(extractor="augmented", kind="rgbd", B=3, n_replay=12),
    "depth_nature": dict(extractor="nature", kind="depth", B=4, n_replay=16, act_dim=3),
    "mlp_features": dict(extractor="mlp", B=16, n_replay=64),
    "mlp_wide_3layer": dict(extractor="mlp", B=8, n_replay=32, layers=(128, 128, 32), obs_dim=37),
    "depth_no_normalize": dict(extractor="augmented", kind="depth", B=4, n_replay=16, normalize=False),
    # VecNormalize(norm_obs=True, norm_reward=False) / (False, True): each flag honoured on its own
    "depth_norm_obs_only": dict(extractor="augmented", kind="depth", B=4, n_replay=16, normalize="obs"),
    "mlp_norm_reward_only": dict(extractor="mlp", B=8, n_replay=32, normalize="reward"),
    # RGB-D ring with byte colours (grl_config.replay_rgb_u8): lossless on the camera's integer colours
    "rgbd_u8_replay": dict(extractor="augmented", kind="rgbd", B=3, n_replay=12, rgb_u8=True),
}


def test_layout_matches_tf_names(hostemu_lib):
    """Names, order and shapes must equal the TF variable list of the shipped zips."""
    for kw in (dict(extractor="augmented", kind="depth"), dict(extractor="mlp"), dict(extractor="nature", kind="depth")):
        case = pu.make_case(B=2, n_replay=4, **kw)
        eng = pu.SacEngine(case["cfg"], backend=NumpyHostBackend(), lib_path=hostemu_lib)
        ref = osac.param_shapes(case["spec"])
        assert [t[0] for t in eng.table] == list(ref.keys())
        for name, off, numel, shape, tr in eng.table:
            assert tuple(shape) == tuple(ref[name]), name
            assert off % 4 == 0
            assert tr == (not name.startswith("target/"))
        eng.close()


@pytest.mark.parametrize("name", list(CASES))
def test_plan_matches_oracle(hostemu_lib, name):
    case = pu.make_case(n_steps=2, **CASES[name])
    ref, orc = pu.oracle_run(case)
    eng = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    eng.train(1, case["idx"][:1], case["eps"][:1])
    pu.compare_first_step(eng, case, ref[0])
    eng.train(1, case["idx"][1:2], case["eps"][1:2])
    pu.compare_params(eng, orc, case["spec"].lr, 2)
    # split API == fused API
    eng2 = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    for s in range(2):
        eng2.compute_grads(case["idx"][s:s + 1], case["eps"][s:s + 1])
        eng2.apply_grads(1.0)
    Pa, Pb = eng.get_parameters(), eng2.get_parameters()
    for n in Pa:
        assert np.array_equal(Pa[n], Pb[n]), n
    eng.close(); eng2.close()


@pytest.mark.parametrize("name", ["depth_augmented", "mlp_features"])
def test_optimiser_step_is_tf_adam_on_identical_inputs(hostemu_lib, name):
    """Adam / Polyak bookkeeping of the plan in isolation: same gradients, moments and weights into the oracle's
    TF-Adam -> weights within 1e-3 of one step, moments 1e-6 relative (tests/parity_util.py: check_apply_step)."""
    case = pu.make_case(n_steps=2, **CASES[name])
    pu.check_optimiser_steps(case, backend=NumpyHostBackend(), lib_path=hostemu_lib, n=3)


@pytest.mark.parametrize("var", ["GRL_NO_HEADS_MFMA=1", "GRL_TUNE=conv_stack=0", "GRL_TUNE=conv_stack_bwd=1"])
def test_two_launch_head_chains_still_match(hostemu_lib, monkeypatch, var):
    """GRL_NO_HEADS_MFMA=1: the plan with heads_fwd + heads_bwd (heads_kernels.h) instead of the fused launch;
    GRL_TUNE conv_stack=0: one implicit-GEMM launch per convolution instead of the sample-local stack (conv_stack.h);
    GRL_TUNE conv_stack_bwd=1: the opt-in sample-local backward-data launch."""
    monkeypatch.setenv(*var.split("=", 1))
    case = pu.make_case(n_steps=2, **CASES["depth_augmented"])
    ref, orc = pu.oracle_run(case)
    eng = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    eng.train(1, case["idx"][:1], case["eps"][:1])
    pu.compare_first_step(eng, case, ref[0])
    eng.train(1, case["idx"][1:2], case["eps"][1:2])
    pu.compare_params(eng, orc, case["spec"].lr, 2)
    eng.close()


def test_act_matches_oracle(hostemu_lib):
    case = pu.make_case(extractor="augmented", kind="depth", B=2, n_replay=4)
    eng = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    orc = osac.SacOracle(case["spec"], case["params"])
    st = case["stats"]
    obs = osac.normalize_obs(case["tr"]["obs"][:3], st["mean"], st["var"]).astype(np.float32)
    eps = np.random.default_rng(3).standard_normal((3, 5)).astype(np.float32)
    pu.close(eng.act(obs, True), orc.act(obs, True), what="deterministic action")
    pu.close(eng.act(obs, False, eps), orc.act(obs, False, eps), what="stochastic action")
    eng.close()


def test_60_update_trajectory_follows_the_oracle(hostemu_lib):
    """The long-trajectory comparison of tests/test_gpu_learning.py (200 updates of the CNN on the GPU), here on the
    MLP variant through the emulation build: identical index / noise streams, nine metrics within 2 %."""
    from test_gpu_learning import trajectory_check
    case = pu.make_case(extractor="mlp", B=16, n_replay=128, n_steps=60, seed=4)
    eng = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    worst, _ = trajectory_check(case, eng, 60, every=10)
    assert max(worst.values()) < 0.02
    eng.close()


def test_multi_update_calls_prefetch_the_next_minibatch_bit_identically(hostemu_lib, monkeypatch):
    """A call of n updates on the device RNG gathers minibatch t+1 inside the last launch of update t
    (reduce_slabs_gather_kernel) and opens each update in the head launch: parameters, Adam state and the RNG counter
    must equal n single-update calls bit for bit (and the switch GRL_TUNE gather_prefetch=0 must change nothing)."""
    def run(split, env=None):
        if env:
            monkeypatch.setenv(*env)
        case = pu.make_case(extractor="mlp", B=16, n_replay=64, n_steps=1)
        eng = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
        for n in split:
            eng.train(n)
        P = eng.get_parameters()
        extra = [eng.fetch("adam_m").copy(), eng.fetch("adam_v").copy(), eng.fetch("idx_raw").copy()]
        m = eng.metrics()
        eng.close()
        if env:
            monkeypatch.delenv(env[0])
        return P, extra, m
    Pa, xa, ma = run([1, 1, 1, 1, 1])
    for split in ([5], [2, 3], [3, 1, 1]):
        Pb, xb, mb = run(split)
        for n in Pa:
            assert np.array_equal(Pa[n], Pb[n]), (split, n)
        for a, b in zip(xa, xb):
            assert np.array_equal(a, b), split
        assert ma == mb
    Pc, xc, _ = run([5], env=("GRL_TUNE", "gather_prefetch=0"))
    assert all(np.array_equal(Pa[n], Pc[n]) for n in Pa) and all(np.array_equal(a, b) for a, b in zip(xa, xc))
    # the CNN plan too (one shape)
    case = pu.make_case(extractor="augmented", kind="depth", B=4, n_replay=16, n_steps=1)
    outs = []
    for split in ([1, 1, 1], [3]):
        eng = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
        for n in split:
            eng.train(n)
        outs.append(eng.get_parameters())
        eng.close()
    assert all(np.array_equal(outs[0][n], outs[1][n]) for n in outs[0])


def test_image_gather_riding_on_the_head_launch_is_bit_identical(hostemu_lib, monkeypatch, capfd):
    """plan_sac "gather_ride" (CNN policies, batch a multiple of 16): the images of minibatch t+1 are gathered by extra workgroups
    of update t's head launch into the image buffer update t does not read, the per-row extras by the reduction launch; the
    riders draw with DevScalars.rng_img.  Calls of 2 .. 5 updates (both buffer parities at the end of a call, calls that start on
    either) must leave the parameters, the Adam state, the last minibatch's indices and the metrics of single-update calls."""
    case = pu.make_case(extractor="augmented", kind="depth", B=16, n_replay=48, n_steps=1)

    def run(split, tune=None):
        if tune:
            monkeypatch.setenv("GRL_TUNE", tune)
        eng = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
        for n in split:
            eng.train(n)
        out = (eng.get_parameters(), [eng.fetch("adam_m").copy(), eng.fetch("adam_v").copy(), eng.fetch("idx_raw").copy()], eng.metrics())
        eng.close()
        if tune:
            monkeypatch.delenv("GRL_TUNE")
        return out
    monkeypatch.setenv("GRL_PLAN_DUMP", "1")
    Pa, xa, ma = run([1, 1, 1, 1, 1])
    assert "gather_ride" in capfd.readouterr().err
    monkeypatch.delenv("GRL_PLAN_DUMP")
    for split in ([5], [2, 3], [4, 1]):
        Pb, xb, mb = run(split)
        assert all(np.array_equal(Pa[n], Pb[n]) for n in Pa), split
        assert all(np.array_equal(a, b) for a, b in zip(xa, xb)), split
        assert ma == mb, split
    Pc, xc, mc = run([5], tune="gather_ride=0")
    assert all(np.array_equal(Pa[n], Pc[n]) for n in Pa) and all(np.array_equal(a, b) for a, b in zip(xa, xc)) and ma == mc


def test_overlapped_in_graph_exchange_with_one_rank_equals_the_fused_update(hostemu_lib):
    """grl_allreduce_set_overlap with world = 1: the staged plan, the dense pieces of the bucket exchanged on channel 0 and
    the convolution pieces on channel 1 (several disjoint ranges each), Adam waiting for both -- exactly the parameters of
    compute_grads + apply_grads(1.0), also over a wrap of the per-rank chunking (ragged piece sizes)."""
    case = pu.make_case(extractor="augmented", kind="depth", B=4, n_replay=16, n_steps=3)
    a = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    b = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    a.allreduce_connect([a.allreduce_init(0, 1)])
    a.allreduce_set_overlap(True)
    for s in range(3):
        a.train_allreduce(1, case["idx"][s:s + 1], case["eps"][s:s + 1])
        b.compute_grads(case["idx"][s:s + 1], case["eps"][s:s + 1])
        b.apply_grads(1.0)
    assert a.allreduce_status() == 3
    a.allreduce_set_timeout(5000)                  # host-side bound of the waits: any time, graphs untouched; 0 = default again
    a.allreduce_set_timeout(0)
    with pytest.raises(RuntimeError):
        b.allreduce_set_timeout(5000)              # (a handle that was never initialised for the exchange has no mailbox)
    a.allreduce_set_overlap(False)                 # and back to the plain exchange on the same handle
    a.train_allreduce(1, case["idx"][:1], case["eps"][:1])
    b.compute_grads(case["idx"][:1], case["eps"][:1])
    b.apply_grads(1.0)
    assert a.allreduce_status() == 4
    Pa, Pb = a.get_parameters(), b.get_parameters()
    assert all(np.array_equal(Pa[k], Pb[k]) for k in Pa)
    a.close(); b.close()
    # vector observations have no staged plan: the switch says so
    case = pu.make_case(extractor="mlp", B=8, n_replay=32, n_steps=1)
    c = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    c.allreduce_connect([c.allreduce_init(0, 1)])
    with pytest.raises(RuntimeError):
        c.allreduce_set_overlap(True)
    c.close()


@pytest.mark.parametrize("mode", ["oneshot", "twoshot"])
def test_in_graph_exchange_on_the_device_rng_prefetches_like_the_fused_update(hostemu_lib, mode):
    """Several data-parallel updates per call on the device RNG: the gather of update t+1 rides on the (unfused) reduction of
    update t, the exchange and Adam follow -- with one rank exactly the parameters, moments and RNG position of
    grl_train_step(n) on a second handle, call after call."""
    case = pu.make_case(extractor="augmented", kind="depth", B=16, n_replay=48, n_steps=1)
    a = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    b = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    a.allreduce_connect([a.allreduce_init(0, 1)])
    a.allreduce_set_mode(mode)
    for n in (5, 1, 2):
        a.train_allreduce(n)
        b.train_device(n)
    assert a.allreduce_status() == 8
    Pa, Pb = a.get_parameters(), b.get_parameters()
    assert all(np.array_equal(Pa[k], Pb[k]) for k in Pa)
    for name in ("idx_raw", "adam_m", "adam_v"):
        assert np.array_equal(a.fetch(name), b.fetch(name)), name
    a.close(); b.close()


@pytest.mark.parametrize("mode", ["auto", "oneshot", "twoshot"])
def test_in_graph_exchange_with_one_rank_equals_the_fused_update(hostemu_lib, mode):
    """grl_allreduce_init / connect / grl_train_step_allreduce with world = 1 (the peer is the rank itself): the
    reduce + publish -> (reduce-scatter ->) apply chain must leave exactly the parameters of compute_grads +
    apply_grads(1.0), one-shot (source buffers alternating) and two-shot, also after switching between them.
    (Several ranks in several processes run on the GPU box: tests/test_gpu_data_parallel.py.)"""
    case = pu.make_case(extractor="augmented", kind="depth", B=4, n_replay=16, n_steps=3)
    a = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    b = pu.engine_setup(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    h = a.allreduce_init(0, 1)
    assert len(h) == 128
    a.allreduce_connect([h])
    a.allreduce_set_mode(mode)
    for s in range(3):
        a.train_allreduce(1, case["idx"][s:s + 1], case["eps"][s:s + 1])
        b.compute_grads(case["idx"][s:s + 1], case["eps"][s:s + 1])
        b.apply_grads(1.0)
    assert a.allreduce_status() == 3
    a.allreduce_set_mode("twoshot" if mode != "twoshot" else "oneshot")     # (idle handle: the modes may alternate)
    for s in range(2):
        a.train_allreduce(1, case["idx"][s:s + 1], case["eps"][s:s + 1])
        b.compute_grads(case["idx"][s:s + 1], case["eps"][s:s + 1])
        b.apply_grads(1.0)
    assert a.allreduce_status() == 5
    Pa, Pb = a.get_parameters(), b.get_parameters()
    assert all(np.array_equal(Pa[k], Pb[k]) for k in Pa)
    a.close(); b.close()
