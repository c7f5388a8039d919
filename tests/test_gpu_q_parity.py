"""DQN / BDQ update on the MI355X (through the C ABI) against the oracle (oracle/dqn.py)."""
import pytest

import q_parity_util as qu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(qu.CASES))
def test_q_update_matches_oracle(name):
    qu.run_and_compare(qu.make_q_case(**qu.CASES[name]))


@pytest.mark.parametrize("name", ["dqn_reference_shape", "bdq_reference_shape"])
def test_q_update_per_layer_gemm_fallback(name, monkeypatch):
    """GRL_TUNE fused_q=0: the per-layer GEMM launches stay correct next to the row-local chains."""
    monkeypatch.setenv("GRL_TUNE", "fused_q=0")
    qu.run_and_compare(qu.make_q_case(**qu.CASES[name]))


@pytest.mark.parametrize("name", ["dqn_reference_shape", "bdq_reference_shape", "bdq_baseline_config3"])
def test_q_update_valu_chains(name, monkeypatch, capfd):
    """GRL_TUNE q_mfma=0: the VALU stage chains of q_kernels.h (what networks wider than 64 units run on) stay correct next
    to the matrix-core stages of q_mfma.h -- and the default plan does take the matrix-core kernels for these shapes."""
    monkeypatch.setenv("GRL_PLAN_DUMP", "1")
    qu.run_and_compare(qu.make_q_case(**qu.CASES[name]))
    assert "matrix-core stages" in capfd.readouterr().err
    monkeypatch.setenv("GRL_TUNE", "q_mfma=0")
    qu.run_and_compare(qu.make_q_case(**qu.CASES[name]))
    assert "VALU stages" in capfd.readouterr().err


@pytest.mark.parametrize("name", ["dqn_reference_shape", "bdq_reference_shape", "bdq_baseline_config3", "bdq_5_branches"])
def test_q_update_separate_loss_and_weight_gradient_launches(name, monkeypatch, capfd):
    """The default plan forms the loss and the weight gradients inside the backward chains (q_chain.h: five launches per
    update); GRL_TUNE q_chain=0 keeps the loss and weight-gradient launches of their own (seven).  Both against the oracle --
    and the TD errors, priorities and parameters of the two agree: the loss arithmetic is the same instruction for
    instruction, only the summation order of the weight gradients differs."""
    import numpy as np
    monkeypatch.setenv("GRL_PLAN_DUMP", "1")
    qu.run_and_compare(qu.make_q_case(**qu.CASES[name]))
    assert "inside the backward chains" in capfd.readouterr().err
    monkeypatch.setenv("GRL_TUNE", "q_chain=0")
    qu.run_and_compare(qu.make_q_case(**qu.CASES[name]))
    assert "inside the backward chains" not in capfd.readouterr().err
    outs = []
    for tune in ("q_chain=1", "q_chain=0"):
        monkeypatch.setenv("GRL_TUNE", tune)
        case = qu.make_q_case(**qu.CASES[name])
        eng = qu.q_engine_setup(case)
        eng.train(1, case["idx"][:1], case["weights"][:1])
        outs.append((eng.td_errors(), eng.priorities(), eng.get_gradients(), eng.get_parameters()))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    for k, g1 in outs[1][2].items():
        assert np.max(np.abs(outs[0][2][k] - g1)) <= 1e-5 * max(1e-6, float(np.max(np.abs(g1)))), k
    for k in outs[0][3]:
        np.testing.assert_allclose(outs[0][3][k], outs[1][3][k], rtol=0, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("name", ["bdq_reference_shape", "bdq_baseline_config3", "bdq_5_branches"])
def test_q_update_tower_weight_gradients_at_the_end_of_their_chains(name, monkeypatch):
    """GRL_TUNE q_chain_late=0: every tower chain forms its weight-gradient slabs itself (what networks without a trunk do);
    the default for networks with a trunk hands them to workgroups of the trunk launch.  Same tiles, same sums: bit-identical."""
    import numpy as np
    outs = []
    for tune in ("q_chain_late=0", "q_chain_late=1"):
        monkeypatch.setenv("GRL_TUNE", tune)
        case = qu.make_q_case(**qu.CASES[name])
        if tune.endswith("0"):
            qu.run_and_compare(case)
        eng = qu.q_engine_setup(case)
        eng.train(2, case["idx"][:2], case["weights"][:2])
        outs.append((eng.get_gradients(), eng.get_parameters()))
        eng.close()
    for k in outs[0][0]:
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k


@pytest.mark.parametrize("name", ["dqn_reference_shape", "bdq_reference_shape"])
def test_q_update_three_launch_apply(name, monkeypatch):
    """GRL_TUNE fused_qapply=0: slab reduction, clip_by_norm and Adam as three launches instead of the fused one."""
    monkeypatch.setenv("GRL_TUNE", "fused_qapply=0")
    qu.run_and_compare(qu.make_q_case(**qu.CASES[name]))


def test_bdq_uniform_replay_on_the_device_rng():
    """configs[2] as the YAML selects it (gripper_grasp.yaml:104-118: 101-d observations, 5 x 33 bins, batch 64,
    prioritized_replay False): next to the oracle comparison of `bdq_baseline_config3_uniform` above, the production form --
    indices drawn on the device, every importance weight exactly 1, all indices inside the stored range, deterministic."""
    import numpy as np
    outs = []
    for _ in range(2):
        case = qu.make_q_case(**qu.CASES["bdq_baseline_config3_uniform"])
        eng = qu.q_engine_setup(case)
        eng.train(7)                                   # device RNG: idx / weights NULL
        idx, w = eng.sampled_indices(), eng.importance_weights()
        assert np.all(w == 1.0) and idx.min() >= 0 and idx.max() < eng.replay_size() and len(np.unique(idx)) > 8
        outs.append(eng.get_parameters())
        eng.close()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_q_update_with_vecnormalize():
    qu.run_and_compare(qu.make_q_case(normalize=True, **qu.CASES["bdq"]))


def test_q_device_rng_mode_trains():
    import numpy as np
    case = qu.make_q_case(**qu.CASES["bdq_reference_shape"])
    eng = qu.q_engine_setup(case)
    p0 = eng.get_parameters()
    eng.train(5)
    p1 = eng.get_parameters()
    assert np.isfinite(eng.metrics()["policy_loss"])
    assert not np.array_equal(p0["bdq/model/common_net/fully_connected/weights:0"],
                              p1["bdq/model/common_net/fully_connected/weights:0"])
    eng.close()


def test_prioritised_replay_gpu():
    """Segment-tree sampler (indices bit-exact), importance weights and priority write-back on the MI355X
    against oracle/per.py; the last case is BASELINE configs[2] (BDQ 5 x 33 bins on 101-d observations, batch 64,
    prioritised replay on -- gripper_grasp.yaml:102,104-118)."""
    qu.per_check()
    qu.per_check(cap=5000, n_store=5000, B=64, n_steps=3, seed=11)
    qu.per_check(cap=20000, n_store=17000, B=64, n_steps=4, seed=5, case_name="bdq_baseline_config3")


def test_shipped_bdq_weights_on_real_observations():
    """Golden vectors: the reference's trained BDQ network (trained_models/BDQ_33pads_big/best_model: 100-d
    observations, 3 branches x 33 bins, layers [[512,256],[128],[128]]) on the two real auto-encoder feature vectors
    the reference ships (tests/golden/oracle_pins.json, made by scripts/make_golden.py): dueling Q-values and greedy bins."""
    import json
    import os
    import numpy as np
    import parity_util as pu
    from grasp_rl import _capi
    from grasp_rl.engine import QEngine
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    pins = json.load(open(os.path.join(gold, "oracle_pins.json")))["bdq_real_obs"]
    z = np.load(os.path.join(gold, "bdq_33_big_best_model.npz"))
    sp = pins["spec"]
    cfg = _capi.make_q_config("bdq", sp["obs_dim"], sp["branches"], sp["bins"], common=tuple(sp["common"]),
                              branch_hidden=(sp["branch"],), value_hidden=(sp["value"],), batch_size=4, act_batch=2,
                              replay_capacity=8)
    eng = QEngine(cfg)
    eng.set_parameters({k: z[k] for k in z.files})
    obs = np.load(os.path.join(gold, "vecnorm_encoder.npz"))["real_obs"][:, :sp["obs_dim"]].astype(np.float32)
    q = eng.q_values(obs)
    pu.close(q, np.asarray(pins["q_values"], np.float32), atol=2e-5, rtol=2e-4, what="golden BDQ Q-values")
    assert np.array_equal(q.argmax(axis=2), np.asarray(pins["greedy_bins"]))
    assert np.allclose(q.mean(axis=2), q.mean(axis=2)[:, :1], atol=1e-4)       # every branch mean equals V (dueling aggregation)
    eng.close()


@pytest.mark.parametrize("name,cap,n_store", [("bdq_reference_shape", 3000, 2500), ("dqn_reference_shape", 5000, 5000)])
def test_per_multi_update_call_keeps_the_block_sums_current(name, cap, n_store, monkeypatch):
    """Prioritised replay on the device RNG: ONE call of n updates == n calls of one update, bit for bit
    (q_parity_util.per_multi_update_check; the CPU suite runs it on the emulation build)."""
    qu.per_multi_update_check(monkeypatch, name, cap, n_store)


@pytest.mark.parametrize("name", ["bdq_baseline_config3_uniform", "dqn_reference_shape"])
def test_uniform_multi_update_call_prefetches_the_next_minibatch(name, monkeypatch):
    """configs[2] as gripper_grasp.yaml selects it (uniform replay) and the reference's DQN shape: one call of n updates -- draw +
    gather of the next minibatch on the apply launch, four launches per update -- == n single-update calls, bit for bit
    (q_parity_util.uniform_multi_update_check; the CPU suite runs it on the emulation build)."""
    qu.uniform_multi_update_check(monkeypatch, name, 3000, n=9)
