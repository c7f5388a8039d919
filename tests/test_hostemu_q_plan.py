"""DQN / BDQ launch plan (tables, descriptors, loss / clip / Adam bookkeeping) against the oracle on CPU
through the TEST-ONLY g++ emulation build (see tests/test_hostemu_plan.py)."""
import pytest

import q_parity_util as qu
from hostemu_backend import NumpyHostBackend
from oracle import dqn as od


@pytest.mark.parametrize("name", ["dqn", "bdq", "bdq_5_branches", "bdq_baseline_config3", "bdq_loss_sum", "bdq_no_trunk_rescale"])
def test_q_plan_matches_oracle(hostemu_lib, name):
    case = qu.make_q_case(**qu.CASES[name])
    qu.run_and_compare(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)


@pytest.mark.parametrize("name,tune,chained", [("bdq_baseline_config3", "", True), ("bdq_baseline_config3", "q_chain_late=0", True),
                                               ("bdq_5_branches", "q_chain=0", False), ("dqn", "q_chain=0", False), ("dqn", "", True)])
def test_q_plan_chained_backward_and_its_fallbacks(hostemu_lib, name, tune, chained, monkeypatch, capfd):
    """The default plan forms the loss and the weight gradients inside the backward chains (csrc/q_chain.h; slabs per row block
    summed by the apply launch; with a trunk the towers' slabs by workgroups of the trunk launch) -- the emulation runs the same
    launch plan through sequential reference kernels (tests/hostemu/q_chain_ref1.h); GRL_TUNE q_chain=0 / q_chain_late=0 keep
    the other forms.  All against the oracle."""
    monkeypatch.setenv("GRL_PLAN_DUMP", "1")
    monkeypatch.setenv("GRL_TUNE", tune)
    case = qu.make_q_case(**qu.CASES[name])
    qu.run_and_compare(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    assert ("inside the backward chains" in capfd.readouterr().err) == chained


@pytest.mark.parametrize("name", ["dqn", "bdq_5_branches"])
def test_q_plan_per_layer_gemm_fallback(hostemu_lib, name, monkeypatch):
    """GRL_TUNE fused_q=0: one GEMM launch per layer instead of the row-local chains of q_kernels.h."""
    monkeypatch.setenv("GRL_TUNE", "fused_q=0")
    case = qu.make_q_case(**qu.CASES[name])
    qu.run_and_compare(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)


def test_q_plan_with_vecnormalize(hostemu_lib):
    case = qu.make_q_case(normalize=True, **qu.CASES["bdq"])
    qu.run_and_compare(case, backend=NumpyHostBackend(), lib_path=hostemu_lib)


def test_layout_matches_shipped_zip_names(hostemu_lib):
    """Engine variable names / order / shapes == TF names of trained_models/{DQN_4pads,BDQ_*}/*.zip."""
    for spec, kw in ((od.QSpec(algo="dqn", obs_dim=100, n_bins=12), qu.CASES["dqn_reference_shape"]),
                     (od.bdq_spec(100, 3, 33, [[64, 64], [32], [32]]), qu.CASES["bdq_reference_shape"])):
        case = qu.make_q_case(**kw)
        eng = qu.QEngine(case["cfg"], backend=NumpyHostBackend(), lib_path=hostemu_lib)
        ref = od.param_shapes(spec)
        assert [t[0] for t in eng.table] == list(ref.keys())
        for name, _, _, shape, _ in eng.table:
            assert tuple(shape) == tuple(ref[name]), name
        eng.close()


def test_prioritised_replay_plan(hostemu_lib):
    """Sampler / priority write-back / add semantics of the device PER against oracle/per.py (CPU, host emulation)."""
    from hostemu_backend import NumpyHostBackend
    qu.per_check(backend=NumpyHostBackend(), lib_path=hostemu_lib)
    qu.per_check(backend=NumpyHostBackend(), lib_path=hostemu_lib, stratified=True)
    qu.per_check(backend=NumpyHostBackend(), lib_path=hostemu_lib, cap=4096, n_store=4096, B=32, n_steps=5, seed=9)
    qu.per_check(backend=NumpyHostBackend(), lib_path=hostemu_lib, cap=2500, n_store=2100, B=64, n_steps=5, seed=5,
                 case_name="bdq_baseline_config3")
    qu.per_check(backend=NumpyHostBackend(), lib_path=hostemu_lib, cap=1024, n_store=1023, B=8, n_steps=5, seed=2)   # leaf size - 2 ends a block


def test_multi_update_per_call_keeps_the_block_sums_current(hostemu_lib, monkeypatch):
    """One grl_train_step_per call of n updates (block sums refreshed by the apply launch) == n single-update calls, on the
    emulation build: the launch sequences of capi.inl / plan_q.inl and the refresh's ownership rule, bit for bit."""
    from hostemu_backend import NumpyHostBackend
    qu.per_multi_update_check(monkeypatch, "bdq", 2100, 2050, backend=NumpyHostBackend(), lib_path=hostemu_lib, n=6)


@pytest.mark.parametrize("name", ["bdq", "dqn"])
def test_multi_update_uniform_call_prefetches_the_next_minibatch(hostemu_lib, monkeypatch, name, capfd):
    """plan_q "q_pf" on the emulation build: one call of n uniform-replay updates == n single-update calls, bit for bit."""
    from hostemu_backend import NumpyHostBackend
    monkeypatch.setenv("GRL_PLAN_DUMP", "1")
    qu.uniform_multi_update_check(monkeypatch, name, 300, backend=NumpyHostBackend(), lib_path=hostemu_lib, n=6)
    assert "q_pf " in capfd.readouterr().err
