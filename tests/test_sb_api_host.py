"""API-conformance test: replays the call sequence of the reference's SAC branch
(/root/reference/manipulation_main/training/sb_helper.py:68-128,175-181,228-247 and
train_stable_baselines.py:75-109 `run`) through the ``stable_baselines`` alias package with a fake
environment.  On CPU the engine behind the model is the TEST-ONLY g++ emulation build (host logic
only); tests/test_gpu_api.py runs the same sequence on the MI355X."""
import os

import numpy as np
import pytest

import stable_baselines as sb
from fake_env import FakeGraspEnv
from grasp_rl.engine import SacEngine
from grasp_rl.sb.sac import SAC
from hostemu_backend import NumpyHostBackend
from stable_baselines.bench import Monitor
from stable_baselines.common.callbacks import BaseCallback, EvalCallback
from stable_baselines.common.vec_env import DummyVecEnv, VecNormalize
from stable_baselines.sac.policies import CnnPolicy as sacCnn
from stable_baselines.sac.policies import MlpPolicy as sacMlp


def create_augmented_nature_cnn(num_direct_features):
    """Same shape as the reference's factory (custom_obs_policy.py:6-44): a closure named
    ``augmented_nature_cnn`` over ``num_direct_features`` -- recognised, never called."""
    def augmented_nature_cnn(scaled_images, **kwargs):
        raise AssertionError("the TF extractor must not be called")
    assert num_direct_features >= 0
    return augmented_nature_cnn if num_direct_features is None else _bind(augmented_nature_cnn, num_direct_features)


def _bind(fn, n):
    def augmented_nature_cnn(scaled_images, **kwargs):
        return fn(scaled_images, n=n, **kwargs)
    return augmented_nature_cnn


class CountingCallback(BaseCallback):
    def __init__(self):
        super().__init__()
        self.steps, self.rollouts, self.started, self.ended = 0, 0, 0, 0

    def _on_training_start(self):
        self.started += 1
        assert "writer" in self.locals and self.model is not None and self.training_env is not None

    def _on_rollout_start(self):
        self.rollouts += 1

    def _on_step(self):
        self.steps += 1
        assert self.training_env.get_attr("curriculum")[0]._lambda == 0.0     # sb_helper.py:42-45
        return True

    def _on_training_end(self):
        self.ended += 1


class SaveVecNormalizeCallback(BaseCallback):        # base_callbacks.py:119-149
    def __init__(self, path):
        super().__init__()
        self.path = path

    def _on_step(self):
        vn = self.model.get_vec_normalize_env()
        assert vn is not None
        vn.save(self.path)
        return True


@pytest.fixture
def emulated_engine(hostemu_lib, monkeypatch):
    monkeypatch.setattr(SAC, "_engine_factory",
                        staticmethod(lambda cfg, device: SacEngine(cfg, backend=NumpyHostBackend(), lib_path=hostemu_lib)))


def run_reference_sequence(tmp_path, policy, make_env, policy_kwargs, batch_size=4, total=24):
    model_dir = str(tmp_path)
    env = DummyVecEnv([lambda: Monitor(make_env(0), os.path.join(model_dir, "log_file"))])
    test_env = DummyVecEnv([lambda: make_env(1)])
    test_env = VecNormalize(test_env, norm_obs=True, norm_reward=False, clip_obs=10.)
    save_vn = SaveVecNormalizeCallback(os.path.join(model_dir, "best_model_vecnormalize.pkl"))
    eval_cb = EvalCallback(test_env, best_model_save_path=os.path.join(model_dir, "best_model"),
                           log_path=os.path.join(model_dir, "best_model", "logs"), eval_freq=10, n_eval_episodes=2,
                           callback_on_new_best=save_vn, deterministic=True, render=False)
    assert env.envs[0].depth_obs in (True, False) and not env.envs[0].is_simplified()      # sb_helper.py:86
    env = VecNormalize(env, norm_obs=True, norm_reward=True, clip_obs=10.)
    model = sb.SAC(policy, env, policy_kwargs=policy_kwargs, verbose=2, gamma=0.99, buffer_size=64,
                   batch_size=batch_size, learning_rate=3e-4, tensorboard_log=None, learning_starts=8)
    counter = CountingCallback()
    p0 = model.get_parameters()
    model.learn(total_timesteps=int(str(total)), callback=[counter, eval_cb])
    assert counter.started == 1 and counter.ended == 1 and counter.steps == total and counter.rollouts >= total
    assert model.num_timesteps == total and model.n_updates == total - max(8, batch_size) + 1   # can_sample + learning_starts
    p1 = model.get_parameters()
    assert any(not np.array_equal(p0[k], p1[k]) for k in p0 if not k.startswith("target"))
    assert os.path.exists(os.path.join(model_dir, "best_model", "best_model.zip"))
    assert os.path.exists(os.path.join(model_dir, "best_model", "logs", "evaluations.npz"))
    assert os.path.exists(os.path.join(model_dir, "best_model_vecnormalize.pkl"))
    ev = np.load(os.path.join(model_dir, "best_model", "logs", "evaluations.npz"))
    assert ev["results"].shape[1] == 2 and list(ev["timesteps"]) == list(range(10, total + 1, 10))
    # SBPolicy.save (sb_helper.py:228-247)
    path = os.path.join(model_dir, "SAC_model")
    model.save(path)
    model.get_vec_normalize_env().save(os.path.join(model_dir, "vecnormalize.pkl"))
    with open(os.path.join(model_dir, "log_file.monitor.csv")) as f:
        lines = f.read().splitlines()
    assert lines[0].startswith("#{") and lines[1] == "r,l,t" and len(lines) >= 4
    # `run` (train_stable_baselines.py:75-109)
    task = DummyVecEnv([lambda: make_env(2)])
    task = VecNormalize(task, training=False, norm_obs=False, norm_reward=False, clip_obs=10.)
    task = VecNormalize.load(os.path.join(model_dir, "vecnormalize.pkl"), task)
    assert task.norm_obs and task.norm_reward and task.obs_rms.count > 20
    agent = sb.SAC.load(path)
    obs = task.reset()
    action = agent.predict(obs, deterministic=True)               # utils.py:71
    assert action[0].shape == (1, 5) and np.all(np.abs(action[0]) <= 1)
    assert np.allclose(action[0], model.predict(obs, deterministic=True)[0], atol=1e-6)
    obs, reward, done, _ = task.step(action[0])
    assert "episode_step" in task.buf_infos[0]                    # utils.py:76
    # --load_dir transfer (sb_helper.py:97-115)
    model2 = sb.SAC(policy, env, policy_kwargs=policy_kwargs, buffer_size=64, batch_size=batch_size)
    model2.load_parameters(sb.SAC.load(path, env).get_parameters(), exact_match=False)
    for k, v in model2.get_parameters().items():
        assert np.array_equal(v, p1[k]), k
    return model


def test_sac_cnn_reference_sequence(tmp_path, emulated_engine):
    kwargs = {"layers": [64, 64], "cnn_extractor": create_augmented_nature_cnn(1)}
    m = run_reference_sequence(tmp_path, sacCnn, lambda s: FakeGraspEnv("depth", seed=s), kwargs)
    assert m.engine.cfg.extractor == 1 and m.engine.cfg.n_direct == 1 and m.engine.cfg.normalize == 1


def test_sac_mlp_reference_sequence(tmp_path, emulated_engine):
    kwargs = {"layers": [64, 64], "layer_norm": False}
    m = run_reference_sequence(tmp_path, sacMlp, lambda s: FakeGraspEnv(seed=s, vector_dim=101), kwargs, batch_size=8)
    assert m.engine.cfg.extractor == 0 and m.engine.cfg.obs_dim == 101


def test_loads_reference_shipped_artifacts(emulated_engine):
    """The zip / pickle the reference ships (extracted to tests/golden by scripts/make_golden.py) in
    stable-baselines' own container format."""
    import io, json, zipfile
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(gold, "sac_mlp_best_model.npz"))
    from grasp_rl.sb import save_util
    from grasp_rl.sb.spaces import Box
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        data = {"gamma": 0.99, "batch_size": 64, "buffer_size": 1000, "policy_kwargs": {"layers": [64, 64], "layer_norm": False},
                "observation_space": Box(-1, 1, shape=(101,)), "action_space": Box(-1, 1, shape=(5,)), "policy": sacMlp,
                "ent_coef": "auto", "n_envs": 1}
        path = save_util.save_to_zip(os.path.join(d, "m"), data, {k: z[k] for k in z.files})
        model = sb.SAC.load(path)
        P = model.get_parameters()
        for k in z.files:
            assert np.array_equal(P[k], z[k]), k
        pins = json.load(open(os.path.join(gold, "oracle_pins.json")))
        vn = np.load(os.path.join(gold, "vecnorm_encoder.npz"))
        obs = np.clip((vn["real_obs"] - vn["mean"]) / np.sqrt(vn["var"] + 1e-8), -10, 10).astype(np.float32)
        a, _ = model.predict(obs, deterministic=True)
        assert np.allclose(a, np.asarray(pins["sac_mlp_real_obs"]["det_action"]), atol=2e-5)


def test_unsupported_paths_fail_loudly(emulated_engine):
    env = DummyVecEnv([lambda: FakeGraspEnv(vector_dim=11)])
    with pytest.raises(NotImplementedError):
        sb.SAC(sacMlp, env, ent_coef=0.1)
    with pytest.raises(NotImplementedError):
        sb.SAC(sacMlp, env, policy_kwargs={"layer_norm": True})
    with pytest.raises(NotImplementedError):
        sb.TRPO(None, env)


# ---------------------------------------------------------------------------------------------- DQN / BDQ
from grasp_rl.engine import QEngine                                     # noqa: E402
from grasp_rl.sb.dqn import BDQ, DQN                                    # noqa: E402
from stable_baselines.bdq.policies import MlpActPolicy                  # noqa: E402
from stable_baselines.deepq.policies import MlpPolicy as DQNMlpPolicy  # noqa: E402


@pytest.fixture
def emulated_q_engine(hostemu_lib, monkeypatch):
    f = staticmethod(lambda cfg, device: QEngine(cfg, backend=NumpyHostBackend(), lib_path=hostemu_lib))
    monkeypatch.setattr(DQN, "_engine_factory", f)
    monkeypatch.setattr(BDQ, "_engine_factory", f)


def run_q_sequence(tmp_path, make_model, make_env, n_steps=40):
    env = DummyVecEnv([lambda: Monitor(make_env(0), os.path.join(str(tmp_path), "log_file"))])
    model = make_model(env)
    counter = CountingCallback()
    p0 = model.get_parameters()
    model.learn(total_timesteps=n_steps, callback=[counter])
    assert counter.steps == n_steps and model.num_timesteps == n_steps
    assert model.n_updates == n_steps - max(model.learning_starts, model.batch_size - 1)
    p1 = model.get_parameters()
    changed = [k for k in p0 if "target_q_func" not in k and "eps" not in k and not np.array_equal(p0[k], p1[k])]
    assert len(changed) > 4
    eps_name = [k for k in p1 if k.endswith("eps:0")][0]
    assert abs(float(p1[eps_name]) - model.exploration_final_eps) < 1e-6           # schedule finished
    tgt = [k for k in p1 if "target_q_func" in k and k.endswith("weights:0")][0]
    assert not np.array_equal(p0[tgt], p1[tgt])                                      # hard update happened
    if n_steps % model.target_network_update_freq == 0:
        # like the reference's final zips (tests/golden/oracle_pins.json: b6_q_zip_relationships): the run ends right
        # after a target update, so target == online bit for bit
        assert all(np.array_equal(p1[k], p1[k.replace("/target_q_func", "")]) for k in p1 if "/target_q_func/" in k)
    path = os.path.join(str(tmp_path), "q_model")
    model.save(path)
    agent = type(model).load(path)
    task = DummyVecEnv([lambda: make_env(1)])
    obs = task.reset()
    action = agent.predict(obs, deterministic=True)
    assert np.array_equal(np.asarray(action[0]), np.asarray(model.predict(obs, deterministic=True)[0]))
    task.step(action[0])
    # sb_helper.py:186-198 / 205-225: transfer only the non-'action_value' layers whose name contains '2'
    usable = {k: v for k, v in agent.get_parameters().items() if "action_value" not in k and "2" in k}
    fresh = make_model(env)
    fresh.load_parameters(usable, exact_match=False)
    for k, v in usable.items():
        assert np.array_equal(fresh.get_parameters()[k], v)
    return model


def test_dqn_reference_sequence(tmp_path, emulated_q_engine):
    make_env = lambda s: FakeGraspEnv(seed=s, vector_dim=20, discrete_actions=12)
    mk = lambda env: sb.DQN(DQNMlpPolicy, env, verbose=2, gamma=0.99, batch_size=8, prioritized_replay=True,
                            tensorboard_log=None, learning_starts=10, target_network_update_freq=10, buffer_size=64)
    m = run_q_sequence(tmp_path, mk, make_env)
    assert m.engine.cfg.algo == 1 and m.engine.cfg.q_huber == 1 and m.engine.cfg.q_bins == 12
    assert [n for n in m.get_parameter_list()][:2] == ["deepq/eps:0", "deepq/model/action_value/fully_connected/weights:0"]


def test_bdq_reference_sequence(tmp_path, emulated_q_engine):
    make_env = lambda s: FakeGraspEnv(seed=s, vector_dim=20, act_dim=3)
    mk = lambda env: sb.BDQ(MlpActPolicy, env, verbose=2, policy_kwargs={"layers": [[16, 16], [8], [8]]}, gamma=0.99,
                            batch_size=8, buffer_size=64, epsilon_greedy=True, exploration_fraction=0.3,
                            exploration_final_eps=0.1, num_actions_pad=5, learning_starts=10,
                            target_network_update_freq=10, prioritized_replay=False, tensorboard_log=None)
    m = run_q_sequence(tmp_path, mk, make_env)
    assert m.engine.cfg.algo == 2 and m.engine.cfg.q_branches == 3 and abs(m.engine.cfg.q_trunk_scale - 0.25) < 1e-7
    a, _ = m.predict(np.zeros(20, np.float32))
    assert a.shape == (3,) and set(np.round((a + 1) * 2, 5)) <= {0.0, 1.0, 2.0, 3.0, 4.0}     # bin centres


def test_continued_training_runs_total_timesteps_more_steps(emulated_engine, emulated_q_engine):
    """stable-baselines loops `for _ in range(total_timesteps)`: `learn(n, reset_num_timesteps=False)` on a trained model
    takes n MORE environment steps (`while num_timesteps < total_timesteps` ran n - num_timesteps, i.e. none)."""
    env = DummyVecEnv([lambda: FakeGraspEnv(seed=0, vector_dim=20, discrete_actions=12)])
    q = sb.DQN(DQNMlpPolicy, env, batch_size=8, learning_starts=10, target_network_update_freq=10, buffer_size=64)
    c = CountingCallback()
    q.learn(total_timesteps=24, callback=c)
    assert q.num_timesteps == 24 and c.steps == 24
    q.learn(total_timesteps=24, callback=c, reset_num_timesteps=False)
    assert q.num_timesteps == 48 and c.steps == 48
    q.learn(total_timesteps=8, callback=c)                       # default: the counter starts over
    assert q.num_timesteps == 8 and c.steps == 56
    env = DummyVecEnv([lambda: FakeGraspEnv(seed=0, vector_dim=20)])
    m = sb.SAC(sacMlp, env, batch_size=4, buffer_size=64, learning_starts=4, policy_kwargs={"layers": [16, 16]})
    c = CountingCallback()
    m.learn(total_timesteps=12, callback=c)
    m.learn(total_timesteps=12, callback=c, reset_num_timesteps=False)
    assert m.num_timesteps == 24 and c.steps == 24


def test_vecnormalize_flags_are_honoured_separately(emulated_engine, emulated_q_engine):
    """VecNormalize(norm_obs=, norm_reward=) each reach the device gather on their own (stable-baselines honours
    them separately at sample time): the engine's `normalize` mode is 1 both / 2 observations / 3 rewards / 0."""
    for flags, mode in (((True, True), 1), ((True, False), 2), ((False, True), 3), ((False, False), 0)):
        mk = lambda: VecNormalize(DummyVecEnv([lambda: FakeGraspEnv(seed=0, vector_dim=20, discrete_actions=6)]),
                                  norm_obs=flags[0], norm_reward=flags[1])
        m = sb.DQN(DQNMlpPolicy, mk(), batch_size=4, buffer_size=32, learning_starts=5)
        assert m.engine.cfg.normalize == mode, (flags, m.engine.cfg.normalize)
        m.learn(total_timesteps=12)
        env = VecNormalize(DummyVecEnv([lambda: FakeGraspEnv(seed=0, vector_dim=20)]), norm_obs=flags[0], norm_reward=flags[1])
        s = sb.SAC(sacMlp, env, batch_size=4, buffer_size=32, learning_starts=5)
        assert s.engine.cfg.normalize == mode
        s.learn(total_timesteps=12)
        if mode in (2, 3):        # what the update saw: normalised observations xor normalised rewards
            rew = s.engine.fetch("rew", (4,))
            raw_scale = np.abs(rew).max() > 10.0 + 1e-6          # raw rewards of the fake env are ~ -200
            assert raw_scale == (mode == 2)


def test_learning_rate_schedule_is_evaluated_before_every_update(emulated_engine):
    """A callable `learning_rate` is SB's schedule lr(progress_remaining): re-evaluated for each update (the step
    size lives in device memory, grl_set_learning_rate), not frozen at construction."""
    seen = []

    def schedule(frac):
        seen.append(frac)
        return 3e-4 * frac
    env = DummyVecEnv([lambda: FakeGraspEnv(seed=0, vector_dim=11)])
    m = sb.SAC(sacMlp, env, batch_size=4, buffer_size=32, learning_starts=4, learning_rate=schedule)
    seen.clear()
    m.learn(total_timesteps=20)
    assert len(seen) == m.n_updates > 10 and seen[0] > seen[-1] > 0.0
    before = m.get_parameters()
    m.learning_rate = lambda frac: 0.0            # a zero step size must freeze the weights (Adam moments still move)
    m.learn(total_timesteps=8)
    after = m.get_parameters()
    assert all(np.array_equal(before[k], after[k]) for k in before if k.startswith("model/pi/"))


def test_untrusted_pickles_cannot_name_arbitrary_callables(tmp_path, monkeypatch):
    """vecnormalize.pkl / zip members come from elsewhere: only NumPy, plain containers and the mapped classes
    may be constructed; a pickle that names os.system (or any other global) is refused."""
    import pickle

    class Evil:
        def __reduce__(self):
            import os as _os
            return (_os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))
    with open(tmp_path / "vecnormalize.pkl", "wb") as f:
        pickle.dump(Evil(), f)
    venv = DummyVecEnv([lambda: FakeGraspEnv(seed=0, vector_dim=11)])
    with pytest.raises(pickle.UnpicklingError):
        VecNormalize.load(str(tmp_path / "vecnormalize.pkl"), venv)
    assert not (tmp_path / "pwned").exists()
    from grasp_rl.sb import save_util
    import base64, json
    blob = json.dumps({"gamma": 0.9, "policy_kwargs": {":type:": "<class 'dict'>",
                                                        ":serialized:": base64.b64encode(pickle.dumps(Evil())).decode()}})
    data = save_util.json_to_data(blob)                       # refused -> rebuilt from readable attributes -> None
    assert data["gamma"] == 0.9 and data["policy_kwargs"] is None and not (tmp_path / "pwned").exists()

    # globals inside the numpy package are allowed one by one, not by prefix: numpy.testing's exec wrapper is refused
    class EvilNumpy:
        def __reduce__(self):
            from numpy.testing._private.utils import runstring
            return (runstring, ("open(%r, 'w').write('x')" % str(tmp_path / "pwned2"), {}))
    with open(tmp_path / "vn2.pkl", "wb") as f:
        pickle.dump(EvilNumpy(), f)
    with pytest.raises(pickle.UnpicklingError):
        VecNormalize.load(str(tmp_path / "vn2.pkl"), venv)
    assert not (tmp_path / "pwned2").exists()


def test_extractor_is_inferred_when_the_closure_is_not_unpickled(tmp_path, emulated_engine):
    """A zip saved with the reference's cloudpickled `cnn_extractor` closure loads without executing the pickle:
    extractor / direct features / layers come from the parameter names and shapes."""
    env = DummyVecEnv([lambda: FakeGraspEnv("depth", seed=0)])
    kwargs = {"layers": [32, 48], "cnn_extractor": create_augmented_nature_cnn(1)}
    m = sb.SAC(sacCnn, env, policy_kwargs=kwargs, buffer_size=16, batch_size=2)
    m.save(str(tmp_path / "m"))
    m2 = sb.SAC.load(str(tmp_path / "m"))
    c = m2.engine.cfg
    assert (c.extractor, c.n_direct, c.n_layers, c.layers[0], c.layers[1]) == (1, 1, 2, 32, 48)
    m2.save(str(tmp_path / "m2"))                              # and the stand-in extractor object round-trips
    assert sb.SAC.load(str(tmp_path / "m2")).engine.cfg.extractor == 1


def test_monitor_logs_are_readable_by_results_plotter(tmp_path):
    """scripts/plot.py:7,64,76 of the reference: `load_results(folder)` + `ts2xy(result, 'timesteps')` on the CSV
    our Monitor writes, and on a log the reference ships (its fork adds columns)."""
    from stable_baselines.results_plotter import load_results, ts2xy
    env = Monitor(FakeGraspEnv(seed=0, vector_dim=10, episode_len=4), os.path.join(str(tmp_path), "log_file"))
    for _ in range(3):
        env.reset()
        done = False
        while not done:
            _, _, done, _ = env.step(np.zeros(5, np.float32))
    env.close()
    df = load_results(str(tmp_path))
    assert list(df.columns[:3]) == ["r", "l", "t"] and len(df) == 3 and np.all(df.l.values == 4)
    x, y = ts2xy(df, "timesteps")
    assert list(x) == [4, 8, 12] and len(y) == 3
    shipped = "/root/reference/trained_models/SAC_depth_1mbuffer"
    if os.path.isdir(shipped):
        ref = load_results(shipped)
        assert {"r", "l", "t", "s"} <= set(ref.columns) and len(ref) > 1000
        xs, ys = ts2xy(ref, "timesteps", y_column="s")
        assert np.all(np.diff(xs) > 0) and ys.min() >= 0.0 and ys.max() <= 1.0        # running success rate


def test_running_statistics_on_the_device_equal_the_host_wrapper(hostemu_lib, emulated_engine):
    """grl_norm_update (RunningMeanStd.update on the device) against the host wrapper's NumPy restatement, bit for bit:
    float32 batch moments, float64 Chan merge, count; then the raw-observation act path (grl_act flag 2) against acting
    on host-normalised observations."""
    from grasp_rl import _capi
    rng = np.random.default_rng(3)
    for obs_shape, extractor in (((64, 64, 2), "augmented"), ((37,), "mlp")):
        if extractor == "mlp":
            cfg = _capi.make_config("mlp", obs_dim=37, act_dim=5, layers=(64, 64), batch_size=4, replay_capacity=8,
                                    normalize=True, act_batch=6)
        else:
            cfg = _capi.make_config("augmented", obs_channels=2, n_direct=1, act_dim=5, layers=(64, 64), batch_size=4,
                                    replay_capacity=8, normalize=True, act_batch=6)
        eng = SacEngine(cfg, backend=NumpyHostBackend(), lib_path=hostemu_lib)
        from grasp_rl.init import init_parameters
        eng.set_parameters(init_parameters(eng.table, seed=1))
        from grasp_rl.sb.running_mean_std import RunningMeanStd
        rms = RunningMeanStd(shape=obs_shape)
        for n in (6, 1, 5, 6):
            x = (rng.normal(0.4, 0.3, (n,) + obs_shape) * rng.uniform(0.5, 2.0, obs_shape)).astype(np.float32)
            if extractor != "mlp":
                x[..., 1] = 0.0
                x[:, 0, 0, 1] = rng.uniform(0, 1, n)
            rms.update(x)
            eng.norm_update(x)
            mean, var, count = eng.get_obs_stats(obs_shape)
            assert count == rms.count
            assert np.array_equal(mean, rms.mean) and np.array_equal(var, rms.var)
        # acting on raw observations == acting on clip((obs - mean) / sqrt(var + eps), +-10) formed by the host
        x = (rng.normal(0.4, 0.3, (4,) + obs_shape)).astype(np.float32)
        xn = np.clip((x - rms.mean) / np.sqrt(rms.var + 1e-8), -10.0, 10.0)
        a_host = eng.act(xn.astype(np.float32), deterministic=True)
        a_dev = eng.act(x, deterministic=True, raw=True)
        assert np.array_equal(a_host, a_dev)
        # a loaded pickle's statistics continue on the device
        eng.set_obs_stats(rms.mean * 0.5, rms.var * 2.0, 3.0)
        eng.set_running_stats(rms.mean * 0.5, rms.var * 2.0, 123.0)
        rms.mean, rms.var, rms.count = rms.mean * 0.5, rms.var * 2.0, 123.0
        x = rng.normal(0.2, 0.5, (3,) + obs_shape).astype(np.float32)
        rms.update(x)
        eng.norm_update(x)
        mean, var, count = eng.get_obs_stats(obs_shape)
        assert count == rms.count and np.array_equal(mean, rms.mean) and np.array_equal(var, rms.var)
        eng.close()


def test_learning_with_device_statistics_equals_the_host_path(tmp_path, emulated_engine):
    """SAC(device_norm=True): the wrapper hands out raw observations while learning, statistics / normalisation run in
    the engine -- parameters after learn() and the pickled statistics equal the default (host) path bit for bit."""
    outs = []
    for dev in (False, True):
        env = DummyVecEnv([(lambda s=s: FakeGraspEnv("depth", seed=s)) for s in range(3)])
        env = VecNormalize(env, norm_obs=True, norm_reward=True, clip_obs=10.)
        m = sb.SAC(sacCnn, env, policy_kwargs={"layers": [64, 64], "cnn_extractor": create_augmented_nature_cnn(1)},
                   buffer_size=64, batch_size=4, learning_starts=6, seed=5, device_norm=dev)
        seen = []

        class Peek(BaseCallback):
            def _on_step(self):
                seen.append(float(np.abs(self.locals["new_obs"]).max()))
                return True
        m.learn(total_timesteps=30, callback=Peek())
        env.save(str(tmp_path / ("vn%d.pkl" % dev)))
        outs.append((m.get_parameters(), env.obs_rms.mean.copy(), env.obs_rms.var.copy(), env.obs_rms.count, seen))
        assert env._dev is None                                  # detached again: the wrapper normalises on the host
        assert np.abs(env.reset()).max() <= 10.0
    (Pa, ma, va, ca, _), (Pb, mb, vb, cb, _) = outs
    assert ca == cb and np.array_equal(ma, mb) and np.array_equal(va, vb)
    assert all(np.array_equal(Pa[k], Pb[k]) for k in Pa)
    vn = VecNormalize.load(str(tmp_path / "vn1.pkl"), DummyVecEnv([lambda: FakeGraspEnv("depth", seed=9)]))
    assert np.array_equal(vn.obs_rms.mean, ma) and vn.obs_rms.count == ca


@pytest.mark.parametrize("extractor,channels,rgb_u8", [("augmented", 2, False), ("augmented", 5, True), ("mlp", 0, False)])
def test_observations_uploaded_once_leave_the_engine_in_the_same_state(hostemu_lib, extractor, channels, rgb_u8):
    """grl_observe + grl_act(GRL_ACT_OBSERVED) + grl_replay_add_observed against the separate uploads (emulation build;
    tests/test_gpu_api.py runs the same body on the MI355X)."""
    import observed_util
    observed_util.check_observed_path(lambda cfg: SacEngine(cfg, backend=NumpyHostBackend(), lib_path=hostemu_lib),
                                      extractor, channels, rgb_u8=rgb_u8)
