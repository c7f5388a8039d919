"""The reference's OWN training driver on top of this package.

`/root/reference/manipulation_main/training/sb_helper.py` (class SBPolicy: policy selection :85-96, model
construction :120-128, callbacks :70-84, `model.learn` :175-177, `save` :228-247) together with its
`base_callbacks.py` and `custom_obs_policy.py` is imported from where it lies -- unmodified, nothing copied --
with `stable_baselines` resolving to the alias package of this repository and two stub modules for what the
container lacks (`tensorflow`: only attribute access at import time, the extractor closure is never called;
`gym`: the `Env` name used in type annotations).  The engine behind the model is the TEST-ONLY g++ emulation
build, the environment is tests/fake_env.py.  Skipped where /root/reference does not exist (the GPU box)."""
import importlib
import os
import sys
import types

import numpy as np
import pytest

import stable_baselines as sb
from fake_env import FakeGraspEnv
from grasp_rl.engine import QEngine, SacEngine
from grasp_rl.sb import spaces
from grasp_rl.sb.dqn import DQN
from grasp_rl.sb.sac import SAC
from hostemu_backend import NumpyHostBackend
from stable_baselines.bench import Monitor
from stable_baselines.common.vec_env import DummyVecEnv, VecNormalize

REF_TRAINING = "/root/reference/manipulation_main/training"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF_TRAINING, "sb_helper.py")),
                                reason="/root/reference is not present on this box")


@pytest.fixture
def reference_sb_helper(monkeypatch, hostemu_lib):
    tf = types.ModuleType("tensorflow")
    tf.nn = types.SimpleNamespace(relu=lambda x: x)
    tf.contrib = types.SimpleNamespace()
    tf.Summary = type("Summary", (), {"Value": staticmethod(lambda **k: k), "__init__": lambda self, **k: self.__dict__.update(k)})
    gym = types.ModuleType("gym")
    gym.Env = type("Env", (), {})
    gym.spaces = spaces
    monkeypatch.setitem(sys.modules, "tensorflow", tf)
    monkeypatch.setitem(sys.modules, "gym", gym)
    monkeypatch.syspath_prepend(REF_TRAINING)
    for name in ("sb_helper", "base_callbacks", "custom_obs_policy"):
        monkeypatch.delitem(sys.modules, name, raising=False)
    monkeypatch.setattr(SAC, "_engine_factory",
                        staticmethod(lambda cfg, device: SacEngine(cfg, backend=NumpyHostBackend(), lib_path=hostemu_lib)))
    monkeypatch.setattr(DQN, "_engine_factory",
                        staticmethod(lambda cfg, device: QEngine(cfg, backend=NumpyHostBackend(), lib_path=hostemu_lib)))
    mod = importlib.import_module("sb_helper")
    assert os.path.realpath(mod.__file__).startswith("/root/reference/")
    yield mod
    for name in ("sb_helper", "base_callbacks", "custom_obs_policy"):
        sys.modules.pop(name, None)


def test_reference_sbpolicy_trains_saves_and_reloads(reference_sb_helper, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    os.makedirs("models/run")
    config = {"normalize": True, "discount_factor": 0.99,
              "SAC": {"tensorboard_logs": None, "layers": [64, 64], "buffer_size": 256, "batch_size": 4,
                      "step_size": 3e-4, "total_timesteps": 112}}
    env = DummyVecEnv([lambda: Monitor(FakeGraspEnv("depth", seed=0), os.path.join("models/run", "log_file"))])
    test_env = DummyVecEnv([lambda: FakeGraspEnv("depth", seed=1)])
    driver = reference_sb_helper.SBPolicy(env, test_env, config, "models/run", algo="SAC")
    driver.learn()                                            # the reference's code path end to end
    assert os.path.isfile("models/run/run.zip") and os.path.isfile("models/run/vecnormalize.pkl")
    assert os.path.isdir("models/run/best_model")             # EvalCallback._init_callback of base_callbacks.py
    # what train_stable_baselines.py `run` does with the result (:86-109)
    venv = VecNormalize.load("models/run/vecnormalize.pkl", DummyVecEnv([lambda: FakeGraspEnv("depth", seed=2)]))
    venv.training = False
    model = sb.SAC.load("models/run/run.zip")
    assert model.engine.cfg.extractor == 1                    # custom_obs_policy.create_augmented_nature_cnn(1) was recognised
    obs = venv.reset()
    action, _ = model.predict(obs, deterministic=True)
    assert action.shape == (1, 5) and np.all(np.isfinite(action)) and np.all(np.abs(action) <= 1)
    P = model.get_parameters()
    assert "model/pi/c1/w:0" in P or any(k.startswith("model/pi/") for k in P)
    assert venv.obs_rms.count > 100                           # statistics gathered during the reference's learn()


def test_reference_sbpolicy_load_dir_transfer_and_tensorboard(reference_sb_helper, tmp_path, monkeypatch):
    """sb_helper.py:97-115, the `--load_dir` branch: `VecNormalize(env, training=True, norm_obs=False,
    norm_reward=False)` -> `VecNormalize.load(<top folder>/vecnormalize.pkl, env)` -> `sb.SAC(...)` ->
    `sb.SAC.load(load_dir, env)` -> `get_parameters()` -> `load_parameters(params, exact_match=False)`; with
    `tensorboard_logs` set (config/full_depth_obs.yaml:65) so that the reference's TensorboardCallback reaches
    `self.locals['writer'].add_summary(...)` (sb_helper.py:50-52)."""
    monkeypatch.chdir(tmp_path)
    os.makedirs("models/first")
    os.makedirs("models/second")
    sac = {"tensorboard_logs": None, "layers": [64, 64], "buffer_size": 256, "batch_size": 4, "step_size": 3e-4,
           "total_timesteps": 108}
    mk = lambda d, s: DummyVecEnv([lambda: Monitor(FakeGraspEnv("depth", seed=s), os.path.join(d, "log_file"))])
    first = reference_sb_helper.SBPolicy(mk("models/first", 0), DummyVecEnv([lambda: FakeGraspEnv("depth", seed=1)]),
                                         {"normalize": True, "discount_factor": 0.99, "SAC": dict(sac)}, "models/first",
                                         algo="SAC")
    first.learn()
    P_first = sb.SAC.load("models/first/first.zip").get_parameters()
    cfg2 = {"normalize": True, "discount_factor": 0.99, "SAC": dict(sac, tensorboard_logs=True, total_timesteps=104)}
    second = reference_sb_helper.SBPolicy(mk("models/second", 2), DummyVecEnv([lambda: FakeGraspEnv("depth", seed=3)]),
                                          cfg2, "models/second", load_dir="models/first/first.zip", algo="SAC")
    seen = {}
    real_load_parameters = SAC.load_parameters

    def spy(self, params, exact_match=True):
        seen["exact_match"], seen["n"] = exact_match, len(params)
        real_load_parameters(self, params, exact_match=exact_match)
        seen["after"] = self.get_parameters()
    monkeypatch.setattr(SAC, "load_parameters", spy)
    second.learn()
    assert seen["exact_match"] is False and seen["n"] == len(P_first)
    for k, v in P_first.items():                               # the transferred weights are the trained ones ...
        assert np.array_equal(seen["after"][k], v), k
    vn = second.env                                            # ... and the statistics of run 1 carried over
    assert isinstance(vn, VecNormalize) and vn.norm_obs and vn.norm_reward and isinstance(vn.venv, VecNormalize)
    assert not vn.venv.norm_obs and not vn.venv.norm_reward
    assert vn.obs_rms.count > 200                              # 108 steps of run 1 + 104 of run 2
    assert os.path.isfile("models/second/second.zip") and os.path.isfile("models/second/vecnormalize.pkl")
    rows = open("tensorboard_logs/models/second/SAC_1/scalars.csv").read().strip().splitlines()
    assert rows[0] == "step,tag,value" and len(rows) > 50 and all(r.split(",")[1] == "success_rate" for r in rows[1:])


def test_reference_callbacks_fire_on_evaluation(reference_sb_helper, tmp_path, monkeypatch):
    """base_callbacks.py of the reference: its EvalCallback (:16-117) with the SaveVecNormalizeCallback (:119-149) as
    `callback_on_new_best` and the TrainingTimeCallback, with an evaluation period short enough to trigger:
    sync_envs_normalization, evaluate_policy(return_episode_rewards=True), best-model zip, evaluations.npz."""
    bc = sys.modules["base_callbacks"]
    monkeypatch.chdir(tmp_path)
    mk = lambda s: FakeGraspEnv(seed=s, vector_dim=101, episode_len=5)
    env = VecNormalize(DummyVecEnv([lambda: mk(0)]), norm_obs=True, norm_reward=True, clip_obs=10.)
    test_env = VecNormalize(DummyVecEnv([lambda: mk(1)]), norm_obs=True, norm_reward=False, clip_obs=10.)
    save_vn = bc.SaveVecNormalizeCallback(save_freq=1, save_path="best")
    evalcb = bc.EvalCallback(test_env, best_model_save_path="best", log_path="best/logs", eval_freq=30, n_eval_episodes=2,
                             callback_on_new_best=save_vn, deterministic=True, render=False)
    model = sb.SAC(reference_sb_helper.sacMlp, env, policy_kwargs={"layers": [64, 64], "layer_norm": False},
                   buffer_size=128, batch_size=8, learning_starts=10)
    model.learn(total_timesteps=70, callback=[evalcb, bc.TrainingTimeCallback()])
    assert os.path.isfile("best/best_model.zip") and os.path.isfile("best/vecnormalize.pkl")
    ev = np.load("best/logs/evaluations.npz")
    assert list(ev["timesteps"]) == [30, 60] and ev["results"].shape == (2, 2) and np.all(ev["ep_lengths"] == 5)
    assert test_env.obs_rms.count > 60                                    # statistics copied over by sync_envs_normalization


def test_reference_sbpolicy_dqn_branch(reference_sb_helper, tmp_path, monkeypatch):
    """sb_helper.py:155-165: `sb.DQN(DQNMlpPolicy, env, verbose, gamma, batch_size, prioritized_replay, tensorboard_log)`
    on the auto-encoder-feature observation (100-d) with 12 discrete actions, prioritised replay on."""
    monkeypatch.chdir(tmp_path)
    os.makedirs("models/dqn")
    config = {"normalize": False, "discount_factor": 0.99,
              "DQN": {"tensorboard_logs": None, "batch_size": 8, "prioritized_replay": True, "total_timesteps": 1100}}
    mk = lambda s: FakeGraspEnv(seed=s, vector_dim=100, discrete_actions=12)
    env = DummyVecEnv([lambda: Monitor(mk(0), os.path.join("models/dqn", "log_file"))])
    driver = reference_sb_helper.SBPolicy(env, DummyVecEnv([lambda: mk(1)]), config, "models/dqn", algo="DQN")
    driver.learn()
    assert os.path.isfile("models/dqn/dqn.zip")
    model = sb.DQN.load("models/dqn/dqn.zip")
    a, _ = model.predict(np.zeros((1, 100), np.float32), deterministic=True)
    assert a.shape == (1,) and 0 <= int(a[0]) < 12
    assert any("action_value" in k for k in model.get_parameters())       # the names sb_helper.load_params filters on (:190-193)


def test_reference_run_agent_evaluation_loop(reference_sb_helper, tmp_path, monkeypatch):
    """`manipulation_main/utils.py: run_agent` (:10-44, the loop behind `train_stable_baselines.py run`, :106-109):
    `agent.predict(obs, deterministic=)`, `task.step(action[0])`, `task.buf_infos[0][...]`."""
    import enum
    import importlib.util
    robot = types.ModuleType("manipulation_main.gripperEnv.robot")
    robot.RobotEnv = type("RobotEnv", (), {"Status": enum.IntEnum("Status", {"RUNNING": 0, "SUCCESS": 1})})
    for name, mod in (("manipulation_main", types.ModuleType("manipulation_main")),
                      ("manipulation_main.gripperEnv", types.ModuleType("manipulation_main.gripperEnv")),
                      ("manipulation_main.gripperEnv.robot", robot)):
        monkeypatch.setitem(sys.modules, name, mod)
    spec = importlib.util.spec_from_file_location("reference_utils", "/root/reference/manipulation_main/utils.py")
    utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(utils)
    task = DummyVecEnv([lambda: FakeGraspEnv("depth", seed=3, episode_len=5)])
    model = sb.SAC(reference_sb_helper.sacCnn, task, policy_kwargs={
        "layers": [64, 64], "cnn_extractor": reference_sb_helper.custom_obs_policy.create_augmented_nature_cnn(1)},
        buffer_size=16, batch_size=4)
    rewards, steps, success, timings = utils.run_agent(task, model, n_episodes=3)
    assert rewards.shape == (3,) and np.all(steps == 5) and np.all(success == 1) and np.all(np.isfinite(rewards))


def test_reference_train_script_train_then_run(reference_sb_helper, tmp_path, monkeypatch):
    """`train_stable_baselines.py`: its `train(args)` (:26-75) and `run(args)` (:77-109) functions, imported from
    the reference and called as its `__main__` would, on the reference's own `config/gripper_grasp.yaml`
    (replay / batch size reduced for the CPU emulation; `gym.make('gripper-env-v0', ...)` returns the fake env)."""
    import enum
    import importlib.util
    import yaml

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    robot = types.ModuleType("manipulation_main.gripperEnv.robot")
    robot.RobotEnv = type("RobotEnv", (), {"Status": enum.IntEnum("Status", {"RUNNING": 0, "SUCCESS": 1})})
    wrapper = types.ModuleType("manipulation_main.training.wrapper")
    wrapper.TimeFeatureWrapper = lambda env: env
    pkgs = {n: types.ModuleType(n) for n in ("manipulation_main", "manipulation_main.gripperEnv", "manipulation_main.training",
                                             "manipulation_main.common")}
    for m in pkgs.values():
        m.__path__ = []
    pkgs.update({"manipulation_main.gripperEnv.robot": robot, "manipulation_main.training.wrapper": wrapper})
    for n, m in pkgs.items():
        monkeypatch.setitem(sys.modules, n, m)
    real_yaml_load = yaml.load                                  # the reference targets PyYAML < 6: `yaml.load(f)` without a Loader
    monkeypatch.setattr(yaml, "load", lambda f, Loader=None: real_yaml_load(f, Loader=Loader or yaml.FullLoader))
    io_utils = load("manipulation_main.common.io_utils", "/root/reference/manipulation_main/common/io_utils.py")
    monkeypatch.setitem(sys.modules, "manipulation_main.common.io_utils", io_utils)
    pkgs["manipulation_main.common"].io_utils = io_utils
    utils = load("manipulation_main.utils", "/root/reference/manipulation_main/utils.py")
    monkeypatch.setitem(sys.modules, "manipulation_main.utils", utils)
    made = []

    def make(name, config=None, **kw):
        assert name == "gripper-env-v0" and config["robot"]["discrete"] is False
        made.append(kw)
        return FakeGraspEnv("depth", seed=len(made), episode_len=5)
    sys.modules["gym"].make = make
    monkeypatch.delitem(sys.modules, "train_stable_baselines", raising=False)
    script = importlib.import_module("train_stable_baselines")
    assert os.path.realpath(script.__file__).startswith("/root/reference/")

    with open("/root/reference/config/gripper_grasp.yaml") as f:
        cfg = yaml.safe_load(f)
    assert cfg["SAC"]["buffer_size"] == 1000000 and cfg["normalize"] is True        # the reference's values ...
    cfg["SAC"]["buffer_size"], cfg["SAC"]["batch_size"] = 256, 4                    # ... scaled for the emulated engine
    monkeypatch.chdir(tmp_path)
    os.makedirs("trained")
    with open("cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    args = types.SimpleNamespace(config="cfg.yaml", model_dir="trained/sac_test", algo="SAC", load_dir=None, timestep="112",
                                 simple=False, shaped=False, visualize=False, timefeature=False)
    script.train(args)
    assert os.path.isfile("trained/sac_test/sac_test.zip") and os.path.isfile("trained/sac_test/vecnormalize.pkl")
    assert os.path.isfile("trained/sac_test/config.yaml") and os.path.isfile("trained/sac_test/best_model/config.yaml")
    calls = []
    full_run_agent = utils.run_agent

    def run_few(task, agent, stochastic=False):               # the script asks for 100 episodes; four are enough here
        calls.append(1)
        return full_run_agent(task, agent, stochastic, n_episodes=4)
    monkeypatch.setattr(script, "run_agent", run_few)
    script.run(types.SimpleNamespace(model="trained/sac_test/sac_test.zip", visualize=False, test=True, stochastic=False))
    assert calls == [1] and len(made) == 3                    # train env, eval env, run env
    sys.modules.pop("train_stable_baselines", None)


def test_reference_train_script_fans_out_with_grl_num_envs(reference_sb_helper, tmp_path, monkeypatch):
    """Row J3: the reference's OWN `train(args)` with GRL_NUM_ENVS=4 in the environment.  Its
    `DummyVecEnv([lambda: Monitor(gym.make(...), model_dir/log_file)])` (train_stable_baselines.py:52-54) is answered as it
    always was (`env.envs[0].is_simplified()`, sb_helper.py:86) and, when sb_helper hands it to `sb.SAC(...)` as the training
    env, becomes 4 worker processes each calling that same lambda: 4 monitor files, `get_attr('history')` (sb_helper.py:42-45,
    the reference's TensorboardCallback) served by the workers, one update per environment step, evaluation env single."""
    import enum
    import importlib.util
    import yaml

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    robot = types.ModuleType("manipulation_main.gripperEnv.robot")
    robot.RobotEnv = type("RobotEnv", (), {"Status": enum.IntEnum("Status", {"RUNNING": 0, "SUCCESS": 1})})
    wrapper = types.ModuleType("manipulation_main.training.wrapper")
    wrapper.TimeFeatureWrapper = lambda env: env
    pkgs = {n: types.ModuleType(n) for n in ("manipulation_main", "manipulation_main.gripperEnv", "manipulation_main.training",
                                             "manipulation_main.common")}
    for m in pkgs.values():
        m.__path__ = []
    pkgs.update({"manipulation_main.gripperEnv.robot": robot, "manipulation_main.training.wrapper": wrapper})
    for n, m in pkgs.items():
        monkeypatch.setitem(sys.modules, n, m)
    real_yaml_load = yaml.load
    monkeypatch.setattr(yaml, "load", lambda f, Loader=None: real_yaml_load(f, Loader=Loader or yaml.FullLoader))
    io_utils = load("manipulation_main.common.io_utils", "/root/reference/manipulation_main/common/io_utils.py")
    monkeypatch.setitem(sys.modules, "manipulation_main.common.io_utils", io_utils)
    pkgs["manipulation_main.common"].io_utils = io_utils
    utils = load("manipulation_main.utils", "/root/reference/manipulation_main/utils.py")
    monkeypatch.setitem(sys.modules, "manipulation_main.utils", utils)
    made = []

    def make(name, config=None, **kw):
        made.append(os.getpid())
        return FakeGraspEnv("depth", seed=len(made) + 17 * (os.getpid() % 1000), episode_len=5)
    sys.modules["gym"].make = make
    monkeypatch.delitem(sys.modules, "train_stable_baselines", raising=False)
    script = importlib.import_module("train_stable_baselines")
    assert os.path.realpath(script.__file__).startswith("/root/reference/")
    with open("/root/reference/config/gripper_grasp.yaml") as f:
        cfg = yaml.safe_load(f)
    cfg["SAC"]["buffer_size"], cfg["SAC"]["batch_size"] = 256, 4
    monkeypatch.chdir(tmp_path)
    os.makedirs("trained")
    with open("cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    # the workers are forked here: the lambda's globals (the stub `gym`, FakeGraspEnv) exist only in this test process, and the
    # emulation engine has no HIP context a fork could damage (the product default is forkserver)
    monkeypatch.setenv("GRL_NUM_ENVS", "4")
    monkeypatch.setenv("GRL_ENV_START_METHOD", "fork")
    seen = {}
    real_learn = SAC.learn

    def spy(self, total_timesteps, callback=None, **kw):
        env = self.env
        seen["n_envs"], seen["num_envs"] = self.n_envs, env.num_envs
        seen["history"] = env.get_attr("history")
        seen["act_batch"] = self.engine.cfg.act_batch
        out = real_learn(self, total_timesteps, callback=callback, **kw)
        seen["updates"], seen["steps"] = self.n_updates, self.num_timesteps
        seen["ret_shape"] = self.get_vec_normalize_env().ret.shape
        seen["sr"] = env.get_attr("sr_mean")
        return out
    monkeypatch.setattr(SAC, "learn", spy)
    args = types.SimpleNamespace(config="cfg.yaml", model_dir="trained/fan", algo="SAC", load_dir=None, timestep="120",
                                 simple=False, shaped=False, visualize=False, timefeature=False)
    script.train(args)
    assert seen["n_envs"] == seen["num_envs"] == seen["act_batch"] == 4 and seen["ret_shape"] == (4,)
    assert seen["history"] == [[], [], [], []] and seen["sr"] == [1.0] * 4          # live attributes of the WORKERS' envs
    assert seen["steps"] == 120 and seen["updates"] == 120 - 100 + 4                # one update per environment step after learning_starts
    files = sorted(f for f in os.listdir("trained/fan") if f.endswith("monitor.csv"))
    assert files == ["log_file.env1.monitor.csv", "log_file.env2.monitor.csv", "log_file.env3.monitor.csv", "log_file.monitor.csv"]
    rows = [open(os.path.join("trained/fan", f)).read().strip().splitlines() for f in files]
    assert all(len(r) == 2 + 6 for r in rows)                                      # header, column names, 30 steps / 5 per episode
    from stable_baselines.results_plotter import load_results
    assert len(load_results("trained/fan")) == 24
    assert os.path.isfile("trained/fan/fan.zip") and os.path.isfile("trained/fan/vecnormalize.pkl")
    # train env template + evaluation env were built in THIS process, the four training envs in four others
    here = os.getpid()
    assert made.count(here) == 2
    sys.modules.pop("train_stable_baselines", None)


class _EncodedObsEnv(FakeGraspEnv):
    """The reference env's observation with `depth_observation: False` (robot.py:83-89,185-190): `np.append` over
    `[EncodedDepthImgSensor, actuator]` -- with the reference's OWN sensor class (sensor.py:176-222, imported from where it
    lies) around a fake camera that cycles through the six real depth frames of tests/golden."""

    def __init__(self, sensor_mod, config, seed=0, episode_len=5):
        super().__init__(seed=seed, vector_dim=101, episode_len=episode_len)
        frames = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "depth_frames.npz"))["frames"]
        self._frames = frames.astype(np.float32).reshape(-1, 64, 64)
        self._k = seed

        class Camera:
            def get_state(cam):
                self._k += 1
                img = self._frames[self._k % len(self._frames)].copy()
                return None, img, np.full(img.shape, 7, np.int32)          # (no pixel belongs to plane / robot / table / tray)
        self.robot_id = 3
        self.sensor = sensor_mod.EncodedDepthImgSensor(config, Camera(), self)
        self.depth_obs = self.full_obs = False
        dim = int(np.prod(self.sensor.state_space.shape)) + 1
        self.observation_space = spaces.Box(-np.inf, np.inf, shape=(dim,), dtype=np.float32)

    def is_simplified(self):
        return True

    def _obs(self):          # (FakeGraspEnv.reset / step hand out what this returns)
        return np.append(self.sensor.get_state(), 0.05 * (self.episode_step % 3)).astype(np.float32)


def test_reference_train_script_batches_the_auto_encoder_under_grl_num_envs(reference_sb_helper, hostemu_lib, tmp_path, monkeypatch):
    """VERDICT r5 missing 4: config/simplified_object_picking.yaml AS SHIPPED (`depth_observation: False`: observations are the
    auto-encoder's features, sensor.py:176-222) through the reference's own `train(args)` with GRL_NUM_ENVS=4.  Every worker
    builds the reference's `EncodedDepthImgSensor`, whose `encoders.SimpleAutoEncoder(config)` is the deferred form there; the
    parent encodes the images of all four environments in ONE `encode` call per vectorised step, and the observations the
    model sees equal the per-environment batch-1 path of the reference (`encoder.encode(img)` inside each env)."""
    import enum
    import functools
    import importlib.util
    import yaml
    import grasp_rl.autoencoder as gae

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    robot = types.ModuleType("manipulation_main.gripperEnv.robot")
    robot.RobotEnv = type("RobotEnv", (), {"Status": enum.IntEnum("Status", {"RUNNING": 0, "SUCCESS": 1})})
    wrapper = types.ModuleType("manipulation_main.training.wrapper")
    wrapper.TimeFeatureWrapper = lambda env: env
    pkgs = {n: types.ModuleType(n) for n in ("manipulation_main", "manipulation_main.gripperEnv", "manipulation_main.training",
                                             "manipulation_main.common")}
    for m in pkgs.values():
        m.__path__ = []
    # `manipulation_main.gripperEnv.encoders` resolves to grasp_rl.autoencoder (here on the emulation engine)
    encoders = types.ModuleType("manipulation_main.gripperEnv.encoders")

    class EmulatedAutoEncoder(gae.SimpleAutoEncoder):            # same class, engine arguments bound for the CPU container
        def __new__(cls, config=None, *a, **k):
            return gae.SimpleAutoEncoder.__new__(gae.SimpleAutoEncoder if gae.defer_in_this_process() else cls, config)

        def __init__(self, config):
            super().__init__(config, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    encoders.SimpleAutoEncoder = EmulatedAutoEncoder
    pkgs["manipulation_main.gripperEnv"].encoders = encoders
    tu = types.ModuleType("manipulation_main.common.transform_utils")
    cu = types.ModuleType("manipulation_main.common.camera_utils")
    pkgs.update({"manipulation_main.gripperEnv.robot": robot, "manipulation_main.training.wrapper": wrapper,
                 "manipulation_main.gripperEnv.encoders": encoders, "manipulation_main.common.transform_utils": tu,
                 "manipulation_main.common.camera_utils": cu, "cv2": types.ModuleType("cv2"), "pybullet": types.ModuleType("pybullet")})
    pkgs["manipulation_main.common"].transform_utils, pkgs["manipulation_main.common"].camera_utils = tu, cu
    for n, m in pkgs.items():
        monkeypatch.setitem(sys.modules, n, m)
    import contextlib
    sys.modules["tensorflow"].name_scope = lambda name: contextlib.nullcontext()
    real_yaml_load = yaml.load
    monkeypatch.setattr(yaml, "load", lambda f, Loader=None: real_yaml_load(f, Loader=Loader or yaml.FullLoader))
    io_utils = load("manipulation_main.common.io_utils", "/root/reference/manipulation_main/common/io_utils.py")
    monkeypatch.setitem(sys.modules, "manipulation_main.common.io_utils", io_utils)
    pkgs["manipulation_main.common"].io_utils = io_utils
    utils = load("manipulation_main.utils", "/root/reference/manipulation_main/utils.py")
    monkeypatch.setitem(sys.modules, "manipulation_main.utils", utils)
    sensor_mod = load("manipulation_main.gripperEnv.sensor", "/root/reference/manipulation_main/gripperEnv/sensor.py")
    assert os.path.realpath(sensor_mod.__file__).startswith("/root/reference/")
    made = []

    def make(name, config=None, **kw):
        made.append(os.getpid())
        return _EncodedObsEnv(sensor_mod, config, seed=len(made) + 17 * (os.getpid() % 1000))
    sys.modules["gym"].make = make
    monkeypatch.delitem(sys.modules, "train_stable_baselines", raising=False)
    script = importlib.import_module("train_stable_baselines")
    assert os.path.realpath(script.__file__).startswith("/root/reference/")
    with open("/root/reference/config/simplified_object_picking.yaml") as f:
        cfg = yaml.safe_load(f)
    assert cfg["depth_observation"] is False and cfg["sensor"]["encoder_dir"] == "encoder_files/new_gripper_encoder"   # as shipped
    cfg["sensor"]["encoder_dir"] = "/root/reference/encoder_files/new_gripper_encoder"     # (the script is run from the repo root)
    cfg["SAC"]["buffer_size"], cfg["SAC"]["batch_size"] = 256, 4
    monkeypatch.chdir(tmp_path)
    os.makedirs("trained")
    with open("cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    monkeypatch.setenv("GRL_NUM_ENVS", "4")
    monkeypatch.setenv("GRL_ENV_START_METHOD", "fork")
    seen = {"encode_calls": [], "obs": [], "raw": []}
    real_encode = gae.SimpleAutoEncoder.encode

    def counting_encode(self, imgs):
        seen["encode_calls"].append(np.asarray(imgs).reshape(-1, 4096).shape[0])
        return real_encode(self, imgs)
    monkeypatch.setattr(gae.SimpleAutoEncoder, "encode", counting_encode)
    from grasp_rl.sb.vec_env import SubprocVecEnv, VecBatchedEncoder
    real_wait = SubprocVecEnv.step_wait

    def spy_wait(self):
        out = real_wait(self)
        seen["raw"].append(np.array(out[0], copy=True))
        return out
    monkeypatch.setattr(SubprocVecEnv, "step_wait", spy_wait)
    real_learn = SAC.learn

    def spy(self, total_timesteps, callback=None, **kw):
        inner = self.env
        while hasattr(inner, "venv"):
            inner = inner.venv
        seen["fan"] = inner._fan
        seen["obs_space"] = tuple(self.observation_space.shape)
        seen["n_envs"] = self.n_envs
        n0 = len(seen["encode_calls"])
        out = real_learn(self, total_timesteps, callback=callback, **kw)
        seen["learn_calls"] = seen["encode_calls"][n0:]
        seen["steps"], seen["updates"] = self.num_timesteps, self.n_updates
        seen["mlp"] = self.engine.cfg.extractor == 0 and self.engine.cfg.obs_dim == 101
        return out
    monkeypatch.setattr(SAC, "learn", spy)
    real_bwait = VecBatchedEncoder.step_wait

    def spy_bwait(self):
        out = real_bwait(self)
        seen["obs"].append(np.array(out[0], copy=True))
        return out
    monkeypatch.setattr(VecBatchedEncoder, "step_wait", spy_bwait)
    args = types.SimpleNamespace(config="cfg.yaml", model_dir="trained/ae", algo="SAC", load_dir=None, timestep="120",
                                 simple=False, shaped=False, visualize=False, timefeature=False)
    script.train(args)
    assert isinstance(seen["fan"], VecBatchedEncoder) and seen["n_envs"] == 4 and seen["obs_space"] == (101,) and seen["mlp"]
    assert seen["steps"] == 120 and seen["updates"] > 0
    # ONE encode call per vectorised step (30 steps + the reset), each over the 4 environments' images plus the terminal
    # observations of the environments whose episode ended in that step (5-step episodes: all four, every fifth step)
    calls = seen["learn_calls"]
    assert len(calls) == 1 + 30 and calls[0] == 4 and sorted(set(calls[1:])) == [4, 8] and calls.count(8) == 6
    # ... and nothing was encoded batch-1 inside a worker: the workers hold the deferred form (4096 pixels + the actuator width)
    assert all(r.shape == (4, 4097) for r in seen["raw"]) and len(seen["raw"]) == 30
    # the observations equal the reference's per-environment batch-1 path: encoder.encode(one image) inside each env
    enc = seen["fan"].encoder
    for raw, obs in list(zip(seen["raw"], seen["obs"]))[:6]:
        for k in range(4):
            z1 = real_encode(enc, raw[k, :4096].reshape(1, 64, 64, 1))[0]
            assert np.allclose(obs[k, :100], z1, atol=1e-6, rtol=1e-6) and obs[k, 100] == raw[k, 4096]
    # template train env + evaluation env were built in THIS process with REAL encoders (one of them serves the batch)
    assert made.count(os.getpid()) == 2 and enc.model_dir == os.path.realpath("/root/reference/encoder_files/new_gripper_encoder")
    assert os.path.isfile("trained/ae/ae.zip")
    sys.modules.pop("train_stable_baselines", None)


def test_reference_train_encoder_script(hostemu_lib, tmp_path, monkeypatch):
    """`manipulation_main/training/train_encoder.py`: its `train(args)` (:30-49) and `test(args)` (:52-64) on the
    reference's `config/encoder.yaml` (epochs / batch size reduced, `data_path` pointing at a synthetic pickle of
    the documented layout), with `manipulation_main.gripperEnv.encoders` resolving to `grasp_rl.autoencoder`."""
    import functools
    import importlib.util
    import pickle
    import yaml
    from grasp_rl import autoencoder as gae

    real_yaml_load = yaml.load
    monkeypatch.setattr(yaml, "load", lambda f, Loader=None: real_yaml_load(f, Loader=Loader or yaml.FullLoader))
    encoders = types.ModuleType("manipulation_main.gripperEnv.encoders")
    encoders.SimpleAutoEncoder = functools.partial(gae.SimpleAutoEncoder, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    pkgs = {n: types.ModuleType(n) for n in ("manipulation_main", "manipulation_main.gripperEnv", "manipulation_main.common")}
    for m in pkgs.values():
        m.__path__ = []
    pkgs["manipulation_main.gripperEnv"].encoders = encoders
    for n, m in list(pkgs.items()) + [("manipulation_main.gripperEnv.encoders", encoders)]:
        monkeypatch.setitem(sys.modules, n, m)
    spec = importlib.util.spec_from_file_location("manipulation_main.common.io_utils",
                                                  "/root/reference/manipulation_main/common/io_utils.py")
    io_utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(io_utils)
    monkeypatch.setitem(sys.modules, "manipulation_main.common.io_utils", io_utils)
    pkgs["manipulation_main.common"].io_utils = io_utils
    import matplotlib
    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location("reference_train_encoder", os.path.join(REF_TRAINING, "train_encoder.py"))
    script = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(script)

    rng = np.random.default_rng(0)

    def split(n):
        depth = np.zeros((n, 64, 64, 1), np.float32)
        masks = np.zeros((n, 64, 64, 1), np.int32)                      # 0 = flat surface, max = gripper, others = objects
        for i in range(n):
            r0, c0 = rng.integers(5, 40, 2)
            depth[i, r0:r0 + 18, c0:c0 + 16, 0] = rng.uniform(0.2, 0.5)
            masks[i, r0:r0 + 18, c0:c0 + 16, 0] = 3
            depth[i, :6, :6, 0] = 0.1
            masks[i, :6, :6, 0] = 7
        return {"depth": depth, "masks": masks, "rgb": np.zeros((n, 64, 64, 3), np.uint8)}
    with open(tmp_path / "imgs.pkl", "wb") as f:
        pickle.dump({"train": split(8), "test": split(4)}, f)
    with open("/root/reference/config/encoder.yaml") as f:
        cfg = yaml.safe_load(f)
    assert cfg["batch_size"] == 128 and cfg["encoding_dim"] == 100
    cfg.update(batch_size=4, epochs=2, data_path=str(tmp_path / "imgs.pkl"))
    with open(tmp_path / "encoder.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    model_dir = str(tmp_path / "enc")
    script.train(types.SimpleNamespace(config=str(tmp_path / "encoder.yaml"), model_dir=model_dir))
    assert all(os.path.isfile(os.path.join(model_dir, n)) for n in ("config.yaml", "model.h5", "history.csv"))
    loss = script.test(types.SimpleNamespace(model_dir=model_dir))
    assert np.isfinite(loss) and 0.0 < loss < 0.1
