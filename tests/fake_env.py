"""Stand-in for the reference's PyBullet 'gripper-env-v0' (manipulation_main/gripperEnv/robot.py): same
observation / action spaces and the attributes sb_helper.py reads (``is_simplified``, ``depth_obs``,
``full_obs``, ``history``, ``episode_rewards``, ``sr_mean``, ``curriculum``), synthetic dynamics."""
import numpy as np

from grasp_rl import synthetic
from grasp_rl.sb.spaces import Box, Discrete


class _Curriculum:
    _lambda = 0.0


class FakeGraspEnv:
    def __init__(self, kind="depth", episode_len=7, seed=0, vector_dim=None, discrete_actions=None, act_dim=5):
        self._rng = np.random.default_rng(seed)       # seed=None: OS entropy (fan-out workers built from ONE factory)
        self.vector_dim = vector_dim
        if vector_dim:
            self.observation_space = Box(-1.0, 1.0, shape=(vector_dim,), dtype=np.float32)
            self.depth_obs = self.full_obs = False
        else:
            self.stats = synthetic.load_obs_stats(kind)
            C = self.stats["mean"].shape[-1]
            self.observation_space = Box(0, 255, shape=(64, 64, C), dtype=np.float32)
            self.depth_obs, self.full_obs = (kind == "depth"), (kind == "rgbd")
        self.action_space = Discrete(discrete_actions) if discrete_actions else Box(-1.0, 1.0, shape=(act_dim,), dtype=np.float32)
        self.act_dim = act_dim
        self.episode_len = episode_len
        self.episode_step = 0
        self.episode_rewards = 0.0
        self.history = []
        self.sr_mean = 0.0
        self.curriculum = _Curriculum()
        self.n_resets = 0

    def is_simplified(self):
        return False

    def _obs(self):
        if self.vector_dim:
            return self._rng.uniform(-1, 1, self.vector_dim).astype(np.float32)
        m, v = self.stats["mean"], self.stats["var"]
        o = self._rng.normal(m, np.sqrt(v)).astype(np.float32)
        o[..., -1] = 0.0
        o[0, 0, -1] = self._rng.uniform(0, 1)
        return o

    def reset(self):
        self.episode_step = 0
        self.episode_rewards = 0.0
        self.n_resets += 1
        return self._obs()

    def step(self, action):
        if hasattr(self.action_space, "n"):
            assert 0 <= int(action) < self.action_space.n
            action = np.array([float(action)])
        else:
            assert np.asarray(action).shape == (self.act_dim,) and np.all(np.abs(action) <= 1 + 1e-6)
        self.episode_step += 1
        r = float(-200.0 + 100.0 * np.tanh(np.sum(action)))
        self.episode_rewards += r
        done = self.episode_step >= self.episode_len
        if done:
            self.history.append(1)
            self.sr_mean = 1.0
        return self._obs(), r, done, {"is_success": done, "episode_step": self.episode_step,
                                      "episode_rewards": self.episode_rewards, "status": 1}

    def close(self):
        pass


class EncodedFakeEnv(FakeGraspEnv):
    """Observation = [auto-encoder features of a depth image | gripper width] -- the reference env with
    `depth_observation: False` (robot.py:83-89,185-190): the env's sensor builds `encoders.SimpleAutoEncoder(config)`,
    loads `<model_dir>/model.h5` and calls `.encode(img[None, :, :, None]).squeeze()` per step (sensor.py:190-192,220-222)."""
    AE_CONFIG = {"encoding_dim": 100, "learning_rate": 2e-4,
                 "network": [{"filters": 32, "kernel_size": 7, "strides": 2}, {"filters": 32, "kernel_size": 5, "strides": 2},
                             {"filters": 32, "kernel_size": 3, "strides": 2}]}

    def __init__(self, model_dir, seed=0, episode_len=5, encoder_factory=None):
        super().__init__(seed=seed, vector_dim=101, episode_len=episode_len)
        from grasp_rl import autoencoder
        self._encoder = (encoder_factory or autoencoder.SimpleAutoEncoder)(dict(self.AE_CONFIG))
        self._encoder.load_weights(model_dir)
        dim = int(np.prod(self._encoder.encoding_shape)) + 1
        self.observation_space = Box(-np.inf, np.inf, shape=(dim,), dtype=np.float32)
        self.last_image = None

    def is_simplified(self):
        return True

    def _obs(self):
        yy, xx = np.mgrid[0:64, 0:64]
        cx, cy = self._rng.uniform(16, 48, 2)
        img = (0.3 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / 60.0)).astype(np.float32)
        self.last_image = img
        z = self._encoder.encode(img.reshape(1, 64, 64, 1)).squeeze()
        return np.append(z, 0.05 * (self.episode_step % 3)).astype(np.float32)
