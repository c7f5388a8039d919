"""`train_stable_baselines.py run` on what the reference ships: every zip under
/root/reference/trained_models is opened with `sb.<ALGO>.load` -- the class chosen by the `algorithm` key of
the folder's config.yaml, as the script does (:93-104) -- the folder's vecnormalize.pkl with
`VecNormalize.load`, and one deterministic prediction is made on the emulated engine.  Skipped where
/root/reference does not exist."""
import glob
import os

import numpy as np
import pytest
import yaml

import stable_baselines as sb
from grasp_rl.engine import QEngine, SacEngine
from grasp_rl.sb.dqn import BDQ, DQN
from grasp_rl.sb.sac import SAC
from grasp_rl.sb.spaces import Box
from hostemu_backend import NumpyHostBackend
from stable_baselines.common.vec_env import DummyVecEnv, VecNormalize

ROOT = "/root/reference/trained_models"
ZIPS = sorted(glob.glob(os.path.join(ROOT, "**", "*.zip"), recursive=True))
pytestmark = pytest.mark.skipif(not ZIPS, reason="/root/reference is not present on this box")


class _ObsOnlyEnv:
    """An environment with the observation / action spaces a loaded model declares (VecNormalize.load needs one)."""

    def __init__(self, observation_space, action_space):
        self.observation_space, self.action_space = observation_space, action_space

    def reset(self):
        return np.zeros(self.observation_space.shape, np.float32)

    def step(self, action):
        return self.reset(), 0.0, True, {}

    def close(self):
        pass


@pytest.fixture
def emulated_engines(hostemu_lib, monkeypatch):
    monkeypatch.setattr(SAC, "_engine_factory",
                        staticmethod(lambda cfg, device: SacEngine(cfg, backend=NumpyHostBackend(), lib_path=hostemu_lib)))
    f = staticmethod(lambda cfg, device: QEngine(cfg, backend=NumpyHostBackend(), lib_path=hostemu_lib))
    monkeypatch.setattr(DQN, "_engine_factory", f)
    monkeypatch.setattr(BDQ, "_engine_factory", f)


def _algorithm(zip_path):
    d = os.path.dirname(zip_path)
    for folder in (d, os.path.dirname(d)):
        cfg = os.path.join(folder, "config.yaml")
        if os.path.exists(cfg):
            with open(cfg) as f:
                return yaml.safe_load(f).get("algorithm"), folder
    return None, d


@pytest.mark.parametrize("zip_path", ZIPS, ids=[os.path.relpath(z, ROOT) for z in ZIPS])
def test_shipped_zip_loads_and_predicts(zip_path, emulated_engines):
    algo, folder = _algorithm(zip_path)
    cls = {"sac": sb.SAC, "dqn": sb.DQN, "bdq": sb.BDQ}.get(algo)
    if cls is None:
        pytest.skip("algorithm %r of %s is outside the path" % (algo, zip_path))
    model = cls.load(zip_path)
    P = model.get_parameters()
    assert len(P) > 4 and all(np.all(np.isfinite(v)) for v in P.values())
    shape = tuple(model.observation_space.shape)
    obs = np.zeros((1,) + shape, np.float32)
    pkl = os.path.join(folder, "vecnormalize.pkl")
    if os.path.exists(pkl):
        venv = VecNormalize.load(pkl, DummyVecEnv([lambda: _ObsOnlyEnv(model.observation_space, model.action_space)]))
        venv.training = False
        assert tuple(venv.obs_rms.mean.shape) == shape and venv.obs_rms.count > 1000
        obs = venv.normalize_obs(np.asarray(venv.obs_rms.mean, np.float32)[None])        # the mean observation -> zeros
        assert np.allclose(obs, 0.0, atol=1e-6)
    action, _ = model.predict(obs, deterministic=True)
    if isinstance(model.action_space, Box):
        assert action.shape == (1,) + tuple(model.action_space.shape) and np.all(np.abs(action) <= 1.0)
    else:
        assert np.all(np.isfinite(np.asarray(action, np.float64)))
