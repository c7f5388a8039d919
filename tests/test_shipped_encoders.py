"""The auto-encoders the reference ships (encoder_files/*/{config.yaml, model.h5}) through the reference-shaped
surface: `SimpleAutoEncoder(config)` -> `load_weights(model_dir)` (Keras HDF5 read by grasp_rl.keras_h5) ->
`encode` / `predict` on the six real depth frames of tests/golden (sensor.py:206-222 usage).  Engine = TEST-ONLY
g++ emulation build.  Skipped where /root/reference does not exist."""
import os

import numpy as np
import pytest
import yaml

from grasp_rl.autoencoder import SimpleAutoEncoder
from hostemu_backend import NumpyHostBackend
from oracle import autoencoder as oae
from oracle import fixtures

ROOT = "/root/reference/encoder_files"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DIRS = sorted(d for d in (os.listdir(ROOT) if os.path.isdir(ROOT) else []) if os.path.exists(os.path.join(ROOT, d, "model.h5")))
pytestmark = pytest.mark.skipif(not DIRS, reason="/root/reference is not present on this box")


@pytest.mark.parametrize("name", DIRS)
def test_shipped_encoder_encodes_and_reconstructs(name, hostemu_lib):
    model_dir = os.path.join(ROOT, name)
    with open(os.path.join(model_dir, "config.yaml")) as f:
        config = yaml.safe_load(f)
    frames = np.load(os.path.join(GOLD, "depth_frames.npz"))["frames"].astype(np.float32).reshape(-1, 64, 64, 1)
    model = SimpleAutoEncoder(config, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    model.load_weights(model_dir)
    model._engine(len(frames))                      # a small static batch for the emulated engine
    z = model.encode(frames)
    W = fixtures.load_keras_ae_h5(os.path.join(model_dir, "model.h5"))
    assert z.shape == (len(frames), 100) and np.allclose(z, oae.encode(W, frames), atol=2e-5, rtol=2e-4)
    if name == "new_gripper_encoder":
        assert np.allclose(z, np.load(os.path.join(GOLD, "ae_encodings.npz"))["z"], atol=2e-5, rtol=2e-4)
    rec = model.predict(frames)
    mse = float(np.mean((rec - frames) ** 2))
    assert rec.shape == frames.shape and mse < 0.02, mse            # 0.0026 - 0.0086 with the right padding / flatten order
    assert abs(model.test(frames, frames) - mse) < 1e-7


def test_batched_encoding_across_vectorised_envs(hostemu_lib):
    """with N environments the encoder runs ONCE per step in the parent (VecBatchedEncoder) while the
    workers carry a DeferredEncoder -- the observations must equal what the reference's per-environment batch-1
    `encoder.encode(img)` (sensor.py:220-222) would have produced, terminal observations included."""
    from grasp_rl.autoencoder import DeferredEncoder
    from grasp_rl.sb.spaces import Box
    from grasp_rl.sb.vec_env import DummyVecEnv, VecBatchedEncoder
    model_dir = os.path.join(ROOT, "new_gripper_encoder")
    with open(os.path.join(model_dir, "config.yaml")) as f:
        config = yaml.safe_load(f)
    frames = np.load(os.path.join(GOLD, "depth_frames.npz"))["frames"].astype(np.float32).reshape(-1, 64, 64)
    model = SimpleAutoEncoder(config, backend=NumpyHostBackend(), lib_path=hostemu_lib)
    model.load_weights(model_dir)
    model._engine(4)
    calls = {"n": 0}
    enc = model.encode

    def counting(imgs):
        calls["n"] += 1
        return enc(imgs)
    model.encode = counting

    class Env:                                     # the reference env's observation with an encoder sensor: [encoding | width]
        def __init__(self, k):
            self.k, self.t, self.sensor = k, 0, DeferredEncoder()
            self.observation_space = Box(-np.inf, np.inf, shape=(4096 + 1,), dtype=np.float32)
            self.action_space = Box(-1, 1, shape=(3,), dtype=np.float32)

        def _obs(self):
            img = frames[(self.k + self.t) % len(frames)]
            return np.append(self.sensor.encode(img.reshape(1, 64, 64, 1)).squeeze(), 0.1 * self.k).astype(np.float32)

        def reset(self):
            self.t = 0
            return self._obs()

        def step(self, a):
            self.t += 1
            return self._obs(), 0.0, self.t % 2 == 0, {}
    venv = VecBatchedEncoder(DummyVecEnv([(lambda k=k: Env(k)) for k in range(3)]), model)
    assert venv.observation_space.shape == (101,)
    obs = venv.reset()
    assert calls["n"] == 1 and obs.shape == (3, 101)
    want = np.stack([np.append(enc(frames[k].reshape(1, 64, 64, 1))[0], 0.1 * k) for k in range(3)])   # batch-1, per env
    assert np.allclose(obs, want, atol=1e-6, rtol=1e-6)
    obs, _, dones, infos = venv.step(np.zeros((3, 3), np.float32))
    assert calls["n"] == 2 and not dones.any()
    obs, _, dones, infos = venv.step(np.zeros((3, 3), np.float32))
    assert calls["n"] == 3 and dones.all()                         # reset observations + three terminal observations: one call
    for k in range(3):
        t = np.append(enc(frames[(k + 2) % len(frames)].reshape(1, 64, 64, 1))[0], 0.1 * k)
        assert np.allclose(infos[k]["terminal_observation"], t, atol=1e-6, rtol=1e-6)
        assert np.allclose(obs[k], np.append(enc(frames[k % len(frames)].reshape(1, 64, 64, 1))[0], 0.1 * k), atol=1e-6, rtol=1e-6)
