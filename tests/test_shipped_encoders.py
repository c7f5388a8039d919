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
