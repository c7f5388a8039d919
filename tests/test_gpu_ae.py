"""Depth auto-encoder training on the MI355X (through the C ABI) against oracle/autoencoder.py, and the
reference's `SimpleAutoEncoder.train` surface (encoders.py:40-50)."""
import os

import numpy as np
import pytest

import ae_parity_util as au

pytestmark = pytest.mark.gpu


def test_ae_training_step_matches_oracle():
    au.ae_check(B=8, n_steps=3)


def test_ae_training_step_matches_oracle_at_the_configured_batch():
    """config/encoder.yaml / `bench.py --workload ae_train`: batch 128 (other reduction splits, riders and tile
    lists than B = 8)."""
    au.ae_check(B=128, n_steps=2)


def test_simple_autoencoder_train_surface(tmp_path):
    from grasp_rl.autoencoder import SimpleAutoEncoder
    cfg = {"network": [{"filters": 32, "kernel_size": 7, "strides": 2}, {"filters": 32, "kernel_size": 5, "strides": 2},
                       {"filters": 32, "kernel_size": 3, "strides": 2}], "encoding_dim": 100, "learning_rate": 1e-3}
    rng = np.random.default_rng(0)
    x = np.zeros((72, 64, 64, 1), np.float32)
    for i in range(x.shape[0]):
        r0, c0 = rng.integers(5, 40, 2)
        x[i, r0:r0 + 20, c0:c0 + 18, 0] = rng.uniform(0.2, 0.5)
    model = SimpleAutoEncoder(cfg)
    hist = model.train(x, x, batch_size=16, epochs=6, model_dir=str(tmp_path))
    assert len(hist["loss"]) == 6 and hist["loss"][-1] < hist["loss"][0] and np.isfinite(hist["val_loss"]).all()
    assert os.path.exists(tmp_path / "history.csv") and os.path.exists(tmp_path / "model.npz")
    assert os.path.exists(tmp_path / "model.h5")                  # the file Keras' load_weights reads (encoders.py:27-31)
    z = model.encode(x[:5])
    assert z.shape == (5, 100) and model.encoding_shape == (100,)
    os.remove(tmp_path / "model.npz")                             # reload through the HDF5 file alone
    m2 = SimpleAutoEncoder(cfg)
    m2.load_weights(str(tmp_path))
    assert abs(m2.test(x[:8], x[:8]) - min(hist["val_loss"])) < 0.05
    assert m2.encode(x[:5]).shape == (5, 100)
