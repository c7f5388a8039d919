"""Auto-encoder training step: engine vs oracle/autoencoder.py (shared by the CPU plan test and the GPU test)."""
import numpy as np
import torch

from grasp_rl.autoencoder import AeEngine, PARAM_NAMES
from oracle import autoencoder as oae


def ae_check(backend=None, lib_path=None, B=4, n_steps=3, seed=0, lr=2e-4):
    rng = np.random.default_rng(seed + 5)
    P0 = oae.init_params(seed)
    for k in P0:                       # non-zero biases so that every gradient path is exercised
        if k.endswith("bias"):
            P0[k] = rng.normal(0, 0.05, P0[k].shape).astype(np.float32)
    # depth-like images: mostly zeros (masked background) with blobs in [0.2, 0.5]
    x = np.zeros((n_steps * B, 64, 64, 1), np.float32)
    for i in range(x.shape[0]):
        r0, c0 = rng.integers(5, 40, 2)
        x[i, r0:r0 + 20, c0:c0 + 18, 0] = rng.uniform(0.2, 0.5, (20, 18))
    orc = oae.AeOracle(P0, lr=lr)
    eng = AeEngine(B, lr, act_batch=4, backend=backend, lib_path=lib_path)
    assert [n for n, *_ in eng.table] == PARAM_NAMES
    eng.set_parameters(P0)
    for s in range(n_steps):
        xb = x[s * B:(s + 1) * B]
        ref = orc.step(xb)
        loss = eng.train_batches(xb)
        out = eng.reconstruction()
        assert np.allclose(out, ref["out"], atol=2e-5, rtol=1e-4), np.abs(out - ref["out"]).max()
        assert abs(loss - ref["loss"]) <= 1e-4 * abs(ref["loss"]) + 1e-7, (loss, ref["loss"])
        if s == 0:
            G = eng.get_gradients()
            for n, g in ref["grads"].items():
                d = np.abs(G[n] - g).max()
                assert d <= 1e-3 * max(np.abs(g).max(), 1e-12) + 1e-9, "grad %s: %.3e vs max %.3e" % (n, d, np.abs(g).max())
    Pe, Po = eng.get_parameters(), orc.params()
    for n in PARAM_NAMES:              # Adam's first steps move every weight by ~lr: compare on that scale
        d = np.abs(Pe[n] - Po[n])
        assert d.max() <= 0.3 * lr * n_steps + 1e-7, "param %s: max |d| %.3e" % (n, d.max())
        assert d.mean() <= 0.02 * lr * n_steps + 1e-9, "param %s: mean |d| %.3e" % (n, d.mean())
    z = eng.encode(x[:3])
    assert np.allclose(z, oae.encode(Po, x[:3]), atol=2e-5, rtol=2e-4)
    # forward-only pass (Model.predict): same numbers as the oracle's forward at the engine's parameters,
    # for a batch that is not a multiple of the engine's, without touching parameters or optimiser state
    rec = eng.reconstruct(x[:B + 1])
    want = oae.AeOracle(Pe, lr=lr).forward(torch.from_numpy(x[:B + 1]))[0].detach().numpy()
    assert rec.shape == (B + 1, 64, 64, 1) and np.allclose(rec, want, atol=2e-5, rtol=1e-4), np.abs(rec - want).max()
    Pa = eng.get_parameters()
    assert all(np.array_equal(Pa[n], Pe[n]) for n in PARAM_NAMES)
    eng.close()
