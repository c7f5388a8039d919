"""Maximum sizes (BASELINE config 2: `buffer_size` 1 M, gripper_grasp.yaml): a full one-million-transition
depth replay resident in HBM (32.8 GB) -- element offsets beyond 2^31, ring wrap at the capacity, sampling
at the far end -- and the prioritised sampler over the same number of priorities."""
import numpy as np
import pytest
import torch

from grasp_rl import _capi
from grasp_rl.engine import QEngine, SacEngine
from grasp_rl.init import init_parameters

pytestmark = pytest.mark.gpu
CAP = 1_000_000


def _pattern(i0, n, dev):
    """Transition i carries its index: depth plane = (i mod 8191) / 8192 + pixel / 2^20, direct feature = (i mod 97) / 97."""
    i = torch.arange(i0, i0 + n, device=dev, dtype=torch.float32)
    pix = torch.arange(4096, device=dev, dtype=torch.float32) / float(1 << 20)
    obs = torch.zeros((n, 64, 64, 2), device=dev)
    obs[..., 0] = (torch.remainder(i, 8191.0) / 8192.0)[:, None, None] + pix.view(1, 64, 64)
    obs[:, 0, 0, 1] = torch.remainder(i, 97.0) / 97.0
    return obs


def test_one_million_transition_ring():
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * (1 << 30):
        pytest.skip("needs ~40 GB of free HBM")
    dev = torch.device("cuda", 0)
    cfg = _capi.make_config("augmented", obs_channels=2, n_direct=1, act_dim=5, layers=(64, 64), batch_size=8,
                            replay_capacity=CAP, normalize=False, act_batch=1, seed=3)
    eng = SacEngine(cfg)
    eng.set_parameters(init_parameters(eng.table, seed=0))
    chunk = 50_000
    for k0 in range(0, CAP, chunk):
        with torch.cuda.stream(eng.be.stream):
            obs = _pattern(k0, chunk, dev)
            nxt = _pattern(k0 + 1, chunk, dev)
            act = torch.zeros((chunk, 5), device=dev)
            rew = torch.arange(k0, k0 + chunk, device=dev, dtype=torch.float32)
            done = torch.zeros(chunk, device=dev)
            eng.replay_add_device(obs, act, rew, nxt, done)
        eng.be.stream.synchronize()
    assert eng.replay_size() == CAP
    idx = np.array([[0, 1, 524_287, 524_288, 777_777, CAP - 3, CAP - 2, CAP - 1]], np.int64)   # 524288 * 4096 = 2^31 elements
    eps = np.zeros((1, 8, 5), np.float32)
    eng.train(1, idx, eps)
    x = eng.fetch("x_obs", (8, 64, 64, 1))
    xn = eng.fetch("x_next", (8, 64, 64, 1))
    want = _pattern(0, 1, dev)[0, :, :, 0].cpu().numpy() - 0.0
    for k, i in enumerate(idx[0]):
        base = np.float32(np.float32(i % 8191) / np.float32(8192.0))
        nbase = np.float32(np.float32((i + 1) % 8191) / np.float32(8192.0))
        pix = (np.arange(4096, dtype=np.float32) / np.float32(1 << 20)).reshape(64, 64)
        assert np.array_equal(x[k, :, :, 0], (base + pix).astype(np.float32) / np.float32(255.0)), int(i)
        assert np.array_equal(xn[k, :, :, 0], (nbase + pix).astype(np.float32) / np.float32(255.0)), int(i)
    assert np.array_equal(eng.fetch("rew", (8,)), idx[0].astype(np.float32))
    assert want.shape == (64, 64)
    # ring wrap at the capacity: 5 more transitions land in slots 0..4
    with torch.cuda.stream(eng.be.stream):
        eng.replay_add_device(_pattern(7_000_000, 5, dev), torch.zeros((5, 5), device=dev),
                              torch.full((5,), -1.0, device=dev), _pattern(7_000_001, 5, dev), torch.zeros(5, device=dev))
    eng.be.stream.synchronize()
    assert eng.replay_size() == CAP
    eng.train(1, np.array([[0, 4, 5, CAP - 1, 0, 0, 0, 0]], np.int64), eps)
    r = eng.fetch("rew", (8,))
    assert r[0] == -1.0 and r[1] == -1.0 and r[2] == 5.0 and r[3] == float(CAP - 1)
    # device RNG over the full ring: indices must cover the far end too
    eng.train(64)
    m = eng.metrics()
    assert all(np.isfinite(v) for v in m.values()), m
    eng.close()


def _pattern_rgbd(i0, n, dev):
    """RGB-D transition i: R = i mod 256, G = (i // 256) mod 256, B = (pixel + i) mod 256 (integers, as the camera
    delivers them), depth = (i mod 8191) / 8192 + pixel / 2^20, direct feature = (i mod 97) / 97."""
    i = torch.arange(i0, i0 + n, device=dev, dtype=torch.int64)
    pix = torch.arange(4096, device=dev, dtype=torch.int64)
    obs = torch.zeros((n, 4096, 5), device=dev)
    obs[..., 0] = torch.remainder(i, 256).float()[:, None]
    obs[..., 1] = torch.remainder(i // 256, 256).float()[:, None]
    obs[..., 2] = torch.remainder(pix[None, :] + i[:, None], 256).float()
    obs[..., 3] = (torch.remainder(i, 8191).float() / 8192.0)[:, None] + pix.float()[None, :] / float(1 << 20)
    obs[:, 0, 4] = torch.remainder(i, 97).float() / 97.0
    return obs.view(n, 64, 64, 5)


def test_one_million_transition_rgbd_ring_with_byte_colours():
    """BASELINE configs[3]: the reference's 1 M-transition buffer on RGB-D observations (64x64x5) resident in HBM.
    With grl_config.replay_rgb_u8 a stored observation is 32 KB (packed colour dword + float32 depth per pixel):
    65.5 GB for obs + next_obs.  Far-end reads (offsets beyond 2^31 elements) and the wrap must return exactly
    what was stored."""
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * (1 << 30):
        pytest.skip("needs ~75 GB of free HBM")
    dev = torch.device("cuda", 0)
    cfg = _capi.make_config("augmented", obs_channels=5, n_direct=1, act_dim=5, layers=(64, 64), batch_size=8,
                            replay_capacity=CAP, normalize=False, act_batch=1, seed=3, replay_rgb_u8=True)
    eng = SacEngine(cfg)
    assert eng.sizes.replay_bytes < 70 * (1 << 30)          # the float32 layout would need 131 GB
    eng.set_parameters(init_parameters(eng.table, seed=0))
    chunk = 20_000
    for k0 in range(0, CAP, chunk):
        with torch.cuda.stream(eng.be.stream):
            eng.replay_add_device(_pattern_rgbd(k0, chunk, dev), torch.zeros((chunk, 5), device=dev),
                                  torch.arange(k0, k0 + chunk, device=dev, dtype=torch.float32),
                                  _pattern_rgbd(k0 + 1, chunk, dev), torch.zeros(chunk, device=dev))
        eng.be.stream.synchronize()
    assert eng.replay_size() == CAP
    idx = np.array([[0, 1, 262_143, 262_144, 777_777, CAP - 3, CAP - 2, CAP - 1]], np.int64)   # 262144 * 8192 = 2^31 dwords
    eps = np.zeros((1, 8, 5), np.float32)
    eng.train(1, idx, eps)
    x = eng.fetch("x_obs", (8, 64, 64, 4))
    xn = eng.fetch("x_next", (8, 64, 64, 4))
    for k, i in enumerate(idx[0]):
        for arr, j in ((x, int(i)), (xn, int(i) + 1)):
            want = _pattern_rgbd(j, 1, dev)[0, :, :, :4].cpu().numpy() / np.float32(255.0)
            assert np.array_equal(arr[k], want.astype(np.float32)), (int(i), j)
    assert np.array_equal(eng.fetch("rew", (8,)), idx[0].astype(np.float32))
    with torch.cuda.stream(eng.be.stream):                   # wrap: 3 more transitions land in slots 0..2
        eng.replay_add_device(_pattern_rgbd(5_000_000, 3, dev), torch.zeros((3, 5), device=dev),
                              torch.full((3,), -1.0, device=dev), _pattern_rgbd(5_000_001, 3, dev), torch.zeros(3, device=dev))
    eng.be.stream.synchronize()
    eng.train(1, np.array([[0, 2, 3, CAP - 1, 0, 0, 0, 0]], np.int64), eps)
    r = eng.fetch("rew", (8,))
    assert r[0] == -1.0 and r[1] == -1.0 and r[2] == 3.0 and r[3] == float(CAP - 1)
    want = _pattern_rgbd(5_000_000, 1, dev)[0, :, :, :4].cpu().numpy() / np.float32(255.0)
    assert np.array_equal(eng.fetch("x_obs", (8, 64, 64, 4))[0], want.astype(np.float32))
    eng.train(32)                                            # device RNG over the full ring
    assert all(np.isfinite(v) for v in eng.metrics().values())
    eng.close()


def test_prioritised_sampler_over_one_million_priorities():
    cfg = _capi.make_q_config("dqn", 16, 1, 4, branch_hidden=(16,), value_hidden=(16,), batch_size=64,
                              replay_capacity=CAP, lr=1e-3, prioritized=True, per_stratified=True)
    eng = QEngine(cfg)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    P = {}
    rng = np.random.default_rng(0)
    for name, _, _, shape, _ in eng.table:
        if "/target_q_func/" not in name:
            P[name] = (rng.normal(0.0, 0.1, shape) if len(shape) >= 2 else np.zeros(shape)).astype(np.float32)
    for name, *_ in eng.table:
        if "/target_q_func/" in name:
            P[name] = P[name.replace("/target_q_func", "")].copy()
    eng.set_parameters(P)
    chunk = 250_000
    for k0 in range(0, CAP, chunk):
        with torch.cuda.stream(eng.be.stream):
            obs = torch.randn((chunk, 16), generator=g, device=dev)
            nxt = torch.randn((chunk, 16), generator=g, device=dev)
            act = torch.randint(0, 4, (chunk, 1), generator=g, device=dev).float()
            rew = torch.randn(chunk, generator=g, device=dev)
            done = torch.zeros(chunk, device=dev)
            eng.replay_add_device(obs, act, rew, nxt, done)
        eng.be.stream.synchronize()
    # all leaves 1.0: the stratified sampler with u = 0.5 must hit the middle of every stratum of the sampled range
    # [0, CAP - 2] (the published sum(0, len - 1) leaves the newest transition out) exactly
    u = np.full((1, 64), 0.5)
    eng.train_per(1, beta=0.4, u=u)
    idx = eng.sampled_indices()
    want = np.floor((np.arange(64) + 0.5) * ((CAP - 1) / 64.0)).astype(np.int64)
    assert np.array_equal(idx, want), (idx[:4], want[:4])
    w = eng.importance_weights()
    assert np.allclose(w, 1.0, atol=1e-6)
    # arbitrary leaves over the whole ring as the common INITIAL state, then five sample -> update rounds in which the
    # oracle (stable-baselines' trees, 2^20 leaves) and the device (1024 blocks of 1024) evolve independently: equal
    # indices, bit-identical float64 leaves, weights to float32 rounding
    from oracle.per import PerOracle
    prng = np.random.default_rng(21)
    pvals = ((prng.gamma(0.7, 1.0, CAP).astype(np.float32) + np.float32(1e-6)) ** np.float32(0.6)).astype(np.float64)
    pvals[prng.integers(0, CAP, 1000)] *= 300.0                     # a few very heavy transitions
    eng.store_priorities(pvals)
    orc = PerOracle(CAP, 0.6, 1e-6, stratified=True)
    orc.add(CAP)
    orc._it_sum[np.arange(CAP)] = pvals
    orc._it_min[np.arange(CAP)] = pvals
    for trial in range(5):
        uu = prng.random(64)
        eng.train_per(1, beta=0.7, u=uu[None])
        ref_idx, ref_w = orc.sample(uu, 0.7)
        assert np.array_equal(eng.sampled_indices(), ref_idx), trial
        assert np.allclose(eng.importance_weights(), ref_w.astype(np.float32), rtol=1e-6, atol=0)
        orc.update(ref_idx, eng.priorities())
        assert np.array_equal(eng.stored_priorities(), orc.leaves), trial
    # after the write-back the trained transitions carry new priorities; sampling keeps working
    eng.train_per(20, beta=0.5)
    idx = eng.sampled_indices()
    assert idx.min() >= 0 and idx.max() < CAP and len(np.unique(idx)) > 32
    assert all(np.isfinite(v) for v in eng.metrics().values())
    eng.close()
