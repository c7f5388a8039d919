"""The reference's SAC call sequence (tests/test_sb_api_host.py) on the real engine / MI355X."""
import pytest

import test_sb_api_host as host
from fake_env import FakeGraspEnv
from stable_baselines.sac.policies import CnnPolicy as sacCnn
from stable_baselines.sac.policies import MlpPolicy as sacMlp

pytestmark = pytest.mark.gpu


def test_sac_cnn_reference_sequence_gpu(tmp_path):
    kwargs = {"layers": [64, 64], "cnn_extractor": host.create_augmented_nature_cnn(1)}
    m = host.run_reference_sequence(tmp_path, sacCnn, lambda s: FakeGraspEnv("depth", seed=s), kwargs, batch_size=16,
                                    total=40)
    assert m.engine.cfg.extractor == 1


def test_sac_mlp_reference_sequence_gpu(tmp_path):
    kwargs = {"layers": [64, 64], "layer_norm": False}
    host.run_reference_sequence(tmp_path, sacMlp, lambda s: FakeGraspEnv(seed=s, vector_dim=101), kwargs,
                                batch_size=16, total=40)


def test_vectorised_envs_gpu():
    """N sub-environments feeding one engine (north_star: vectorised envs fan out across host cores)."""
    import numpy as np
    import stable_baselines as sb
    from stable_baselines.common.vec_env import DummyVecEnv, VecNormalize
    env = VecNormalize(DummyVecEnv([(lambda s=s: FakeGraspEnv("depth", seed=s)) for s in range(4)]))
    kwargs = {"layers": [64, 64], "cnn_extractor": host.create_augmented_nature_cnn(1)}
    model = sb.SAC(sacCnn, env, policy_kwargs=kwargs, buffer_size=256, batch_size=32, learning_starts=16)
    model.learn(total_timesteps=64)
    assert model.num_timesteps == 64 and model.engine.replay_size() == 64 and model.n_updates > 0
    a, _ = model.predict(env.reset(), deterministic=True)
    assert a.shape == (4, 5) and np.all(np.isfinite(a))


def test_dqn_and_bdq_reference_sequences_gpu(tmp_path):
    import os
    import stable_baselines as sb
    os.makedirs(str(tmp_path / "a")), os.makedirs(str(tmp_path / "b"))
    host.run_q_sequence(tmp_path / "a",
                        lambda env: sb.DQN(host.DQNMlpPolicy, env, verbose=2, gamma=0.99, batch_size=16,
                                           prioritized_replay=True, learning_starts=20,
                                           target_network_update_freq=10, buffer_size=128),
                        lambda s: FakeGraspEnv(seed=s, vector_dim=100, discrete_actions=12), n_steps=60)
    host.run_q_sequence(tmp_path / "b",
                        lambda env: sb.BDQ(host.MlpActPolicy, env, policy_kwargs={"layers": [[64, 64], [32], [32]]},
                                           batch_size=16, buffer_size=128, num_actions_pad=33, learning_starts=20,
                                           target_network_update_freq=10, prioritized_replay=False),
                        lambda s: FakeGraspEnv(seed=s, vector_dim=100, act_dim=3), n_steps=60)


def test_subproc_envs_overlap_gpu():
    """Simulator worker processes step while the GPU runs the update (opt-in overlap_env_step)."""
    import functools
    import numpy as np
    import stable_baselines as sb
    from stable_baselines.common.vec_env import SubprocVecEnv, VecNormalize
    venv = SubprocVecEnv([functools.partial(FakeGraspEnv, "depth", 7, s) for s in range(4)])
    try:
        env = VecNormalize(venv)
        kwargs = {"layers": [64, 64], "cnn_extractor": host.create_augmented_nature_cnn(1)}
        model = sb.SAC(sacCnn, env, policy_kwargs=kwargs, buffer_size=256, batch_size=32, learning_starts=16,
                       overlap_env_step=True)
        model.learn(total_timesteps=96)
        assert model.num_timesteps == 96 and model.engine.replay_size() == 96 and model.n_updates > 0
        a, _ = model.predict(env.reset(), deterministic=True)
        assert a.shape == (4, 5) and np.all(np.isfinite(a))
    finally:
        venv.close()


@pytest.mark.parametrize("extractor,channels,rgb_u8", [("augmented", 2, False), ("augmented", 5, True), ("mlp", 0, False)])
def test_observations_uploaded_once_gpu(extractor, channels, rgb_u8):
    """grl_observe / grl_act(GRL_ACT_OBSERVED) / grl_replay_add_observed leave the replay ring, the running statistics and
    the actions bit-identical to the separate uploads (tests/observed_util.py; the CPU suite runs it on the emulation build)."""
    import observed_util
    from grasp_rl.engine import SacEngine
    observed_util.check_observed_path(lambda cfg: SacEngine(cfg, device="cuda:0"), extractor, channels, rgb_u8=rgb_u8)


def test_learning_on_observations_uploaded_once_equals_the_separate_uploads_gpu(tmp_path):
    """SAC(device_norm=True) -- one upload per env step -- against the default host-statistics path: parameters after
    learn() and the saved statistics, bit for bit (same body as the emulation test in test_sb_api_host.py)."""
    import numpy as np
    import stable_baselines as sb
    from stable_baselines.common.vec_env import DummyVecEnv, VecNormalize
    outs = []
    for dev in (False, True):
        env = DummyVecEnv([(lambda s=s: FakeGraspEnv("depth", seed=s)) for s in range(3)])
        env = VecNormalize(env, norm_obs=True, norm_reward=True, clip_obs=10.)
        m = sb.SAC(sacCnn, env, policy_kwargs={"layers": [64, 64], "cnn_extractor": host.create_augmented_nature_cnn(1)},
                   buffer_size=64, batch_size=4, learning_starts=6, seed=5, device_norm=dev)
        m.learn(total_timesteps=45)
        outs.append((m.get_parameters(), env.obs_rms.mean.copy(), env.obs_rms.var.copy(), env.obs_rms.count,
                     m.engine.fetch("rp_obs"), m.engine.fetch("rp_next"), m.engine.fetch("rp_done")))
        m.engine.close()
    (p0, m0, v0, c0, *r0), (p1, m1, v1, c1, *r1) = outs
    assert c0 == c1 and np.array_equal(m0, m1) and np.array_equal(v0, v1)
    for x, y in zip(r0, r1):
        assert np.array_equal(x, y)
    assert r0[2].sum() > 0          # episodes ended: terminal rows went through the row map
    for k in p0:
        assert np.array_equal(p0[k], p1[k]), k
