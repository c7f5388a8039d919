from grasp_rl.sb.vec_env import (DummyVecEnv, VecEnv, VecEnvWrapper, VecNormalize,  # noqa: F401
                                 sync_envs_normalization, unwrap_vec_normalize)


class SubprocVecEnv(DummyVecEnv):
    """Imported (never instantiated) by sb_helper.py:19.  Stepping in-process keeps the call surface."""


class VecFrameStack:
    def __init__(self, *a, **k):
        raise NotImplementedError("VecFrameStack is not used by the reference's training path")
