from grasp_rl.sb.vec_env import (DummyVecEnv, SubprocVecEnv, VecEnv, VecEnvWrapper, VecNormalize,  # noqa: F401
                                 sync_envs_normalization, unwrap_vec_normalize)


class VecFrameStack:
    def __init__(self, *a, **k):
        raise NotImplementedError("VecFrameStack is not used by the reference's training path")
