from grasp_rl.sb.vec_env import VecNormalize  # noqa: F401
