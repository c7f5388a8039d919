from grasp_rl.sb.sac import SAC  # noqa: F401
