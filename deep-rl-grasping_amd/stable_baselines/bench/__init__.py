from grasp_rl.sb.monitor import Monitor  # noqa: F401
