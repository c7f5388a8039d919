from grasp_rl.sb.running_mean_std import RunningMeanStd  # noqa: F401
