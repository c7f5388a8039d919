from grasp_rl.sb.evaluation import evaluate_policy  # noqa: F401
