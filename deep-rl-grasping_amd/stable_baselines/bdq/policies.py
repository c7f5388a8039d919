from grasp_rl.sb.policies import BdqMlpActPolicy as MlpActPolicy  # noqa: F401
