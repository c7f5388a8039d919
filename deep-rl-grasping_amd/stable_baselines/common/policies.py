from grasp_rl.sb.policies import ActorCriticCnnPolicy as CnnPolicy  # noqa: F401
from grasp_rl.sb.policies import ActorCriticMlpPolicy as MlpPolicy  # noqa: F401
