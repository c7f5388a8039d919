from grasp_rl.sb.policies import DqnCnnPolicy as CnnPolicy  # noqa: F401
from grasp_rl.sb.policies import DqnLnMlpPolicy as LnMlpPolicy  # noqa: F401
from grasp_rl.sb.policies import DqnMlpPolicy as MlpPolicy  # noqa: F401
