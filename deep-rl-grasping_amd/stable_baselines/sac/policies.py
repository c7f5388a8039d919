from grasp_rl.sb.policies import SacCnnPolicy as CnnPolicy  # noqa: F401
from grasp_rl.sb.policies import SacLnCnnPolicy as LnCnnPolicy  # noqa: F401
from grasp_rl.sb.policies import SacLnMlpPolicy as LnMlpPolicy  # noqa: F401
from grasp_rl.sb.policies import SacMlpPolicy as MlpPolicy  # noqa: F401
