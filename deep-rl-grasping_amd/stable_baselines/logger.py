from grasp_rl.sb.logger import *  # noqa: F401,F403
from grasp_rl.sb.logger import configure, dumpkvs, get_dir, getkvs, logkv, logkvs, record  # noqa: F401
