from grasp_rl.sb.callbacks import (BaseCallback, CallbackList, CheckpointCallback, ConvertCallback,  # noqa: F401
                                   EvalCallback, EventCallback)
