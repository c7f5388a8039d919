"""Import-path alias: lets the reference's scripts (``import stable_baselines as sb`` in
train_stable_baselines.py:7, sb_helper.py:6, base_callbacks.py:11-14) run unmodified on top of the
MI355X engine.  Put ``deep-rl-grasping_amd/`` on PYTHONPATH *instead of* installing stable-baselines
(see INTEGRATION.md).  Only the SAC / DQN / BDQ update paths are implemented; the on-policy algorithms
the reference can also select (TRPO / PPO2 / DDPG, sb_helper.py:130-173) are out of scope and say so."""
from grasp_rl.sb import logger  # noqa: F401
from grasp_rl.sb.sac import SAC  # noqa: F401

__version__ = "2.10.1+grasp_rl"


def _unsupported(name):
    class _Unsupported:
        def __init__(self, *a, **k):
            raise NotImplementedError("%s is outside the accelerated hot path (SAC / DQN / BDQ); "
                                      "install stable-baselines itself to use it" % name)

        @classmethod
        def load(cls, *a, **k):
            raise NotImplementedError("%s is outside the accelerated hot path" % name)
    _Unsupported.__name__ = name
    return _Unsupported


TRPO, PPO2, DDPG, A2C, ACKTR, TD3 = (_unsupported(n) for n in ("TRPO", "PPO2", "DDPG", "A2C", "ACKTR", "TD3"))
try:
    from grasp_rl.sb.dqn import BDQ, DQN  # noqa: F401
except ImportError:       # pragma: no cover
    DQN, BDQ = _unsupported("DQN"), _unsupported("BDQ")
