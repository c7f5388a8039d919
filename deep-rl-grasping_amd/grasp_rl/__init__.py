"""grasp_rl -- MI355X-native SAC / DQN / BDQ update path behind the stable-baselines surface that
BarisYazici/deep-rl-grasping drives.  The compute lives in libgrl.so (HIP,
gfx950); this package is the Python host: arenas, configuration, the stable-baselines-shaped model
objects and the environment-side wrappers."""
from ._capi import GrlError, make_config  # noqa: F401

__all__ = ["GrlError", "make_config"]
