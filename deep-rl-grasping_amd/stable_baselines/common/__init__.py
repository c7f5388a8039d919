import random

import numpy as np


def set_global_seeds(seed):
    """Seeds Python / NumPy (the engine's device RNG is seeded through the model's ``seed``)."""
    random.seed(seed)
    np.random.seed(seed)
