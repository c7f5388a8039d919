"""Parameter initialisation of the stable-baselines policies the reference selects
(/root/reference/manipulation_main/training/sb_helper.py:85-96): ``ortho_init(sqrt 2)`` for the conv /
``cnn_fc1`` layers of the extractor (custom_obs_policy.py:34-40 passes init_scale=np.sqrt(2)),
Glorot-uniform for the ``tf.layers.dense`` heads, zero biases, ``log_ent_coef = log(1.0)``; the
target network starts as a copy of ``model/values_fn``."""
from collections import OrderedDict

import numpy as np


def _ortho(shape, scale, rng):
    flat = shape if len(shape) == 2 else (int(np.prod(shape[:-1])), shape[-1])
    u, _, vt = np.linalg.svd(rng.normal(0.0, 1.0, flat), full_matrices=False)
    q = (u if u.shape == flat else vt).reshape(shape)
    return (scale * q).astype(np.float32)


def init_parameters(table, seed=0):
    """table: [(name, offset, numel, shape, trainable)] from the engine; returns name -> ndarray."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, _, _, shape, _ in table:
        if name.startswith("target/"):
            continue
        if name.endswith("/w:0"):
            out[name] = _ortho(tuple(shape), np.sqrt(2.0), rng)
        elif name.endswith("/kernel:0"):
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = rng.uniform(-lim, lim, shape).astype(np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    for name, _, _, shape, _ in table:
        if name.startswith("target/"):
            out[name] = out["model" + name[len("target"):]].copy()
    return OrderedDict((n, out[n]) for n, *_ in table)
