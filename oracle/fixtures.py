"""Readers for the artefacts the reference ships.

TEST INFRASTRUCTURE (see oracle/__init__.py).  None of the reference's dependencies
(stable_baselines, tensorflow, keras, h5py, gym) are importable here, so each format is
read from first principles:

* stable-baselines zip  (``sb_helper.py:228-247`` writes them via ``model.save``):
  zipfile with members ``data`` (JSON), ``parameter_list`` (JSON) and ``parameters`` (npz).
* ``vecnormalize.pkl``   (``sb_helper.py:247``, ``base_callbacks.py:139-149``): a pickled
  ``stable_baselines.common.vec_env.VecNormalize``; unpickled with placeholder classes.
* Keras ``model.h5``     (``encoders.py:27-31`` ``load_weights``): HDF5 v0 file whose 16
  datasets are contiguous little-endian f32; located by scanning for layout messages.
"""
import io
import json
import pickle
import struct
import zipfile
from collections import OrderedDict

import numpy as np


# --------------------------------------------------------------------------- SB zip
def load_sb_zip(path):
    """Return (data_dict, OrderedDict name -> ndarray) of a stable-baselines 2.10 zip."""
    with zipfile.ZipFile(path) as z:
        data = json.loads(z.read("data").decode())
        names = json.loads(z.read("parameter_list").decode())
        npz = np.load(io.BytesIO(z.read("parameters")))
        params = OrderedDict((n, np.array(npz[n])) for n in names)
    return data, params


# --------------------------------------------------------------------------- pickle
class _Stub:
    """Placeholder for classes from modules that are not importable here."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state


class _StubUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("numpy"):
            module = module.replace("numpy.core", "numpy._core")
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                module = module.replace("numpy._core", "numpy.core")
                return super().find_class(module, name)
        if module in ("builtins", "collections", "copyreg", "_codecs"):
            return super().find_class(module, name)
        return type(name, (_Stub,), {"__module__": module})


def load_vecnormalize_pkl(path):
    """Return a dict with the VecNormalize state (obs_rms/ret_rms mean,var,count, clips...)."""
    with open(path, "rb") as f:
        obj = _StubUnpickler(f).load()
    d = obj.__dict__
    out = {}
    for k in ("clip_obs", "clip_reward", "gamma", "epsilon", "training", "norm_obs",
              "norm_reward", "num_envs"):
        if k in d:
            out[k] = d[k]
    for rms in ("obs_rms", "ret_rms"):
        r = d[rms].__dict__
        out[rms] = {"mean": np.asarray(r["mean"], dtype=np.float64),
                    "var": np.asarray(r["var"], dtype=np.float64),
                    "count": float(r["count"])}
    if "old_obs" in d:
        out["old_obs"] = np.asarray(d["old_obs"])
    return out


# --------------------------------------------------------------------------- Keras h5
_AE_SHAPES = OrderedDict([
    # name                        shape              (Keras 2.2.4 save_weights of encoders.py:90-124)
    ("encoder/conv2d_1/kernel", (7, 7, 1, 32)),
    ("encoder/conv2d_1/bias", (32,)),
    ("encoder/conv2d_2/kernel", (5, 5, 32, 32)),
    ("encoder/conv2d_2/bias", (32,)),
    ("encoder/conv2d_3/kernel", (3, 3, 32, 32)),
    ("encoder/conv2d_3/bias", (32,)),
    ("encoder/dense_1/kernel", (2048, 100)),
    ("encoder/dense_1/bias", (100,)),
    ("decoder/dense_2/kernel", (100, 2048)),
    ("decoder/dense_2/bias", (2048,)),
    ("decoder/conv2d_4/kernel", (3, 3, 32, 32)),
    ("decoder/conv2d_4/bias", (32,)),
    ("decoder/conv2d_5/kernel", (5, 5, 32, 32)),
    ("decoder/conv2d_5/bias", (32,)),
    ("decoder/conv2d_6/kernel", (7, 7, 32, 1)),
    ("decoder/conv2d_6/bias", (1,)),
])


def _scan_contiguous_datasets(buf):
    """Find (addr, size) of every contiguous-layout dataset: layout message v3 class 1."""
    found = []
    pos = 0
    while True:
        pos = buf.find(b"\x03\x01", pos)
        if pos < 0:
            break
        if pos + 18 <= len(buf):
            addr, size = struct.unpack_from("<QQ", buf, pos + 2)
            if 0 < addr < len(buf) and 0 < size <= len(buf) and addr + size <= len(buf) \
                    and size % 4 == 0:
                found.append((addr, size))
        pos += 1
    return found


def load_keras_ae_h5(path):
    """Return OrderedDict name -> f32 ndarray for the 16 auto-encoder weight tensors.

    The datasets are matched to names by byte size *and* by the HDF5 object-name order: the
    three 128-byte conv biases and the two 32x32 3x3 / 5x5 kernels share sizes between
    encoder and decoder, so the match is disambiguated with the offsets listed in
    SURVEY.md B.3 (identical for all four shipped files) and cross-checked by the
    reconstruction test in scripts/make_golden.py.
    """
    buf = open(path, "rb").read()
    cands = _scan_contiguous_datasets(buf)
    by_addr = {a: s for a, s in cands}
    # offsets from SURVEY.md B.3 (verified there with h5py under /opt/conda/bin/python3.9)
    addr = {
        "encoder/conv2d_1/kernel": 8976, "encoder/conv2d_1/bias": 17296,
        "encoder/conv2d_2/kernel": 17424, "encoder/conv2d_2/bias": 119824,
        "encoder/conv2d_3/kernel": 122000, "encoder/conv2d_3/bias": 158864,
        "encoder/dense_1/kernel": 162960, "encoder/dense_1/bias": 158992,
        "decoder/dense_2/kernel": 984208, "decoder/dense_2/bias": 1803408,
        "decoder/conv2d_4/kernel": 1813648, "decoder/conv2d_4/bias": 159392,
        "decoder/conv2d_5/kernel": 1852560, "decoder/conv2d_5/bias": 159520,
        "decoder/conv2d_6/kernel": 1957008, "decoder/conv2d_6/bias": 159648,
    }
    out = OrderedDict()
    for name, shape in _AE_SHAPES.items():
        n = int(np.prod(shape))
        a = addr[name]
        if by_addr.get(a) != 4 * n:
            raise ValueError("%s: no contiguous %d-byte dataset at offset %d in %s"
                             % (name, 4 * n, a, path))
        out[name] = np.frombuffer(buf, dtype="<f4", count=n, offset=a).reshape(shape).copy()
    return out
