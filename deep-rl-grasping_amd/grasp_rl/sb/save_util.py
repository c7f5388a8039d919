"""stable-baselines 2.10 model zip format: members ``data`` (JSON; values that are not
JSON-able are stored as {":type:", ":serialized:" base64(cloudpickle)} next to readable copies of
their attributes), ``parameter_list`` (JSON list of TF variable names) and ``parameters`` (npz).

Written by ``model.save`` (reference call sites sb_helper.py:241,244, base_callbacks.py:109) and read
by ``SAC.load`` (sb_helper.py:113, train_stable_baselines.py:96-104).  Zips shipped by the reference
under trained_models/ were pickled under Python 3.6/3.7 with gym + TF installed; whatever cannot be
unpickled here (bytecode of the learning-rate closure, gym spaces) is rebuilt from the readable copies.
"""
import base64
import io
import json
import pickle
import zipfile
from collections import OrderedDict

import numpy as np

from . import spaces as sp
from .vec_env import _CompatUnpickler


def _jsonable(v):
    try:
        json.dumps(v)
        return True
    except (TypeError, OverflowError):
        return False


def data_to_json(data):
    out = {}
    for k, v in data.items():
        if _jsonable(v):
            out[k] = v
            continue
        try:
            import cloudpickle
            raw = cloudpickle.dumps(v)
        except Exception:
            raw = pickle.dumps(v)
        entry = {":type:": str(type(v)), ":serialized:": base64.b64encode(raw).decode()}
        attrs = v.__dict__ if hasattr(v, "__dict__") else {}
        for ak, av in dict(attrs).items():
            if ak.startswith("_") and ak != "__module__":
                continue
            entry[ak] = av if _jsonable(av) else str(av)
        if isinstance(v, type):
            entry["__module__"] = v.__module__
            entry["__name__"] = v.__name__
        out[k] = entry
    return json.dumps(out, indent=4)


def _parse_array(text, dtype):
    return np.array(text.replace("[", " ").replace("]", " ").split(), dtype=np.float64).astype(dtype)


def _rebuild(entry):
    t = entry.get(":type:", "")
    if "Box" in t and "low" in entry:
        dtype = np.dtype(entry.get("dtype", "float32"))
        shape = tuple(entry["shape"])
        return sp.Box(_parse_array(entry["low"], dtype).reshape(shape), _parse_array(entry["high"], dtype).reshape(shape),
                      shape=shape, dtype=dtype)
    if "Discrete" in t and "n" in entry:
        return sp.Discrete(int(entry["n"]))
    return None


def json_to_data(text):
    raw = json.loads(text)
    out = {}
    for k, v in raw.items():
        if isinstance(v, dict) and ":serialized:" in v:
            obj = None
            try:
                obj = _CompatUnpickler(io.BytesIO(base64.b64decode(v[":serialized:"]))).load()
            except Exception:
                obj = _rebuild(v)
            if obj is not None and not (sp.is_box(obj) or sp.is_discrete(obj)) and ("Box" in v.get(":type:", "")):
                obj = _rebuild(v)
            out[k] = obj
        else:
            out[k] = v
    return out


def save_to_zip(path, data, params):
    if isinstance(path, str) and not path.endswith(".zip"):
        path = path + ".zip"
    names = list(params.keys())
    buf = io.BytesIO()
    np.savez(buf, **{n: np.asarray(params[n]) for n in names})
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("data", data_to_json(data))
        z.writestr("parameters", buf.getvalue())
        z.writestr("parameter_list", json.dumps(names))
    return path


def load_from_zip(path):
    if isinstance(path, str) and not path.endswith(".zip"):
        import os
        if not os.path.exists(path) and os.path.exists(path + ".zip"):
            path = path + ".zip"
    with zipfile.ZipFile(path) as z:
        data = json_to_data(z.read("data").decode())
        names = json.loads(z.read("parameter_list").decode())
        npz = np.load(io.BytesIO(z.read("parameters")))
        params = OrderedDict((n, np.array(npz[n])) for n in names)
    return data, params
