"""libgrl.so loads and exports every entry point include/grl.h declares (no compute without a GPU)."""
import ctypes
import os
import re

from grasp_rl import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "grl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(grl_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(_capi.DEFAULT_LIB)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libgrl.so does not export %s" % n
    for n in _capi.EXPORTS:
        assert n in names, n


def test_query_sizes_is_pure_host_and_scales_with_capacity():
    lib = _capi.load_library()
    small, big = _capi.GrlSizes(), _capi.GrlSizes()
    cfg = _capi.make_config("augmented", obs_channels=2, n_direct=1, batch_size=256, replay_capacity=1000)
    _capi.check(lib, lib.grl_query_sizes(ctypes.byref(cfg), ctypes.byref(small)))
    cfg.replay_capacity = 1_000_000                        # BASELINE config 4 scale: must fit 288 GB HBM
    _capi.check(lib, lib.grl_query_sizes(ctypes.byref(cfg), ctypes.byref(big)))
    assert small.n_trainable == big.n_trainable and small.n_trainable >= 1342990   # + 16-byte alignment padding
    assert big.replay_bytes > 30e9 and big.replay_bytes < 288e9 and big.replay_bytes > 900 * small.replay_bytes
    assert small.grads_bytes >= 4 * small.n_trainable


def test_bad_config_is_rejected_with_a_message():
    lib = _capi.load_library()
    cfg = _capi.make_config("augmented", obs_channels=2, n_direct=1, batch_size=0)
    s = _capi.GrlSizes()
    assert lib.grl_query_sizes(ctypes.byref(cfg), ctypes.byref(s)) < 0
    assert b"batch_size" in lib.grl_last_error()
