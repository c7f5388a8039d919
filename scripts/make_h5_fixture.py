"""Generates tests/golden/keras_like_h5py.h5 (+ .npz with the expected contents) with the real HDF5 library:
run with an interpreter that has h5py (here: /opt/conda/bin/python3.9 scripts/make_h5_fixture.py).
The file mimics what Keras 2.2.4 `save_weights` writes for the reference's auto-encoder
(/root/reference/manipulation_main/gripperEnv/encoders.py:48): `layer_names` / `weight_names` fixed-length
string arrays, `backend` / `keras_version` as variable-length strings, nested layer groups, float32
datasets -- with small random tensors, plus a group with more than 8 members (several symbol-table nodes)
and integer / float64 / scalar datasets."""
import os

import h5py
import numpy as np

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
rng = np.random.default_rng(7)
expect = {}
with h5py.File(os.path.join(out, "keras_like_h5py.h5"), "w", libver="earliest") as f:
    f.attrs["layer_names"] = np.array([b"input_1", b"encoder", b"decoder"])
    f.attrs["backend"] = "tensorflow".encode("utf8")
    f.attrs["keras_version"] = "2.2.4".encode("utf8")
    f.create_group("input_1").attrs["weight_names"] = np.zeros((0,), np.float64)
    for layer, names in (("encoder", ["conv2d_1/kernel:0", "conv2d_1/bias:0", "dense_1/kernel:0", "dense_1/bias:0"]),
                         ("decoder", ["dense_2/kernel:0", "dense_2/bias:0", "conv2d_6/kernel:0", "conv2d_6/bias:0"])):
        g = f.create_group(layer)
        g.attrs["weight_names"] = np.array([n.encode() for n in names])
        for n in names:
            shape = {"conv2d_1/kernel:0": (3, 3, 1, 4), "conv2d_6/kernel:0": (3, 3, 4, 1), "dense_1/kernel:0": (16, 5),
                     "dense_2/kernel:0": (5, 16)}.get(n, (rng.integers(1, 6),))
            a = rng.normal(size=shape).astype(np.float32)
            g.create_dataset(n, data=a)
            expect[layer + "/" + n] = a
    many = f.create_group("many")
    for k in range(21):
        a = rng.integers(-5, 5, size=(k % 4 + 1, 2)).astype(np.int32 if k % 2 else np.float64)
        many.create_dataset("item_%02d" % k, data=a)
        expect["many/item_%02d" % k] = a
    many.attrs["scale"] = np.float64(0.25)
    many.attrs["ids"] = np.arange(5, dtype=np.int64)
    f.create_dataset("scalar", data=np.float32(3.5))
    expect["scalar"] = np.float32(3.5)
np.savez(os.path.join(out, "keras_like_h5py.npz"), **{k.replace("/", "|"): v for k, v in expect.items()})
print("wrote", len(expect), "datasets")
