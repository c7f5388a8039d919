"""Depth auto-encoder on the HIP engine: the call surface of
/root/reference/manipulation_main/gripperEnv/encoders.py (`SimpleAutoEncoder(config)`, `.train(inputs,
targets, batch_size, epochs, model_dir)` :40-50, `.test` :52-53, `.predict` :55-57, `.encode` :59-61,
`.load_weights(model_dir)` :26-30, `.encoding_shape` :63-65) for the network of :70-136 and
config/encoder.yaml.  Training (forward, MSE, backward, Keras-Adam) runs in libgrl.so (GRL_ALGO_AE);
this file owns the epoch loop Keras' `Model.fit` provides in the reference: shuffling, the 10 %
validation split, CSV history, best-weights checkpoint and EarlyStopping(patience=25).

Weight files: the reference stores Keras HDF5 (`model.h5`, written by `save_weights`, read back by
`load_weights`, encoders.py:27-31).  `grasp_rl.keras_h5` reads and writes that format without h5py, so the
encoders shipped under `encoder_files/` load directly and a model trained here is a `model.h5` the
reference's Keras code accepts; `model.npz` (keys = Keras weight names) is written next to it.
"""
import csv
import ctypes as C
import os
import weakref

import numpy as np

from . import _capi, keras_h5
from ._capi import check
from .engine import SacEngine

PARAM_NAMES = ["encoder/conv2d_1/kernel", "encoder/conv2d_1/bias", "encoder/conv2d_2/kernel", "encoder/conv2d_2/bias",
               "encoder/conv2d_3/kernel", "encoder/conv2d_3/bias", "encoder/dense_1/kernel", "encoder/dense_1/bias",
               "decoder/dense_2/kernel", "decoder/dense_2/bias", "decoder/conv2d_4/kernel", "decoder/conv2d_4/bias",
               "decoder/conv2d_5/kernel", "decoder/conv2d_5/bias", "decoder/conv2d_6/kernel", "decoder/conv2d_6/bias"]


def glorot_uniform_params(seed=0):
    """Keras defaults (encoders.py builds every layer with them): glorot_uniform kernels, zero biases."""
    rng = np.random.default_rng(seed)
    shapes = [(7, 7, 1, 32), (32,), (5, 5, 32, 32), (32,), (3, 3, 32, 32), (32,), (2048, 100), (100,),
              (100, 2048), (2048,), (3, 3, 32, 32), (32,), (5, 5, 32, 32), (32,), (7, 7, 32, 1), (1,)]
    P = {}
    for name, shp in zip(PARAM_NAMES, shapes):
        if name.endswith("bias"):
            P[name] = np.zeros(shp, np.float32)
        else:
            rf = int(np.prod(shp[:-2])) if len(shp) == 4 else 1
            lim = np.sqrt(6.0 / (shp[-2] * rf + shp[-1] * rf))
            P[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
    return P


class AeEngine(SacEngine):
    """Handle of a GRL_ALGO_AE engine: parameters = the 16 Keras tensors, one call = n minibatch updates."""

    def __init__(self, batch_size=128, lr=2e-4, act_batch=16, backend=None, lib_path=None, device="cuda:0"):
        super().__init__(_capi.make_ae_config(batch_size, lr, act_batch), backend=backend, lib_path=lib_path, device=device)

    def train_batches(self, imgs):
        """imgs [n_steps*B, 64, 64, 1] float32 (host): n_steps updates; returns the loss of the last one."""
        imgs = np.ascontiguousarray(imgs, dtype=np.float32).reshape(-1, 4096)
        if imgs.shape[0] % self.B:
            raise _capi.GrlError("number of images must be a multiple of the batch size %d" % self.B)
        d = self.be.to_device(imgs)
        check(self.lib, self.lib.grl_ae_train_step(self.h, C.c_void_p(self.be.ptr(d)), imgs.shape[0] // self.B))
        self._keep = [d]
        return self.metrics()["policy_loss"]

    def reconstruction(self):
        """Decoder output of the last training minibatch [B, 64, 64, 1]."""
        return self.fetch("out", (self.B, 64, 64, 1))

    def reconstruct(self, imgs):
        """Forward pass only (Keras `Model.predict`): imgs [N, 64, 64, 1] -> reconstructions [N, 64, 64, 1].
        The engine's batch is static: the last minibatch is padded with zeros."""
        imgs = np.ascontiguousarray(imgs, dtype=np.float32).reshape(-1, 4096)
        n = imgs.shape[0]
        out = np.empty((n, 4096), np.float32)
        for k0 in range(0, n, self.B):
            chunk = np.zeros((self.B, 4096), np.float32)
            m = min(self.B, n - k0)
            chunk[:m] = imgs[k0:k0 + m]
            d_in = self.be.to_device(chunk)
            d_out = self.be.to_device(np.empty((self.B, 4096), np.float32))
            check(self.lib, self.lib.grl_ae_reconstruct(self.h, C.c_void_p(self.be.ptr(d_in)), C.c_void_p(self.be.ptr(d_out))))
            out[k0:k0 + m] = self.be.to_host(d_out)[:m]
        return out.reshape(n, 64, 64, 1)


ENV_WORKER_VAR = "GRL_ENV_WORKER"      # set by grasp_rl.sb.vec_env while a fanned-out worker builds its environment
_LIVE = weakref.WeakSet()              # encoders of THIS process (the parent looks for the template env's here)
_DEFERRED = []                         # (config, model_dir) of every DeferredEncoder built in THIS (worker) process


def defer_in_this_process():
    """True inside an env worker of ``DummyVecEnv.fan_out`` (GRL_NUM_ENVS) unless GRL_BATCHED_ENCODER=0."""
    return os.environ.get(ENV_WORKER_VAR) == "1" and os.environ.get("GRL_BATCHED_ENCODER", "1") != "0"


def deferred_records():
    return [dict(r) for r in _DEFERRED]


def find_live_encoder(model_dir):
    """A ``SimpleAutoEncoder`` of this process holding the weights of `model_dir` (the template env's, built by the script
    before the model existed), or None."""
    want = os.path.realpath(os.path.expanduser(model_dir)) if model_dir else None
    for e in list(_LIVE):
        if want is not None and getattr(e, "model_dir", None) == want:
            return e
    return None


class SimpleAutoEncoder:
    def __new__(cls, config=None, *args, **kwargs):
        # Inside a fanned-out env worker the sensor's `encoders.SimpleAutoEncoder(config)` (sensor.py:190-192) yields the
        # deferred form: no HIP context and no batch-1 launch per environment -- the parent encodes all of them at once
        if cls is SimpleAutoEncoder and defer_in_this_process():
            return DeferredEncoder(config)
        return super().__new__(cls)

    def __init__(self, config, backend=None, lib_path=None, device="cuda:0", seed=0):
        net = config.get("network", [])
        want = [(32, 7, 2), (32, 5, 2), (32, 3, 2)]
        got = [(l["filters"], l["kernel_size"], l["strides"]) for l in net]
        if got != want or config.get("encoding_dim", 100) != 100 or config.get("alpha", 0.1) != 0.1:
            raise NotImplementedError("the HIP engine implements the reference's shipped network (config/encoder.yaml)")
        self.config = config
        self.act_batch = 16                # observations one grl_encode call takes (`ensure_act_batch`)
        self._mk = lambda bs: AeEngine(bs, float(config.get("learning_rate", 2e-4)), act_batch=self.act_batch, backend=backend,
                                       lib_path=lib_path, device=device)
        self.engine = None
        self._params = glorot_uniform_params(seed)
        self._bs = None
        self.model_dir = None              # where `load_weights` read from
        _LIVE.add(self)

    def ensure_act_batch(self, n):
        """At least n images per `encode` call (vectorised envs: all environments + their terminal observations at once)."""
        if n > self.act_batch:
            self.act_batch = int(n)
            if self.engine is not None:
                bs = self._bs
                self._params = self.engine.get_parameters()
                self.engine.close()
                self.engine, self._bs = None, None
                self._engine(bs)

    # ------------------------------------------------------------------ engine / weights
    def _engine(self, batch_size):
        if self.engine is None or self._bs != batch_size:
            if self.engine is not None:
                self._params = self.engine.get_parameters()
                self.engine.close()
            self.engine = self._mk(batch_size)
            self.engine.set_parameters(self._params)
            self._bs = batch_size
        return self.engine

    def get_weights(self):
        return self.engine.get_parameters() if self.engine is not None else dict(self._params)

    def set_weights(self, params):
        self.model_dir = None
        self._params = {k: np.asarray(params[k], np.float32) for k in PARAM_NAMES}
        if self.engine is not None:
            self.engine.set_parameters(self._params)

    def load_weights(self, model_dir):
        """encoders.py:26-30: weights from ``<model_dir>/model.h5`` (Keras HDF5); ``model.npz`` as a fallback."""
        model_dir = os.path.expanduser(model_dir)
        npz, h5 = os.path.join(model_dir, "model.npz"), os.path.join(model_dir, "model.h5")
        if os.path.exists(h5):
            w = keras_h5.read_keras_weights(h5)
            missing = [k for k in PARAM_NAMES if k + ":0" not in w]
            if missing:
                raise KeyError("%s lacks the auto-encoder weights %s" % (h5, missing))
            self.set_weights({k: w[k + ":0"] for k in PARAM_NAMES})
        elif os.path.exists(npz):
            with np.load(npz) as f:
                self.set_weights({k: f[k] for k in PARAM_NAMES})
        else:
            raise FileNotFoundError("neither model.h5 nor model.npz in %s" % model_dir)
        self.model_dir = os.path.realpath(model_dir)

    def save_weights(self, model_dir):
        """``model.h5`` in the layout Keras' ``save_weights`` gives this model (layers input_1 / encoder /
        decoder, weight names ``conv2d_1/kernel:0`` ...) plus ``model.npz``."""
        os.makedirs(model_dir, exist_ok=True)
        w = self.get_weights()
        np.savez(os.path.join(model_dir, "model.npz"), **w)
        keras_h5.write_keras_weights(os.path.join(model_dir, "model.h5"),
                                     {k + ":0": np.asarray(w[k], np.float32) for k in PARAM_NAMES},
                                     extra_layers=("input_1",))

    # ------------------------------------------------------------------ reference surface
    def train(self, inputs, targets, batch_size, epochs, model_dir, validation_split=0.1, patience=25, seed=0):
        """encoders.py:40-50: fit with 10 % validation (the LAST 10 % of the data, as Keras splits before
        shuffling), per-epoch shuffle, history.csv, best-val checkpoint, EarlyStopping(patience=25)."""
        if targets is not inputs and not np.array_equal(inputs, targets):
            raise NotImplementedError("the auto-encoder is trained to reconstruct its input")
        x = np.ascontiguousarray(inputs, np.float32).reshape(-1, 64, 64, 1)
        n_val = int(x.shape[0] * validation_split)
        xt, xv = x[: x.shape[0] - n_val], x[x.shape[0] - n_val:]
        if xt.shape[0] < batch_size:
            raise ValueError("%d training images after the validation split, fewer than one batch of %d: the device "
                             "batch is static (Keras would train on the partial batch)" % (xt.shape[0], batch_size))
        eng = self._engine(batch_size)
        rng = np.random.default_rng(seed)
        os.makedirs(model_dir, exist_ok=True)
        hist = {"loss": [], "val_loss": []}
        best, since = np.inf, 0
        with open(os.path.join(model_dir, "history.csv"), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["epoch", "loss", "val_loss"])
            for ep in range(epochs):
                order = rng.permutation(xt.shape[0])
                n_full = (xt.shape[0] // batch_size) * batch_size      # the engine's batch is static: drop the remainder
                losses = []
                for k0 in range(0, n_full, batch_size):
                    losses.append(eng.train_batches(xt[order[k0:k0 + batch_size]]))
                loss = float(np.mean(losses)) if losses else float("nan")
                val = self.test(xv, xv) if n_val else loss
                hist["loss"].append(loss); hist["val_loss"].append(val)
                wr.writerow([ep, loss, val]); f.flush()
                if val < best:
                    best, since = val, 0
                    self.save_weights(model_dir)
                else:
                    since += 1
                    if since >= patience:          # Keras EarlyStopping: stop when `wait >= patience`
                        break
        return hist

    def predict(self, imgs):
        """encoders.py:55-57: reconstructions [N, 64, 64, 1] -- the forward half of the training graph on the device."""
        eng = self.engine if self.engine is not None else self._engine(self._bs or 128)
        return eng.reconstruct(imgs)

    def test(self, inputs, targets):
        out = self.predict(inputs)
        return float(np.mean((out - np.asarray(targets, np.float32).reshape(out.shape)) ** 2))

    def encode(self, imgs):
        eng = self.engine if self.engine is not None else self._engine(self._bs or 128)
        imgs = np.ascontiguousarray(imgs, np.float32).reshape(-1, 64, 64, 1)
        nb = int(eng.cfg.act_batch)
        return np.concatenate([eng.encode(imgs[k:k + nb]) for k in range(0, imgs.shape[0], nb)], axis=0)

    @property
    def encoding_shape(self):
        return (100,)


class DeferredEncoder:
    """What an ENV WORKER holds in place of the encoder when environments are vectorised.  The
    reference's sensor calls ``self._encoder.encode(img)`` with a batch of one inside every environment
    (gripperEnv/sensor.py:220-222); with N environments in N worker processes that would be N HIP contexts and N
    batch-1 launches per step.  This object keeps the sensor code unchanged -- ``encode`` returns the (filtered) depth
    image itself, flattened to 4096 floats, and ``encoding_shape`` says so -- and
    ``grasp_rl.sb.vec_env.VecBatchedEncoder`` in the parent process encodes the images of ALL environments in ONE
    ``grl_encode`` call per step.  No GPU, no libgrl in the worker.

    Under ``GRL_NUM_ENVS`` nobody writes this by hand: inside a fanned-out worker ``SimpleAutoEncoder(config)`` -- what the
    reference's sensor constructs (sensor.py:190-192) -- IS this object, ``load_weights(model_dir)`` records where the weights
    are, and ``DummyVecEnv.fan_out`` asks the workers for those records and puts the batched encoder in front of them."""
    encoding_shape = (64 * 64,)

    def __init__(self, config=None):
        self.config = dict(config) if config else None
        self.model_dir = None
        self._record = {"config": self.config, "model_dir": None}
        _DEFERRED.append(self._record)

    def load_weights(self, model_dir):
        self.model_dir = os.path.realpath(os.path.expanduser(model_dir))
        self._record["model_dir"] = self.model_dir

    def encode(self, imgs):
        return np.ascontiguousarray(imgs, np.float32).reshape(-1, 64 * 64)

    def predict(self, imgs):      # (only the sensor's visualisation calls this)
        raise RuntimeError("DeferredEncoder does not reconstruct: visualise through the parent's SimpleAutoEncoder")
