"""CPU restatement of the Keras depth auto-encoder forward.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Follows ``/root/reference/manipulation_main/gripperEnv/encoders.py:70-136`` (network
definition: ``_build`` :70-136, encoder half :90-108, decoder :110-124), hyper-parameters
``config/encoder.yaml:2-7`` (filters 32, kernels 7/5/3, strides 2, encoding 100, LeakyReLU
alpha 0.1 at ``encoders.py:87``) and the call site ``sensor.py:206-222`` (mask filtering,
reshape to (1,64,64,1), ``encoder.encode``).  Keras 2.2.4 / TF 1.14 semantics restated:
Conv2D(padding='same') uses TensorFlow's asymmetric padding, Flatten is NHWC row-major.
"""
import numpy as np
import torch
import torch.nn.functional as F

LEAKY_ALPHA = 0.1          # encoders.py:87
KERNELS = (7, 5, 3)        # config/encoder.yaml
STRIDE = 2


def tf_same_pad(n_in, k, s):
    """TensorFlow 'SAME': out = ceil(in/s); total = max((out-1)*s + k - in, 0); lo = total//2."""
    out = -(-n_in // s)
    tot = max((out - 1) * s + k - n_in, 0)
    return tot // 2, tot - tot // 2


def _conv_same(x_nhwc, w_hwio, b, stride):
    k = w_hwio.shape[0]
    lo, hi = tf_same_pad(x_nhwc.shape[1], k, stride)
    x = x_nhwc.permute(0, 3, 1, 2)
    x = F.pad(x, (lo, hi, lo, hi))
    y = F.conv2d(x, w_hwio.permute(3, 2, 0, 1), stride=stride)
    return y.permute(0, 2, 3, 1) + b.reshape(1, 1, 1, -1)


def filter_depth(depth, mask, robot_id, on_table=True):
    """sensor.py:212-217: zero the plane (id 0), the robot and -- OnTable scene -- table (1) and
    tray (2) pixels of the depth image before encoding."""
    drop = [0, robot_id] + ([1, 2] if on_table else [])
    return np.where(np.isin(mask, np.asarray(drop)), 0.0, depth).astype(np.float32)


def encode(W, depth_nhwc):
    """encoders.py:90-108 + Encoder.encode :59-61.  depth_nhwc [N,64,64,1] float32 -> [N,100]."""
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    x = t(depth_nhwc)
    for i in (1, 2, 3):
        x = _conv_same(x, t(W["encoder/conv2d_%d/kernel" % i]), t(W["encoder/conv2d_%d/bias" % i]),
                       STRIDE)
        x = F.leaky_relu(x, LEAKY_ALPHA)
    flat = x.reshape(x.shape[0], -1)                         # Flatten, NHWC: (h*8+w)*32+c
    z = flat @ t(W["encoder/dense_1/kernel"]) + t(W["encoder/dense_1/bias"])
    return F.leaky_relu(z, LEAKY_ALPHA).numpy()


def decode(W, z):
    """encoders.py:110-124 (used only to pin the conventions through reconstruction MSE)."""
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    x = t(z) @ t(W["decoder/dense_2/kernel"]) + t(W["decoder/dense_2/bias"])
    x = F.leaky_relu(x, LEAKY_ALPHA).reshape(-1, 8, 8, 32)
    for i in (4, 5):
        x = x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)     # UpSampling2D(2)
        x = _conv_same(x, t(W["decoder/conv2d_%d/kernel" % i]), t(W["decoder/conv2d_%d/bias" % i]), 1)
        x = F.leaky_relu(x, LEAKY_ALPHA)
    x = x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    x = _conv_same(x, t(W["decoder/conv2d_6/kernel"]), t(W["decoder/conv2d_6/bias"]), 1)
    return x.numpy()


# ---------------------------------------------------------------------------------------------------
# Training step (encoders.py:40-50 `Encoder.train` -> keras `Model.fit`; loss 'mean_squared_error'
# :127, `Adam(lr=config['learning_rate'])` :130, config/encoder.yaml:8-10: lr 2e-4, batch 128).
# Keras 2.2.4 Adam [un-vendored dependency, setup.py:12]: lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
# m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t * m / (sqrt(v) + 1e-7).  parity unpinned.
PARAM_ORDER = ["encoder/conv2d_1/kernel", "encoder/conv2d_1/bias", "encoder/conv2d_2/kernel", "encoder/conv2d_2/bias",
               "encoder/conv2d_3/kernel", "encoder/conv2d_3/bias", "encoder/dense_1/kernel", "encoder/dense_1/bias",
               "decoder/dense_2/kernel", "decoder/dense_2/bias", "decoder/conv2d_4/kernel", "decoder/conv2d_4/bias",
               "decoder/conv2d_5/kernel", "decoder/conv2d_5/bias", "decoder/conv2d_6/kernel", "decoder/conv2d_6/bias"]
PARAM_SHAPES = [(7, 7, 1, 32), (32,), (5, 5, 32, 32), (32,), (3, 3, 32, 32), (32,), (2048, 100), (100,),
                (100, 2048), (2048,), (3, 3, 32, 32), (32,), (5, 5, 32, 32), (32,), (7, 7, 32, 1), (1,)]


def init_params(seed=0):
    """Keras defaults: glorot_uniform kernels (fan_in/out over the receptive field), zero biases."""
    rng = np.random.default_rng(seed)
    P = {}
    for name, shp in zip(PARAM_ORDER, PARAM_SHAPES):
        if name.endswith("bias"):
            P[name] = np.zeros(shp, np.float32)
        else:
            rf = int(np.prod(shp[:-2])) if len(shp) == 4 else 1
            fan_in, fan_out = shp[-2] * rf, shp[-1] * rf
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            P[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
    return P


class AeOracle:
    def __init__(self, params, lr=2e-4, eps=1e-7):
        self.P = {k: torch.tensor(np.asarray(v, np.float32), requires_grad=True) for k, v in params.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.lr, self.eps, self.t = lr, eps, 0

    def forward(self, x):
        P = self.P
        h = x
        acts = {}
        for i in (1, 2, 3):
            h = F.leaky_relu(_conv_same(h, P["encoder/conv2d_%d/kernel" % i], P["encoder/conv2d_%d/bias" % i], STRIDE), LEAKY_ALPHA)
            acts["e%d" % i] = h
        z = F.leaky_relu(h.reshape(h.shape[0], -1) @ P["encoder/dense_1/kernel"] + P["encoder/dense_1/bias"], LEAKY_ALPHA)
        acts["z"] = z
        h = F.leaky_relu(z @ P["decoder/dense_2/kernel"] + P["decoder/dense_2/bias"], LEAKY_ALPHA).reshape(-1, 8, 8, 32)
        for i in (4, 5):
            h = h.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
            h = F.leaky_relu(_conv_same(h, P["decoder/conv2d_%d/kernel" % i], P["decoder/conv2d_%d/bias" % i], 1), LEAKY_ALPHA)
            acts["d%d" % i] = h
        h = h.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
        out = _conv_same(h, P["decoder/conv2d_6/kernel"], P["decoder/conv2d_6/bias"], 1)
        return out, acts

    def step(self, x_np):
        """One minibatch: returns {'loss', 'out', 'z', 'grads'} and applies the Keras-Adam update."""
        x = torch.from_numpy(np.asarray(x_np, np.float32))
        out, acts = self.forward(x)
        loss = torch.mean((out - x) ** 2)
        grads = torch.autograd.grad(loss, list(self.P.values()))
        res = {"loss": float(loss.detach()), "out": out.detach().numpy().copy(), "z": acts["z"].detach().numpy().copy(),
               "grads": {k: g.numpy().copy() for k, g in zip(self.P, grads)}}
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - 0.999 ** self.t) / (1.0 - 0.9 ** self.t)
        with torch.no_grad():
            for (k, p), g in zip(self.P.items(), grads):
                self.m[k].mul_(0.9).add_(g, alpha=0.1)
                self.v[k].mul_(0.999).addcmul_(g, g, value=0.001)
                p.sub_(np.float32(lr_t) * self.m[k] / (torch.sqrt(self.v[k]) + self.eps))
        return res

    def params(self):
        return {k: v.detach().numpy().copy() for k, v in self.P.items()}
