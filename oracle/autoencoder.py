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
