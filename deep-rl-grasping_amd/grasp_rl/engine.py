"""SacEngine -- thin Python owner of the device arenas + handle of libgrl.so.

The arithmetic of the SAC update (replay gather/normalise, CNN + MLP forward/backward, losses, Adam,
Polyak) runs in the HIP kernels behind the C ABI of include/grl.h; PyTorch-ROCm only provides the
device memory (four flat tensors), the stream and -- for data parallelism -- the RCCL all-reduce of
the flat gradient tensor.  There is no CPU fallback: constructing an engine without a HIP device or
without the built library raises.

Reference call sites this object stands behind: sb.SAC(...) construction and .learn()'s per-step
update at /root/reference/manipulation_main/training/sb_helper.py:104-128,175-177.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _capi
from ._capi import GrlError, check


class TorchCudaBackend:
    """Device memory = PyTorch-ROCm tensors; work is enqueued on a dedicated HIP stream."""

    def __init__(self, device="cuda:0"):
        import torch
        if not torch.cuda.is_available():
            raise GrlError("no HIP device visible to PyTorch: the grasp_rl engine runs only on an AMD GPU "
                           "(MI355X / gfx950); there is no CPU fallback")
        self.torch = torch
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)

    def alloc_f32(self, nbytes):
        t = self.torch.zeros((nbytes + 3) // 4, dtype=self.torch.float32, device=self.device)
        self.torch.cuda.synchronize(self.device)
        return t

    def ptr(self, t):
        return t.data_ptr()

    def stream_ptr(self):
        return self.stream.cuda_stream

    def to_device(self, arr):
        # allocated and filled ON the engine stream: the caching allocator then only hands the block to someone
        # else after work queued on this stream (the kernels that read it) has been ordered before the reuse
        with self.torch.cuda.stream(self.stream):
            t = self.torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
        return t

    def to_host(self, t):
        self.stream.synchronize()
        return t.detach().cpu().numpy()

    def write(self, view, arr):
        with self.torch.cuda.stream(self.stream):
            view.copy_(self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).reshape(view.shape),
                       non_blocking=False)

    def synchronize(self):
        self.stream.synchronize()

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def comm_fork(self):
        """Marks the current end of the engine stream: a later `comm_context(fork)` starts its work there."""
        ev = self.torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def comm_context(self, fork=None):
        """Context whose work runs on a second stream, ordered after `fork` (default: everything enqueued on the engine
        stream so far): the exchange of a finished gradient bucket, overlapping what the engine stream does next."""
        if getattr(self, "comm", None) is None:
            self.comm = self.torch.cuda.Stream(device=self.device)
        if fork is None:
            self.comm.wait_stream(self.stream)
        else:
            self.comm.wait_event(fork)
        return self.torch.cuda.stream(self.comm)

    def comm_join(self):
        """Engine stream waits for the second stream."""
        if getattr(self, "comm", None) is not None:
            self.stream.wait_stream(self.comm)

    def as_torch(self, t):
        return t


class SacEngine:
    def __init__(self, cfg, backend=None, lib_path=None, device="cuda:0"):
        self.lib = _capi.load_library(lib_path)
        self.be = backend if backend is not None else TorchCudaBackend(device)
        self.cfg = cfg
        sizes = _capi.GrlSizes()
        check(self.lib, self.lib.grl_query_sizes(C.byref(cfg), C.byref(sizes)))
        self.sizes = sizes
        self.state = self.be.alloc_f32(sizes.state_bytes)
        self.grads = self.be.alloc_f32(sizes.grads_bytes)
        self.work = self.be.alloc_f32(sizes.work_bytes)
        self.replay = self.be.alloc_f32(sizes.replay_bytes)
        bufs = _capi.GrlBuffers(self.be.ptr(self.state), self.be.ptr(self.grads), self.be.ptr(self.work),
                                self.be.ptr(self.replay))
        h = C.c_void_p()
        check(self.lib, self.lib.grl_create(C.byref(cfg), C.byref(bufs), C.byref(h)))
        self.h = h
        check(self.lib, self.lib.grl_set_stream(self.h, C.c_void_p(self.be.stream_ptr())))
        self.table = _capi.param_table(self.lib, self.h)
        self.n_trainable = sizes.n_trainable
        self.B, self.A = cfg.batch_size, cfg.act_dim
        self.observe_rows = int(cfg.act_batch)      # one env step that `observe` + `act(observed=True)` cover in one call
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.grl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def _view(self, off, numel, shape):
        return self.state[off:off + numel].reshape(shape if len(shape) else ())

    def param_names(self, trainable_only=False):
        return [n for n, _, _, _, tr in self.table if tr or not trainable_only]

    def get_parameters(self):
        """OrderedDict TF-name -> float32 ndarray (SB ``get_parameters``, sb_helper.py:114)."""
        self.be.synchronize()
        out = OrderedDict()
        flat = self.be.to_host(self.state[: self.sizes.n_params])
        for name, off, numel, shape, _ in self.table:
            out[name] = flat[off:off + numel].reshape(shape).copy()
        return out

    def set_parameters(self, params, exact_match=True):
        """SB ``load_parameters(params, exact_match)`` (sb_helper.py:115,198,225)."""
        known = {n for n, *_ in self.table}
        if exact_match and set(params.keys()) != known:
            raise GrlError("parameter names do not match: missing %s, unexpected %s" %
                           (sorted(known - set(params)), sorted(set(params) - known)))
        for name, off, numel, shape, _ in self.table:
            if name not in params:
                continue
            arr = np.asarray(params[name], dtype=np.float32)
            if int(arr.size) != numel:
                raise GrlError("%s: expected %d values, got shape %s" % (name, numel, arr.shape))
            self.be.write(self._view(off, numel, shape), arr.reshape(shape))
        self.be.synchronize()

    def reset_optimizer(self):
        check(self.lib, self.lib.grl_reset_optimizer(self.h))

    def grad_tensor(self):
        """Flat fp32 gradient bucket (the tensor a data-parallel wrapper all-reduces)."""
        return self.grads[: self.n_trainable]

    def get_gradients(self):
        flat = self.be.to_host(self.grads[: self.n_trainable])
        out = OrderedDict()
        for name, off, numel, shape, tr in self.table:
            if tr:
                out[name] = flat[off:off + numel].reshape(shape).copy()
        return out

    # ------------------------------------------------------------------ data
    def set_obs_stats(self, mean, var, ret_var):
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        var = np.ascontiguousarray(var, dtype=np.float64)
        check(self.lib, self.lib.grl_set_obs_stats(self.h, mean.ctypes.data, var.ctypes.data, float(ret_var)))

    # ---- data parallel with the exchange inside the update graph (grl_allreduce_*, csrc/dp_kernels.h)
    HANDLE_BYTES = 128          # include/grl.h: GRL_ALLREDUCE_HANDLE_BYTES (two 64-byte IPC handles: flags, data)

    def allreduce_init(self, rank, world):
        """Allocates this rank's exchange memory; returns its 128 bytes of IPC handles for the out-of-band exchange."""
        buf = C.create_string_buffer(self.HANDLE_BYTES)
        check(self.lib, self.lib.grl_allreduce_init(self.h, int(rank), int(world), buf))
        return buf.raw

    def allreduce_connect(self, handles):
        """handles: the 128-byte handle blobs of all ranks, in rank order."""
        blob = b"".join(handles)
        check(self.lib, self.lib.grl_allreduce_connect(self.h, C.c_char_p(blob)))

    def allreduce_disconnect(self):
        """Back to a single-process handle: peers unmapped, exchange memory freed (grl_allreduce_disconnect).  The caller
        makes sure no exchange is in flight on any rank (DataParallelInGraph.close)."""
        check(self.lib, self.lib.grl_allreduce_disconnect(self.h))

    def allreduce_set_overlap(self, on=True):
        """Exchange the dense bucket on a side lane of the update's graph while the convolution backward runs
        (include/grl.h: grl_allreduce_set_overlap).  Raises when the configuration has no staged plan."""
        check(self.lib, self.lib.grl_allreduce_set_overlap(self.h, 1 if on else 0))

    MODES = {"auto": 0, "twoshot": 1, "oneshot": 2}

    def allreduce_set_mode(self, mode="auto"):
        """'auto' (one-shot for world <= 2), 'twoshot', 'oneshot' (include/grl.h: grl_allreduce_set_mode).  All ranks alike,
        while no exchange is in flight."""
        check(self.lib, self.lib.grl_allreduce_set_mode(self.h, self.MODES[mode]))

    def train_allreduce(self, n_steps=1, idx=None, eps=None):
        pi, pe, keep = self._noise(idx, eps, n_steps)
        check(self.lib, self.lib.grl_train_step_allreduce(self.h, n_steps, pi, pe))
        self._keep = [keep]

    def allreduce_set_timeout(self, ms):
        """Bound of every wait for a peer from now on, in ms (0: the default again); include/grl.h: grl_allreduce_set_timeout."""
        check(self.lib, self.lib.grl_allreduce_set_timeout(self.h, int(ms)))

    def allreduce_status(self):
        n, err = C.c_int64(), C.c_int()
        check(self.lib, self.lib.grl_allreduce_status(self.h, C.byref(n), C.byref(err)))
        return n.value

    # ---- VecNormalize running statistics kept on the device (grl_norm_update; SAC handles)
    def norm_update(self, obs):
        """RunningMeanStd.update(obs) of one env step's raw observations [n, ...] on the device (stream-ordered)."""
        obs = np.ascontiguousarray(obs, dtype=np.float32)
        check(self.lib, self.lib.grl_norm_update(self.h, obs.ctypes.data, obs.shape[0]))

    def observe(self, obs, update_stats=False):
        """Upload one env step's raw observations [n, ...] ONCE (grl_observe): `act(observed=True)` and
        `replay_add_observed` read the device copy; update_stats folds them into the running statistics as `norm_update`
        does.  Returns the serial number of this call (consecutive calls: consecutive numbers)."""
        obs = np.ascontiguousarray(obs, dtype=np.float32)
        check(self.lib, self.lib.grl_observe(self.h, obs.ctypes.data, obs.shape[0], 1 if update_stats else 0))
        self._observe_serial = getattr(self, "_observe_serial", 0) + 1
        return self._observe_serial

    def replay_add_observed(self, act, rew, done, term_rows=None, term_obs=None):
        """The transitions between the last two `observe` calls.  term_rows / term_obs: rows whose episode ended and the
        terminal observations stored as their next observation."""
        rew = np.ascontiguousarray(rew, dtype=np.float32).reshape(-1)
        n = rew.shape[0]
        act = np.ascontiguousarray(act, dtype=np.float32).reshape(n, self.A)
        done = np.ascontiguousarray(done, dtype=np.float32).reshape(n)
        pr = po = None
        n_term = 0
        if term_rows is not None and len(term_rows):
            tr = np.ascontiguousarray(term_rows, dtype=np.int32)
            to = np.ascontiguousarray(term_obs, dtype=np.float32)
            n_term, pr, po = tr.shape[0], tr.ctypes.data, to.ctypes.data
        check(self.lib, self.lib.grl_replay_add_observed(self.h, act.ctypes.data, rew.ctypes.data, done.ctypes.data, n, pr, po, n_term))

    def set_running_stats(self, mean, var, count):
        """Starting point of the running statistics grl_norm_update continues from (env layout, float64)."""
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        var = np.ascontiguousarray(var, dtype=np.float64)
        check(self.lib, self.lib.grl_set_running_stats(self.h, mean.ctypes.data, var.ctypes.data, float(count)))

    def set_ret_var(self, ret_var):
        check(self.lib, self.lib.grl_set_ret_var(self.h, float(ret_var)))

    def get_obs_stats(self, shape):
        """(mean, var, count) of the device statistics, float64, in the observation's shape (synchronises)."""
        mean, var = np.empty(shape, np.float64), np.empty(shape, np.float64)
        cnt = C.c_double()
        check(self.lib, self.lib.grl_get_obs_stats(self.h, mean.ctypes.data, var.ctypes.data, C.byref(cnt)))
        return mean, var, cnt.value

    def set_learning_rate(self, lr):
        check(self.lib, self.lib.grl_set_learning_rate(self.h, float(lr)))

    def replay_add(self, obs, act, rew, next_obs, done):
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (obs, act, rew, next_obs, done)]
        n = arrs[2].reshape(-1).shape[0]
        check(self.lib, self.lib.grl_replay_add(self.h, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                                                arrs[3].ctypes.data, arrs[4].ctypes.data, n))

    def replay_add_device(self, obs, act, rew, next_obs, done):
        """Same with float32 device tensors (already visible to the engine stream)."""
        n = int(rew.numel())
        p = self.be.ptr
        check(self.lib, self.lib.grl_replay_add_device(self.h, p(obs), p(act), p(rew), p(next_obs), p(done), n))

    def replay_size(self):
        return int(self.lib.grl_replay_size(self.h))

    # ------------------------------------------------------------------ update
    def _noise(self, idx, eps, n_steps):
        if idx is None and eps is None:
            return None, None, (None, None)
        idx = np.ascontiguousarray(idx, dtype=np.int64).reshape(n_steps, self.B)
        eps = np.ascontiguousarray(eps, dtype=np.float32).reshape(n_steps, self.B, self.A)
        if idx.min() < 0 or idx.max() >= self.replay_size():
            raise GrlError("replay index out of range")
        di, de = self.be.to_device(idx), self.be.to_device(eps)
        return C.c_void_p(self.be.ptr(di)), C.c_void_p(self.be.ptr(de)), (di, de)

    def train(self, n_steps=1, idx=None, eps=None):
        """n_steps SAC updates; idx/eps (host arrays) make the minibatch and policy noise explicit."""
        pi, pe, keep = self._noise(idx, eps, n_steps)
        check(self.lib, self.lib.grl_train_step(self.h, n_steps, pi, pe))
        self._keep = [keep]   # keep the staged tensors alive until the stream has consumed them

    def train_device(self, n_steps, idx_dev=None, eps_dev=None):
        """Same with idx/eps already resident on the device (or None for the device RNG)."""
        pi = C.c_void_p(self.be.ptr(idx_dev)) if idx_dev is not None else None
        pe = C.c_void_p(self.be.ptr(eps_dev)) if eps_dev is not None else None
        check(self.lib, self.lib.grl_train_step(self.h, n_steps, pi, pe))

    def compute_grads(self, idx=None, eps=None):
        pi, pe, keep = self._noise(idx, eps, 1)
        check(self.lib, self.lib.grl_compute_grads(self.h, pi, pe))
        self._keep = [keep]

    def compute_grads_staged(self, stage, idx=None, eps=None):
        """Stage 0 / 1 of the gradient computation (grl_compute_grads_staged): bucket 0 is final after stage 0."""
        if stage == 0:
            pi, pe, keep = self._noise(idx, eps, 1)
            check(self.lib, self.lib.grl_compute_grads_staged(self.h, 0, pi, pe))
            self._keep = [keep]
        else:
            check(self.lib, self.lib.grl_compute_grads_staged(self.h, 1, None, None))

    def grad_ranges(self, bucket):
        """[(offset, numel)] of gradient bucket 0 (final after stage 0) / 1 (final after stage 1)."""
        offs, nums = (C.c_int64 * 8)(), (C.c_int64 * 8)()
        n = check(self.lib, self.lib.grl_grad_ranges(self.h, bucket, 8, offs, nums))
        return [(int(offs[k]), int(nums[k])) for k in range(n)]

    def apply_grads(self, grad_scale=1.0):
        check(self.lib, self.lib.grl_apply_grads(self.h, float(grad_scale)))

    def metrics(self):
        m = _capi.GrlMetrics()
        check(self.lib, self.lib.grl_get_metrics(self.h, C.byref(m)))
        return {n: getattr(m, n) for n, _ in m._fields_}

    def synchronize(self):
        self.be.synchronize()

    # ------------------------------------------------------------------ inference
    def act(self, obs, deterministic=True, eps=None, raw=False, observed=False):
        """raw=True: `obs` are un-normalised observations, VecNormalize is applied on the device (norm_update statistics).
        observed=True: act on the observations the last `observe` uploaded (`obs` is then only looked at for its length)."""
        if observed:
            n, po = int(obs if np.isscalar(obs) else len(obs)), None
        else:
            obs = np.ascontiguousarray(obs, dtype=np.float32)
            n, po = obs.shape[0], obs.ctypes.data
        out = np.empty((n, self.A), np.float32)
        pe = None
        if not deterministic:
            eps = np.ascontiguousarray(eps, dtype=np.float32).reshape(n, self.A)
            pe = eps.ctypes.data
        flags = (1 if deterministic else 0) | (2 if raw else 0) | (4 if observed else 0)
        check(self.lib, self.lib.grl_act(self.h, po, n, flags, pe, out.ctypes.data))
        return out

    def load_encoder(self, weights):
        """weights: Keras order conv2d_1..3 kernel/bias, dense_1 kernel/bias (encoders.py:90-108)."""
        arrs = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        ptrs = (C.c_void_p * 8)(*[a.ctypes.data for a in arrs])
        nums = (C.c_int64 * 8)(*[a.size for a in arrs])
        check(self.lib, self.lib.grl_encoder_load(self.h, ptrs, nums, 8))

    def encode(self, depth):
        depth = np.ascontiguousarray(depth, dtype=np.float32).reshape(-1, 64, 64, 1)
        n = depth.shape[0]
        out = np.empty((n, 100), np.float32)
        check(self.lib, self.lib.grl_encode(self.h, depth.ctypes.data, n, out.ctypes.data))
        return out

    # ------------------------------------------------------------------ debugging / profiling
    def fetch(self, name, shape=None):
        cap = 1 << 24
        buf = np.empty(cap, np.float32) if shape is None else np.empty(int(np.prod(shape)), np.float32)
        n = check(self.lib, self.lib.grl_debug_fetch(self.h, name.encode(), buf.ctypes.data, buf.size))
        out = buf[:n].copy()
        return out.reshape(shape) if shape is not None else out

    def store(self, name, arr):
        """Overwrite a named internal tensor (parity tests only; grl_debug_store)."""
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        check(self.lib, self.lib.grl_debug_store(self.h, name.encode(), arr.ctypes.data, arr.size))

    def profile(self, on):
        check(self.lib, self.lib.grl_profile_enable(self.h, 1 if on else 0))

    def profile_dump(self):
        buf = C.create_string_buffer(1 << 14)
        check(self.lib, self.lib.grl_profile_dump(self.h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            tag, ms, n, fl, by, fx = line.split(":")
            out[tag] = {"avg_ms": float(ms), "launches": int(n), "flops": float(fl), "bytes": float(by),
                        "flops_executed": float(fx)}
        return out


class QEngine(SacEngine):
    """DQN / BDQ handle (``_capi.make_q_config``): same arenas and calls; the explicit-noise slot of
    ``train`` / ``compute_grads`` carries prioritised-replay importance weights [n_steps, B] and
    ``act`` returns dueling Q-values [n, branches, bins]."""

    def __init__(self, cfg, backend=None, lib_path=None, device="cuda:0"):
        super().__init__(cfg, backend=backend, lib_path=lib_path, device=device)
        self.D, self.bins = cfg.q_branches, cfg.q_bins

    def _noise(self, idx, weights, n_steps):
        if idx is None and weights is None:
            return None, None, (None, None)
        idx = np.ascontiguousarray(idx, dtype=np.int64).reshape(n_steps, self.B)
        if weights is None:
            weights = np.ones((n_steps, self.B), np.float32)
        weights = np.ascontiguousarray(weights, dtype=np.float32).reshape(n_steps, self.B)
        if idx.min() < 0 or idx.max() >= self.replay_size():
            raise GrlError("replay index out of range")
        di, dw = self.be.to_device(idx), self.be.to_device(weights)
        return C.c_void_p(self.be.ptr(di)), C.c_void_p(self.be.ptr(dw)), (di, dw)

    def q_values(self, obs):
        obs = np.ascontiguousarray(obs, dtype=np.float32)
        n = obs.shape[0]
        out = np.empty((n, self.D * self.bins), np.float32)
        check(self.lib, self.lib.grl_act(self.h, obs.ctypes.data, n, 1, None, out.ctypes.data))
        return out.reshape(n, self.D, self.bins)

    def act(self, obs, deterministic=True, eps=None):
        return self.q_values(obs).argmax(axis=2)

    def update_target(self):
        check(self.lib, self.lib.grl_q_update_target(self.h))

    # ---- prioritised replay on the device (cfg.q_per; csrc/per_kernels.h)
    def train_per(self, n_steps=1, beta=1.0, u=None):
        """n_steps updates on minibatches drawn proportionally to the stored priority**alpha leaves, with importance
        weights (N p)^-beta / max; the leaves of the trained transitions are refreshed on the device afterwards
        (stable-baselines PrioritizedReplayBuffer.sample / update_priorities).
        u: optional host array [n_steps, B] of float64 uniforms in [0, 1) (parity tests); None = device RNG."""
        pu, keep = None, None
        if u is not None:
            u = np.ascontiguousarray(u, dtype=np.float64).reshape(n_steps, self.B)
            keep = self.be.to_device(u)
            pu = C.c_void_p(self.be.ptr(keep))
        check(self.lib, self.lib.grl_train_step_per(self.h, n_steps, float(beta), pu))
        self._keep = [keep]

    def sampled_indices(self):
        return self.fetch("idx_raw", (2 * self.B,)).view(np.int64).copy()

    def importance_weights(self):
        return self.fetch("weights", (self.B,))

    def stored_priorities(self):
        """float64 leaves priority**alpha of every slot of the ring (zeros where nothing is stored)"""
        return self.fetch("per_p", (2 * int(self.cfg.replay_capacity),)).view(np.float64).copy()

    def store_priorities(self, leaves):
        """Overwrite the float64 leaves (parity tests only)."""
        self.store("per_p", np.ascontiguousarray(leaves, np.float64).view(np.float32))

    def td_errors(self):
        return self.fetch("td", (self.B, self.D))

    def priorities(self):
        return self.fetch("priority", (self.B,))
