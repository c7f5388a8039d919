"""Data-parallel SAC update: one process per GPU, per-GPU minibatch fixed, replay sharded by rank
(each rank samples its own shard), parameters / Adam state / target net replicated.

The only exchange of the path is the all-reduce(sum) of the flat fp32 gradient bucket per update
(1 342 990 floats = 5.4 MB for depth SAC) through ``torch.distributed`` (backend "nccl" = RCCL
over xGMI; "gloo" in the CPU tests), followed by the Adam/Polyak kernel with grad_scale = 1/world.  All three
SAC losses are batch means, so the mean of per-shard gradients equals the gradient of the global batch.

Two schedules:
  * one bucket: compute_grads -> all_reduce(whole bucket) on the engine stream -> apply_grads;
  * two buckets, overlapped (opt-in: ``overlap=True``): stage 0 of the gradient
    computation ends with the fully-connected + head gradients (90 % of the bytes) final; their all-reduce is
    issued on a second stream and travels over xGMI while stage 1 (convolution backward + convolution weight
    gradients, ~40 % of the update) runs on the engine stream; the small convolution bucket follows, then
    apply_grads.  xGMI is point-to-point: a ring all-reduce of 4.8 MB over 8 GPUs is bound by per-link
    bandwidth and latency, comparable to the update itself -- hiding it is worth the two extra launches of the
    staged plan (grl_compute_grads_staged, include/grl.h);
  * ``DataParallelInGraph`` (below): the exchange as kernels of the update's own graph over IPC-mapped buffers.
"""
import numpy as np
import torch
import torch.distributed as dist


def allreduce_mean_scale(world):
    return 1.0 / float(world)


def allreduce_flat_(flat, group=None):
    """Sum-all-reduce of a flat gradient tensor, in place (one bucket, one collective)."""
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        # validation only (ranks sharing one GPU, where RCCL cannot run): gloo moves host buffers
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def allreduce_ranges_(views, group=None):
    """Sum-all-reduce of the contiguous ranges of one bucket.  With RCCL the ranges go out as ONE grouped call
    (ncclGroupStart / End through torch's coalescing manager): a collective call costs tens of microseconds of host
    time, comparable to the whole update."""
    if len(views) > 1 and views[0].is_cuda and hasattr(dist, "_coalescing_manager") and dist.get_backend(group) == "nccl":
        try:        # private API: only the fast path (no `device=`: that also runs the legacy start / end wrapping)
            with dist._coalescing_manager(group=group, async_ops=False):
                for v in views:
                    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
            return
        except TypeError:       # a torch whose signature differs: one collective per range
            pass
    for v in views:
        allreduce_flat_(v, group)


def gather_moments(mean, var, count, group=None):
    """All-gather of one batch's (mean, var, count); returns them in rank order (float64 throughout)."""
    mean, var = np.asarray(mean, np.float64), np.asarray(var, np.float64)
    flat = np.concatenate([[float(count)], mean.ravel(), var.ravel()])
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"      # RCCL moves device buffers only
    mine = torch.from_numpy(flat).to(dev)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, mine, group=group)
    n = mean.size
    out = []
    for p in parts:
        a = p.cpu().numpy()
        out.append((a[1:1 + n].reshape(mean.shape), a[1 + n:].reshape(var.shape), a[0]))
    return out


def share_running_stats(vec_normalize, group=None):
    """Keep the VecNormalize statistics of all replicas identical: every ``update`` of
    obs_rms / ret_rms merges the batch moments of all ranks, in rank order, into each replica."""
    for rms in (vec_normalize.obs_rms, vec_normalize.ret_rms):
        rms.gather = lambda m, v, c, _g=group: gather_moments(m, v, c, _g)
    return vec_normalize


class DataParallelSac:
    def __init__(self, engine, group=None, overlap=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.eng = engine
        self.group = group
        self.world = dist.get_world_size(group)
        self.scale = allreduce_mean_scale(self.world)
        self.bucket = engine.be.as_torch(engine.grad_tensor())
        full = engine.be.as_torch(engine.grads)
        ranges = [engine.grad_ranges(b) for b in (0, 1)]
        self.staged = len(ranges[1]) > 0
        # (opt-in: the staged plan costs +16 % on-GPU and has no multi-GPU measurement yet)
        self.overlap = bool(overlap) and self.staged
        self.views = [[full[o:o + n] for o, n in r] for r in ranges]     # [bucket][range] views of the grads arena

    def broadcast_parameters(self, src=0):
        """Make every replica start from rank `src`'s parameters (state arena prefix)."""
        p = self.eng.be.as_torch(self.eng.state[: self.eng.sizes.n_params])
        with self.eng.be.stream_context():
            dist.broadcast(p, src=src, group=self.group)

    def _step_single(self, idx, eps):
        self.eng.compute_grads(idx, eps)
        with self.eng.be.stream_context():
            allreduce_flat_(self.bucket, self.group)
        self.eng.apply_grads(self.scale)

    def _step_overlapped(self, idx, eps):
        eng, be = self.eng, self.eng.be
        eng.compute_grads_staged(0, idx, eps)
        fork = be.comm_fork()                        # the second stream will start where stage 0 ends ...
        eng.compute_grads_staged(1)                  # ... but stage 1 is enqueued FIRST: the host time of the collective
        with be.comm_context(fork):                  #     calls below must not delay it (measured: it did, by 60 us)
            allreduce_ranges_(self.views[0], self.group)     # dense bucket: travels while stage 1 runs
        with be.stream_context():
            allreduce_ranges_(self.views[1], self.group)     # convolution bucket, after stage 1
        be.comm_join()
        eng.apply_grads(self.scale)

    def train(self, n_steps=1, idx=None, eps=None):
        step = self._step_overlapped if self.overlap else self._step_single
        for s in range(n_steps):
            if idx is None:
                step(None, None)
            else:
                step(idx[s:s + 1], eps[s:s + 1])


class ExchangeSetupError(RuntimeError):
    """Raised by every rank of the group when any of them could not set up the in-graph exchange."""


def vote_all(flag, group=None):
    """True on every rank iff `flag` is true on all of them (one MIN all-reduce; device tensor under RCCL)."""
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())


class DataParallelInGraph:
    """The exchange inside the library (include/grl.h: grl_allreduce_init / connect / grl_train_step_allreduce,
    csrc/dp_kernels.h): hand-written all-reduces over IPC-mapped exchange buffers, captured in the update's hipGraph --
    one C call per update, no collective library, no Python between compute and apply.  ``torch.distributed`` (any
    backend; gloo is enough) is used ONCE, to hand the 128-byte handle blobs around.  Needs
    HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment (dmabuf IPC).  Every replica receives bit-identical sums: every
    element is added in rank order.  ``mode``: 'auto' (one-shot for world <= 2, else two-shot), 'oneshot', 'twoshot';
    ``overlap``: the staged plan with the dense bucket's exchange under the convolution backward (two-shot).
    A rank waits for a late peer (GRL_TUNE dp_timeout_ms, default 120 s); after a time-out `train` / `check` raise on
    EVERY rank (the channel is poisoned: no replica applies a partial exchange silently)."""

    def __init__(self, engine, group=None, overlap=False, mode="auto"):
        """COLLECTIVE, and fail-safe as a collective: set-up is three local phases (allocate + export, map the peers,
        configure), each followed by a vote over the group, so every rank executes the same sequence of collectives
        whatever failed where.  If any rank fails a phase, EVERY rank releases what it had set up
        (grl_allreduce_disconnect: the handle is a plain single-process handle again, `grl_norm_update` no longer waits
        for peers) and raises ``ExchangeSetupError`` from the same place -- callers can fall back together."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.eng, self.group = engine, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.overlap, self.mode = bool(overlap), mode
        err = [None]

        def phase(fn):
            try:
                out = fn()
            except Exception as e:       # noqa: BLE001  (hipIpc unavailable, a peer's handle that does not map, no staged plan ...)
                err[0], out = e, None
            if vote_all(err[0] is None, group):
                return out
            try:
                engine.allreduce_disconnect()
            except Exception:            # noqa: BLE001
                pass
            raise ExchangeSetupError("in-graph exchange set-up failed on %s" % (
                "this rank (%d): %s" % (self.rank, err[0]) if err[0] is not None else "a peer rank")) from err[0]

        mine = phase(lambda: engine.allreduce_init(self.rank, self.world))
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)

        def connect():
            engine.allreduce_connect(handles)
            engine.allreduce_set_mode(mode)
            if self.overlap:               # dense bucket exchanged on a side lane of the graph, under the conv backward
                engine.allreduce_set_overlap(True)
        phase(connect)
        dist.barrier(group=group)          # every rank has mapped every buffer before the first exchange starts

    def set_mode(self, mode=None, overlap=None):
        """Switch the exchange variant.  Collective: every rank drains its stream and meets at a barrier first, so no
        exchange is in flight anywhere while the source buffers change roles."""
        self.eng.synchronize()
        dist.barrier(group=self.group)
        err = None
        try:
            if overlap is not None:
                self.eng.allreduce_set_overlap(bool(overlap))
                self.overlap = bool(overlap)
            if mode is not None:
                self.eng.allreduce_set_mode(mode)
                self.mode = mode
        except Exception as e:       # noqa: BLE001  (raised after the closing barrier: the ranks stay in step)
            err = e
        dist.barrier(group=self.group)
        if err is not None:
            raise err

    def broadcast_parameters(self, src=0):
        P = [self.eng.get_parameters() if self.rank == src else None]
        dist.broadcast_object_list(P, src=src, group=self.group)
        self.eng.set_parameters(P[0])

    def train(self, n_steps=1, idx=None, eps=None):
        self.eng.train_allreduce(n_steps, idx, eps)    # (raises once a previous exchange has timed out, on every rank)

    def train_per(self, n_steps=1, beta=1.0, u=None):
        """DQN / BDQ with prioritised replay: every rank draws from its own priority tree, the exchange is the same."""
        self.eng.train_per(n_steps, beta, u)

    def check(self):
        """Synchronises; raises if an exchange timed out.  Returns the number of exchanges begun."""
        return self.eng.allreduce_status()

    def close(self, disconnect=False):
        """Collective: every rank drains its stream and meets the others BEFORE any of them releases its engine -- a peer's
        kernels read this rank's exchange memory until their last exchange has completed.  ``disconnect``: also unmap the
        peers and free the exchange memory (after a second barrier: nobody unmaps what a peer is still closing)."""
        self.eng.synchronize()
        dist.barrier(group=self.group)
        if disconnect:
            self.eng.allreduce_disconnect()
            dist.barrier(group=self.group)


def loopback_gloo():
    """One node, rendezvous on the loopback address: tell gloo to use the loopback interface outright (GLOO_SOCKET_IFNAME=lo)
    instead of resolving the box's host name first -- on a box without a resolver every process waits for that lookup to time
    out (eight ranks: two minutes before the first collective; measured with the GPU tests' stamps)."""
    import os
    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")


def launched_world():
    """(rank, world, local_rank) of a process started by ``torch.distributed.run`` / torchrun (RANK, WORLD_SIZE,
    LOCAL_RANK in the environment), or None for a plain single-process start."""
    import os
    try:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
    except ValueError:
        return None
    if world <= 1:
        return None
    return rank, world, int(os.environ.get("LOCAL_RANK", str(rank)))


def runtime_for(mode):
    """The ``DataParallelRuntime`` of a model constructed with ``data_parallel=mode``: None for None / False / "off";
    "auto" only under a torch.distributed launch of more than one process; anything else insists."""
    if mode in (None, False, "", "0", "off"):
        return None
    if mode == "auto" and launched_world() is None and not dist.is_initialized():
        return None
    rt = DataParallelRuntime()
    if rt.world == 1 and mode == "auto":
        return None
    return rt


class DataParallelRuntime:
    """What ``grasp_rl.sb.SAC(data_parallel=...)`` needs around the engine: the process group (initialised here when the
    launcher has not: gloo -- the gradient exchange itself runs inside the update graph and needs no collective
    library), a gloo CONTROL group for the few host-side exchanges of the learn loop (handle blobs, return statistics,
    the stop flag), this rank's device and its share of the environments / of the global minibatch."""

    def __init__(self):
        import os
        lw = launched_world()
        if not dist.is_initialized():
            if lw is None:
                raise RuntimeError("data_parallel needs a torch.distributed launch (torchrun: RANK / WORLD_SIZE / MASTER_*)")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
        loopback_gloo()
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=lw[0], world_size=lw[1])
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.local_rank = lw[2] if lw is not None else self.rank
        # host-side traffic never touches a GPU stream
        self.ctrl = None if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
        same = os.environ.get("GRL_DP_SAME_DEVICE", "0") == "1"       # several ranks on one GPU (tests, dress rehearsals)
        self.device = "cuda:%d" % (0 if same else self.local_rank)

    def shard(self, total, what="environments"):
        """This rank's share of `total` items dealt evenly over the ranks."""
        if total % self.world:
            raise ValueError("%d %s do not divide over %d ranks" % (total, what, self.world))
        return total // self.world

    def any(self, flag):
        """True on every rank if `flag` is true on any (one small host all-reduce; also bounds the skew between ranks)."""
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.ctrl)
        return bool(t.item())

    def all(self, flag):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.ctrl)
        return bool(t.item())

    def barrier(self):
        dist.barrier(group=self.ctrl)

    def make_exchange(self, engine, prefer="ingraph", overlap=False, mode="auto"):
        """The gradient exchange for `engine`: in-graph over IPC-mapped memory when every rank can set it up, else
        (or with prefer="collective") compute -> all-reduce -> apply (``DataParallelSac``: RCCL, or gloo where the ranks
        do not own a GPU each).  The decision is COLLECTIVE: ``DataParallelInGraph`` votes after each of its set-up
        phases and raises on every rank or on none (``self.ingraph_error`` keeps the reason)."""
        self.ingraph_error = None
        if prefer == "ingraph":
            try:
                return DataParallelInGraph(engine, group=self.ctrl, overlap=overlap, mode=mode)
            except ExchangeSetupError as e:       # raised on EVERY rank after the same collectives: all of them fall back
                self.ingraph_error = e
        # a collective library: RCCL with one GPU per rank, gloo otherwise (CPU tests, several ranks sharing a GPU)
        if torch.cuda.is_available() and not os_same_device():
            if dist.get_backend() == "nccl":
                return DataParallelSac(engine, overlap=overlap)
            if getattr(self, "_nccl", None) is None:
                self._nccl = dist.new_group(backend="nccl")
            return DataParallelSac(engine, group=self._nccl, overlap=overlap)
        return DataParallelSac(engine, group=self.ctrl, overlap=overlap)


def os_same_device():
    import os
    return os.environ.get("GRL_DP_SAME_DEVICE", "0") == "1"
