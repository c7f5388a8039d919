"""Data-parallel SAC update: one process per GPU, per-GPU minibatch fixed, replay sharded by rank
(each rank samples its own shard), parameters / Adam state / target net replicated.

The only exchange of the path is one all-reduce(sum) of the flat fp32 gradient bucket per update
(1 342 990 floats = 5.4 MB for depth SAC) issued on the engine's HIP stream through
``torch.distributed`` (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests), followed by the
Adam/Polyak kernel with grad_scale = 1/world.  All three SAC losses are batch means, so the mean of
per-shard gradients equals the gradient of the global batch.
"""
import numpy as np
import torch
import torch.distributed as dist


def allreduce_mean_scale(world):
    return 1.0 / float(world)


def allreduce_flat_(flat, group=None):
    """Sum-all-reduce of a flat gradient tensor, in place (one bucket, one collective)."""
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


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
    def __init__(self, engine, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.eng = engine
        self.group = group
        self.world = dist.get_world_size(group)
        self.scale = allreduce_mean_scale(self.world)
        self.bucket = engine.be.as_torch(engine.grad_tensor())

    def broadcast_parameters(self, src=0):
        """Make every replica start from rank `src`'s parameters (state arena prefix)."""
        p = self.eng.be.as_torch(self.eng.state[: self.eng.sizes.n_params])
        with self.eng.be.stream_context():
            dist.broadcast(p, src=src, group=self.group)

    def train(self, n_steps=1, idx=None, eps=None):
        for s in range(n_steps):
            if idx is None:
                self.eng.compute_grads()
            else:
                self.eng.compute_grads(idx[s:s + 1], eps[s:s + 1])
            with self.eng.be.stream_context():
                allreduce_flat_(self.bucket, self.group)
            self.eng.apply_grads(self.scale)
