#!/usr/bin/env python
"""Cost of the split compute_grads / apply_grads path (what data parallelism runs between all-reduces) against
the fused update, on one GPU -- development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd")):
    sys.path.insert(0, p)
import torch
import bench
from grasp_rl import _capi
from grasp_rl.engine import SacEngine
from grasp_rl.init import init_parameters

dev = torch.device("cuda", 0)
cfg = _capi.make_config("augmented", obs_channels=2, n_direct=1, act_dim=5, layers=(64, 64), batch_size=256,
                        replay_capacity=20000, normalize=True, act_batch=16, seed=1)
eng = SacEngine(cfg, device=str(dev))
eng.set_parameters(init_parameters(eng.table, seed=0))
st = bench.fill_replay_on_device(eng, 20000, 100, dev, "depth", 5)
eng.set_obs_stats(st["mean"], st["var"], st["ret_var"])
bucket = eng.be.as_torch(eng.grad_tensor())
for name, fn in (("fused train_device(n)", lambda n: eng.train_device(n)),
                 ("split compute/apply", lambda n: [(eng.compute_grads(), eng.apply_grads(1.0)) for _ in range(n)]),
                 ("split + bucket touch on the stream", lambda n: [(eng.compute_grads(), bucket.mul_(1.0), eng.apply_grads(1.0)) for _ in range(n)]),
                 ("staged (2 buckets) compute / apply", lambda n: [(eng.compute_grads_staged(0), eng.compute_grads_staged(1), eng.apply_grads(1.0)) for _ in range(n)])):
    with torch.cuda.stream(eng.be.stream):
        fn(50); eng.synchronize()
        t0 = time.perf_counter(); fn(500); eng.synchronize()
        dt = time.perf_counter() - t0
        t1 = time.perf_counter(); fn(500); host = time.perf_counter() - t1; eng.synchronize()
    print("%-40s %.1f us/update (host-side enqueue %.1f us/update)" % (name, 1e6 * dt / 500, 1e6 * host / 500))

# the in-graph exchange with world = 1 (the rank exchanges with itself: every kernel of the N-rank update runs, nothing waits)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
eng.allreduce_connect([eng.allreduce_init(0, 1)])
VARIANTS = (("one-shot", "oneshot", False), ("two-shot", "twoshot", False), ("two-shot, overlapped", "twoshot", True))
for name, mode, ov in VARIANTS:
    eng.synchronize()
    eng.allreduce_set_overlap(ov)
    eng.allreduce_set_mode(mode)
    eng.train_allreduce(50); eng.synchronize()
    t0 = time.perf_counter(); eng.train_allreduce(500); eng.synchronize()
    dt = time.perf_counter() - t0
    print("in-graph exchange, %-22s (world 1) %.1f us/update, %d exchanges, no time-out" % (name, 1e6 * dt / 500, eng.allreduce_status()))

# per-launch times of the exchange (eager pass, hipEvents)
eng.profile(True)
for name, mode, ov in VARIANTS:
    eng.synchronize()
    eng.allreduce_set_overlap(ov)
    eng.allreduce_set_mode(mode)
    eng.train_allreduce(30)
    eng.synchronize()
    d = eng.profile_dump()
    print(name, " ".join("%s %.1f" % (k, 1e3 * v["avg_ms"]) for k, v in sorted(d.items()) if k.startswith(("dp_", "reduce", "wgrad"))))
    eng.profile(True)       # (clears the accumulators)
eng.profile(False)
