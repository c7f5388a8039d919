#!/usr/bin/env python
"""Development aid: N processes on ONE GPU drive the in-graph exchange the way bench.py --dp auto does (set-up two-shot,
switch variant, 3 updates on the device RNG at B = 256), with a short time-out and the flag words printed (GRL_PLAN_DUMP).
    python scripts/dp_debug.py <world> <variant: oneshot|twoshot|overlap> [batch]"""
import os, sys, socket, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GRL_TUNE", "dp_timeout_ms=8000")
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, variant, batch):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from grasp_rl import _capi
    from grasp_rl.engine import SacEngine
    from grasp_rl.init import init_parameters
    from grasp_rl.parallel import DataParallelInGraph
    dev = torch.device("cuda", 0)
    cfg = _capi.make_config("augmented", obs_channels=2, n_direct=1, act_dim=5, layers=(64, 64), batch_size=batch,
                            replay_capacity=20000, normalize=True, act_batch=16, seed=1234 + rank)
    eng = SacEngine(cfg, device=str(dev))
    eng.set_parameters(init_parameters(eng.table, seed=0))
    st = bench.fill_replay_on_device(eng, 20000, 100 + rank, dev, "depth", 5)
    eng.set_obs_stats(st["mean"], st["var"], st["ret_var"])
    dp = DataParallelInGraph(eng, mode="twoshot")
    mode, overlap = ("twoshot", True) if variant == "overlap" else (variant, False)
    dp.set_mode(mode, overlap)
    for n in (3, 16, 48):
        t0 = time.perf_counter()
        try:
            dp.train(n)
            print("rank %d: %s train(%d) ok, %d exchanges, %.3f ms/update" % (rank, variant, n, dp.check(), 1e3 * (time.perf_counter() - t0) / n), flush=True)
        except Exception as e:   # noqa: BLE001
            print("rank %d: %s train(%d) FAILED after %.1f s: %s" % (rank, variant, n, time.perf_counter() - t0, e), flush=True)
            os.environ["GRL_PLAN_DUMP"] = "1"
            try:
                dp.check()
            except Exception:   # noqa: BLE001
                pass
            break
    dp.close()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    world, variant = int(sys.argv[1]), sys.argv[2]
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(world, port, variant, batch), nprocs=world, join=True)
