#!/usr/bin/env python
"""Headline benchmark: SAC gradient steps / second on 64x64 depth observations, batch 256 per GPU
(BASELINE.json metric; config 2 = config/gripper_grasp.yaml --algo SAC, depth 64x64x2, A=5,
layers [64,64], VecNormalize on).

One step = draw a minibatch from the HBM-resident replay (device Philox) + normalise + 3 CNN
forwards + heads + losses + backward through both trainable CNNs + 3 Adam applies + Polyak update
(14 kernel launches replayed as one hipGraph).
Inputs are resident in HBM before the timed region.  N > 1: one process per GPU, data parallel,
per-GPU batch fixed at 256 (weak scaling), one RCCL all-reduce of the flat fp32 gradient bucket
per step; `value` counts batch-256 gradient computations per second over the whole job
(= N x global steps/s).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32 MFMA / vector peak
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E spec
BATCH = 256
REPLAY = 100_000            # SURVEY.md 8d bench replay size (3.3 GB resident)
ACT_DIM = 5


def fill_replay_on_device(eng, n, seed, device):
    """Synthetic transitions with the reference's per-pixel statistics, generated on the GPU."""
    from grasp_rl import synthetic
    st = synthetic.load_obs_stats("depth")
    mean = torch.from_numpy(st["mean"].astype(np.float32)).to(device)
    std = torch.from_numpy(np.sqrt(st["var"]).astype(np.float32)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    chunk = 2048
    s = eng.be.stream
    for k0 in range(0, n, chunk):
        m = min(chunk, n - k0)
        with torch.cuda.stream(s):
            def draw():
                o = mean + std * torch.randn((m,) + tuple(mean.shape), generator=g, device=device)
                o[..., 0].clamp_(0.02, 2.0)
                o[..., 1] = 0.0
                o[:, 0, 0, 1] = torch.rand(m, generator=g, device=device)
                return o.contiguous()
            obs, nxt = draw(), draw()
            act = (torch.rand((m, ACT_DIM), generator=g, device=device) * 2 - 1).contiguous()
            u = torch.rand(m, generator=g, device=device)
            rew = torch.where(u < 0.8, torch.full_like(u, -200.0),
                              torch.where(u < 0.99, -100.0 + 1000.0 * (torch.rand(m, generator=g, device=device) * 0.06 - 0.03),
                                          torch.full_like(u, 10000.0))).contiguous()
            done = (torch.rand(m, generator=g, device=device) < 1.0 / 15.0).float().contiguous()
            eng.replay_add_device(obs, act, rew, nxt, done)
        s.synchronize()
    return st


def cpu_baseline(seconds=15.0):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores."""
    from oracle import sac as osac
    from grasp_rl import synthetic
    spec = osac.SacSpec()
    orc = osac.SacOracle(spec, seed=0)
    st = synthetic.load_obs_stats("depth")
    tr = synthetic.make_transitions(1024, "depth", ACT_DIM, 0, st)
    idx, eps = synthetic.make_noise(400, BATCH, ACT_DIM, 1024, 1)
    cores = min(os.cpu_count() or 1, 16)   # the small convs stop scaling (and oversubscribe) beyond this
    torch.set_num_threads(cores)
    stats = {"mean": st["mean"], "var": st["var"], "ret_var": st["ret_var"]}

    def one(s):
        raw = {k: tr[k][idx[s]] for k in ("obs", "act", "rew", "next_obs", "done")}
        orc.step(osac.prepare_batch(spec, raw, stats), eps[s])
    for s in range(3):
        one(s)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds and n < 390:
        one(3 + n)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "grad-steps/s", "cores": cores, "kind": "port",
            "sample": "%d oracle SAC updates at batch %d (sampling + float64 normalisation + PyTorch-CPU fp32 step), %.1f s"
                      % (n, BATCH, dt)}


def pmc_traffic(tag):
    """HBM bytes per launch of launch `tag` from the newest committed PMC summary (separate rocprofv3 --pmc
    passes of this same command, scripts/pmc_traffic.sh: FETCH_SIZE with the gfx950 correction + WRITE_SIZE);
    counters cannot be collected from inside the timed process, so this is the recorded measurement or None."""
    import csv
    import glob
    import re

    def version(path):          # ..._v11.csv after ..._v9.csv
        m = re.search(r"_v(\d+)", os.path.basename(path))
        return (int(m.group(1)) if m else -1, path)
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*pmc_hbm_traffic*.csv")),
                   key=version)
    for f in reversed(files):
        try:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if row.get("launch_tag") == tag:
                        mb = [v for k, v in row.items() if k and k.startswith("HBM_MB_per_launch")][0]
                        return round(float(mb) * 1048576.0), "profiles/" + os.path.basename(f)
        except (OSError, ValueError, IndexError):
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--replay", type=int, default=REPLAY)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from grasp_rl import _capi
    from grasp_rl.engine import SacEngine
    from grasp_rl.init import init_parameters
    from grasp_rl.parallel import DataParallelSac

    cfg = _capi.make_config("augmented", obs_channels=2, n_direct=1, act_dim=ACT_DIM, layers=(64, 64),
                            batch_size=BATCH, replay_capacity=args.replay, normalize=True, act_batch=16,
                            seed=1234 + rank)
    eng = SacEngine(cfg, device=str(device))
    eng.set_parameters(init_parameters(eng.table, seed=0))       # identical on every rank
    st = fill_replay_on_device(eng, args.replay, 100 + rank, device)
    eng.set_obs_stats(st["mean"], st["var"], st["ret_var"])
    dp = DataParallelSac(eng) if world > 1 else None

    def run(n):
        if dp is None:
            eng.train_device(n)
        else:
            dp.train(n)

    def barrier():
        eng.synchronize()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    eng.synchronize()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.barrier()
    metrics = eng.metrics()

    roof = None
    if rank == 0 and not args.no_profile:
        # per-kernel durations: separate eager pass of the same workload bracketed by hipEvents on the
        # engine's stream (graph replay cannot carry per-kernel events); not part of the timed region
        eng.profile(True)
        eng.train_device(min(args.steps, 50))
        eng.synchronize()
        prof = eng.profile_dump()
        eng.profile(False)
        total = {k: v["avg_ms"] * v["launches"] for k, v in prof.items()}
        dom = max((k for k in prof if prof[k]["flops"] > 0), key=lambda k: total[k])
        d = prof[dom]
        ach = d["flops"] / (d["avg_ms"] * 1e-3) / 1e12
        roof = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_F32_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_TFLOPS, 4), "traffic": None,
                "avg_launch_ms": round(d["avg_ms"], 5), "flops_per_launch": d["flops"],
                "measured": "separate eager pass, hipEvents on the engine stream, same workload",
                "step_kernel_ms": {k: round(total[k] / max(1, min(args.steps, 50)), 5) for k in sorted(total)}}
        tr = pmc_traffic(dom)
        if tr:
            roof["traffic"], roof["traffic_unit"], roof["traffic_source"] = tr[0], "bytes/launch", tr[1]
        # whole update: algorithmic FLOPs of every GEMM-shaped launch of one step over the graph-replay step time
        step_flops = sum(v["flops"] * v["launches"] for v in prof.values()) / max(1, min(args.steps, 50))
        roof["step_flops"] = step_flops
        roof["step_achieved"] = round(step_flops / (dt / args.steps) / 1e12, 3)
        roof["step_frac"] = round(roof["step_achieved"] / PEAK_F32_TFLOPS, 4)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        value = world * args.steps / dt
        out = {"metric": "SAC grad-steps/sec (64x64 depth, batch 256 per GPU)", "value": round(value, 2),
               "unit": "grad-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "configs[1]: gripper_grasp.yaml --algo SAC, depth 64x64x2, batch 256/GPU, "
                                      "A=5, layers [64,64], VecNormalize, %d-transition replay in HBM, device RNG" % args.replay,
                          "global_batch": BATCH * world, "parallelism": "dp%d" % world,
                          "global_steps_per_s": round(args.steps / dt, 2)},
               "losses": {k: round(float(v), 6) for k, v in metrics.items()},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
