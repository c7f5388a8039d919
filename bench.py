#!/usr/bin/env python
"""Benchmarks of the update hot path on MI355X.  Default = the headline: SAC gradient steps / second on 64x64 depth
observations, batch 256 per GPU (BASELINE.json metric; configs[1] = config/gripper_grasp.yaml --algo SAC, depth
64x64x2, A=5, layers [64,64], VecNormalize on).

One step = draw a minibatch from the HBM-resident replay (device Philox) + normalise + 3 CNN forwards + heads +
losses + backward through both trainable CNNs + 3 Adam applies + Polyak update, replayed as one hipGraph
(DESIGN.md section 4).  Inputs are resident in HBM before the timed region.  A timed block is exactly --steps updates
per call between barrier + synchronize, the call repeated back to back until the block holds >= 50 ms of work
(`repeats.calls_per_block`); --repeats blocks, `value` is the MEDIAN block.  The timed call shape is warmed untimed.

N > 1: one process per GPU, data parallel, gradients all-reduced over RCCL per update.  Default per-GPU batch 256
(weak scaling, `value` = N x global steps/s); `--global-batch G` fixes the GLOBAL batch (G/N per GPU, strong
scaling: BASELINE configs[4] is `--global-batch 1024` on 8 GPUs).

Other workloads (`--workload`), each with its own roofline and CPU-oracle baseline:
  sac_rgbd    configs[3]: RGB-D 64x64x5, batch 256, replay with byte colours, + auto-encoder feature latency
  sac_nature  configs[0] on the GPU: default nature_cnn over both channels, A=3, batch 64, no VecNormalize
  sac_depth_128  the shipped table-clearing SAC config with layers [128,128], batch 64
  sac_mlp     configs[0] as shipped (depth_observation False): sacMlp on 100-d auto-encoder features, A=3, batch 64
  bdq_uniform configs[2] as gripper_grasp.yaml ships it (prioritized_replay: False): the same network on uniform replay
  bdq_per     configs[2]: BDQ 5 branches x 33 bins on 101-d observations, batch 64, prioritised replay over 1 M
  ae_train    auto-encoder training step, batch 128 (config/encoder.yaml)

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
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

PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32 MFMA / vector peak
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E spec
ACT_DIM = 5

WORKLOADS = {
    "sac_depth": dict(kind="depth", extractor="augmented", batch=256, act_dim=5, normalize=True, replay=100_000,
                      metric="SAC grad-steps/sec (64x64 depth, batch 256 per GPU)",
                      name="configs[1]: gripper_grasp.yaml --algo SAC, depth 64x64x2, batch %d/GPU, A=5, layers [64,64], "
                           "VecNormalize, %d-transition replay in HBM, device RNG"),
    "sac_rgbd": dict(kind="rgbd", extractor="augmented", batch=256, act_dim=5, normalize=True, replay=50_000, rgb_u8=True,
                     metric="SAC grad-steps/sec (64x64 RGB-D, batch 256 per GPU)",
                     name="configs[3]: SAC on RGB-D 64x64x5 (SAC_full_rgbd/config.yaml), batch %d/GPU, A=5, layers [64,64], "
                          "VecNormalize, %d-transition replay with byte colours in HBM, device RNG"),
    "sac_depth_128": dict(kind="depth", extractor="augmented", batch=64, act_dim=5, normalize=True, replay=100_000, layers=(128, 128),
                          metric="SAC grad-steps/sec (64x64 depth, layers [128,128], batch 64)",
                          name="trained_models/table_clearing/SAC_real_2m_buffer_128/config.yaml: SAC, depth 64x64x2, batch %d/GPU, A=5, "
                               "layers [128,128] (heads_fused_kernel<128>: two column blocks per wave), VecNormalize, "
                               "%d-transition replay in HBM, device RNG"),
    "sac_nature": dict(kind="depth", extractor="nature", batch=64, act_dim=3, normalize=False, replay=50_000,
                       metric="SAC grad-steps/sec (64x64 depth, nature_cnn, batch 64)",
                       name="configs[0]: simplified_object_picking.yaml --algo SAC with depth observations: default nature_cnn "
                            "over both channels, batch %d/GPU, A=3, layers [64,64], no VecNormalize, %d-transition replay"),
    "sac_mlp": dict(kind="vector", extractor="mlp", batch=64, act_dim=3, normalize=False, replay=50_000, obs_dim=100,
                    metric="SAC grad-steps/sec (100-d auto-encoder features, sacMlp, batch 64)",
                    name="configs[0] as shipped: simplified_object_picking.yaml --algo SAC (depth_observation False: sacMlp on the "
                         "100-d auto-encoder features, sensor.py:220-222), batch %d/GPU, A=3, layers [64,64], no VecNormalize, "
                         "%d-transition replay"),
}


# ------------------------------------------------------------------------------------------------ helpers
def fill_replay_vectors(eng, n, seed, device, obs_dim, act_dim):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for k0 in range(0, n, 65536):
        m = min(65536, n - k0)
        with torch.cuda.stream(eng.be.stream):
            obs = (0.5 + 0.3 * torch.randn((m, obs_dim), generator=g, device=device)).contiguous()
            nxt = (0.5 + 0.3 * torch.randn((m, obs_dim), generator=g, device=device)).contiguous()
            act = (torch.rand((m, act_dim), generator=g, device=device) * 2 - 1).contiguous()
            rew = (-150.0 + 300.0 * torch.randn(m, generator=g, device=device)).contiguous()
            done = (torch.rand(m, generator=g, device=device) < 1.0 / 15.0).float().contiguous()
            eng.replay_add_device(obs, act, rew, nxt, done)
        eng.be.stream.synchronize()


def fill_replay_on_device(eng, n, seed, device, kind="depth", act_dim=5):
    """Synthetic transitions with the reference's per-pixel statistics, generated on the GPU."""
    import numpy as np
    import torch
    from grasp_rl import synthetic
    st = synthetic.load_obs_stats(kind)
    mean = torch.from_numpy(st["mean"].astype(np.float32)).to(device)
    std = torch.from_numpy(np.sqrt(st["var"]).astype(np.float32)).to(device)
    C = mean.shape[-1]
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    chunk = 2048 if kind == "depth" else 1024
    s = eng.be.stream
    for k0 in range(0, n, chunk):
        m = min(chunk, n - k0)
        with torch.cuda.stream(s):
            def draw():
                o = mean + std * torch.randn((m,) + tuple(mean.shape), generator=g, device=device)
                if kind == "rgbd":
                    o[..., :3] = o[..., :3].round().clamp_(0, 255)
                o[..., C - 2].clamp_(0.02, 2.0)
                o[..., C - 1] = 0.0
                o[:, 0, 0, C - 1] = torch.rand(m, generator=g, device=device)
                return o.contiguous()
            obs, nxt = draw(), draw()
            act = (torch.rand((m, act_dim), generator=g, device=device) * 2 - 1).contiguous()
            u = torch.rand(m, generator=g, device=device)
            rew = torch.where(u < 0.8, torch.full_like(u, -200.0),
                              torch.where(u < 0.99, -100.0 + 1000.0 * (torch.rand(m, generator=g, device=device) * 0.06 - 0.03),
                                          torch.full_like(u, 10000.0))).contiguous()
            done = (torch.rand(m, generator=g, device=device) < 1.0 / 15.0).float().contiguous()
            eng.replay_add_device(obs, act, rew, nxt, done)
        s.synchronize()
    return st


MIN_BLOCK_S = 0.050         # a timed block holds at least this much work (the driver's --steps 20 is 3.6 ms of updates)


def timed_blocks(run, barrier, steps, warmup, repeats, world, device):
    """`repeats` timed blocks (barrier + synchronize on both sides, MAX over ranks).  A block is `calls` back-to-back
    `run(steps)` calls -- each exactly `steps` updates, the call shape the command line names -- with `calls` chosen so that
    a block holds >= MIN_BLOCK_S of work (1 when `steps` is large enough on its own).  Returns (seconds per `steps` updates
    of every block, calls).  Untimed before the first block: `run(warmup)` AND the exact timed shape, `run(steps)`, twice --
    a `steps`-update call replays hipGraphs of 16 + (steps % 16) updates that a `warmup`-update call never instantiates
    (round 5: the driver's `--steps 20 --warmup 5` blocks fell 13 % across the five repeats)."""
    import torch
    import torch.distributed as dist

    def over_ranks(v, op):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=op)
        return float(t.item())
    if warmup > 0:
        run(warmup)
    run(steps)
    barrier()
    t0 = time.perf_counter()
    run(steps)
    barrier()
    est = over_ranks(time.perf_counter() - t0, dist.ReduceOp.MAX)       # the same `calls` on every rank
    calls = max(1, min(4096, int(-(-MIN_BLOCK_S // max(est, 1e-6)))))
    times = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        for _c in range(calls):
            run(steps)
        barrier()
        times.append(over_ranks(time.perf_counter() - t0, dist.ReduceOp.MAX) / calls)
    return times, calls


def block_note(calls, steps):
    return {"calls_per_block": calls, "updates_per_block": calls * steps,
            "block_ms_is": "milliseconds per %d updates: a block is %d back-to-back call(s) of exactly %d updates between barrier + "
                           "synchronize, its time divided by the calls; the timed call shape was warmed untimed" % (steps, calls, steps)}


def profile_pass(eng, go, n):
    """Per-launch durations: a separate eager pass of the same workload bracketed by hipEvents on the engine's
    stream (graph replay cannot carry per-kernel events); not part of the timed region."""
    eng.profile(True)
    go(n)
    eng.synchronize()
    prof = eng.profile_dump()
    eng.profile(False)
    return prof


def roofline_of(prof, n, dt_step, workload):
    """Roofline block of the bench line.  Every number is reproducible from what the line and `profiles/` carry:

    * `frac` / `achieved` of the dominant launch = its ALGORITHMIC FLOPs (2*M*N*K over taps and rows that exist;
      `flops_per_launch`) / `avg_launch_ms`, the launch's average duration in this process's eager pass, timed with
      hipEvents on the engine stream (`timing`: graph replay cannot carry per-kernel events);
    * `frac_graph`: the same FLOPs over the kernel-trace average of the committed rocprofv3 summary of this command
      under hipGraph replay (`graph_source`; launches run 5-7 % shorter back to back inside the graph) -- a recorded
      measurement, like `traffic`;
    * `step_flops` = algorithmic FLOPs of one update (what the path needs: valid taps only, no backward-data of the
      first convolution -- the observations need no gradient), `step_flops_executed` = what the MFMAs execute
      (equal for the SAC update: its backward-data runs over exact taps; the auto-encoder's masked form executes more); `step_frac` = step_flops / ms_per_step /
      peak, i.e. from the timed graph-replay blocks that `value` comes from;
    * `memory`: the HBM-bound launches of the update: algorithmic bytes / eager duration / 8 TB/s."""
    total = {k: v["avg_ms"] * v["launches"] for k, v in prof.items()}
    with_flops = [k for k in prof if prof[k]["flops"] > 0]
    roof = {"measured": "separate eager pass, hipEvents on the engine stream, same workload",
            "step_kernel_ms": {k: round(total[k] / max(1, n), 5) for k in sorted(total)}}
    if with_flops:
        dom = max(with_flops, key=lambda k: total[k])
        d = prof[dom]
        ach = d["flops"] / (d["avg_ms"] * 1e-3) / 1e12
        roof.update({"kernel": dom, "bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / PEAK_F32_TFLOPS, 4), "timing": "eager pass, hipEvents", "traffic": None,
                     "avg_launch_ms": round(d["avg_ms"], 5), "flops_per_launch": d["flops"],
                     "flops_executed_per_launch": d.get("flops_executed", d["flops"])})
        tr = pmc_traffic(dom, workload)
        if tr:
            roof["traffic"], roof["traffic_unit"], roof["traffic_source"] = tr[0], "bytes/launch", tr[1]
        gr = graph_trace_avg(dom, workload, live_value=1.0 / dt_step if dt_step else None)
        if gr[0] is None and gr[1]:
            roof["graph_source_refused"] = gr[1]
        if gr[0] is not None:
            roof["graph_avg_launch_ms"] = round(gr[0], 5)
            roof["frac_graph"] = round(d["flops"] / (gr[0] * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4)
            roof["graph_source"] = gr[1]
        step_flops = sum(v["flops"] * v["launches"] for v in prof.values()) / max(1, n)
        step_exec = sum(v.get("flops_executed", v["flops"]) * v["launches"] for v in prof.values()) / max(1, n)
        roof["step_flops"] = step_flops
        roof["step_flops_executed"] = step_exec
        roof["step_flops_is"] = ("algorithmic: 2*M*N*K of every product over existing taps / rows; no backward-data of the "
                                 "first convolution")
        roof["step_achieved"] = round(step_flops / dt_step / 1e12, 3)
        roof["step_frac"] = round(roof["step_achieved"] / PEAK_F32_TFLOPS, 4)
        roof["step_timing"] = "ms_per_step of the timed graph-replay blocks"
    else:                        # no GEMM-shaped launch carries FLOP counts: report the longest launch against HBM
        dom = max(prof, key=lambda k: total[k])
        d = prof[dom]
        ach = d["bytes"] / (d["avg_ms"] * 1e-3) / 1e9 if d["bytes"] else 0.0
        roof.update({"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None, "avg_launch_ms": round(d["avg_ms"], 5)})
    mem = []
    for k in sorted(prof):
        d = prof[k]
        if d["bytes"] > 0 and d["flops"] == 0 and d["avg_ms"] > 0:
            gbs = d["bytes"] / (d["avg_ms"] * 1e-3) / 1e9
            mem.append({"kernel": k, "bound": "hbm", "bytes_per_launch": d["bytes"], "avg_launch_ms": round(d["avg_ms"], 5),
                        "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)})
    if mem:
        roof["memory"] = mem
    return roof


def graph_trace_avg(tag, workload="sac_depth", live_value=None):
    """Average duration (ms) of launch `tag` under hipGraph replay, from the newest committed rocprofv3 kernel-trace
    summary of this workload (`profiles/rNN_rocprofv3_summary_<workload>.txt`, written by scripts/profile_round.sh
    from `rocprofv3 --kernel-trace --stats -- python bench.py ...`; lines `launch <tag> calls <n> avg <us> us`).
    A summary goes stale silently when a kernel changes: the file carries the bench line of its own (profiled) run and, since
    round 4, a line `unprofiled_value: <updates/s>` of the same build on the same box without the profiler (launch-bound
    workloads run up to 20 % slower under rocprofv3).  A file whose recorded value -- the unprofiled one when present -- is
    more than 5 % away from THIS run's is refused (returns (None, reason))."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_summary_%s.txt" % workload)))
    for f in reversed(files):
        try:
            text = open(f).read()
        except OSError:
            continue
        rec = re.search(r'unprofiled_value:\s*([0-9.]+)', text) or re.search(r'"value":\s*([0-9.]+)', text)
        if live_value and rec and abs(float(rec.group(1)) - live_value) > 0.05 * live_value:
            return None, "profiles/%s refused: recorded value %.1f is more than 5 %% from this run's %.1f" % (
                os.path.basename(f), float(rec.group(1)), live_value)
        for line in text.splitlines():
            m = re.match(r"launch\s+(\S+)\s+calls\s+(\d+)\s+avg\s+([0-9.]+)\s+us", line)
            if m and m.group(1) == tag:
                return float(m.group(3)) * 1e-3, "profiles/" + os.path.basename(f)
    return None, None


def pmc_traffic(tag, workload="sac_depth"):
    """HBM bytes per launch of launch `tag` from the newest committed PMC summary of this workload (separate
    rocprofv3 --pmc passes of this same command, scripts/pmc_traffic.sh: FETCH_SIZE with the gfx950 correction +
    WRITE_SIZE); counters cannot be collected from inside the timed process, so this is the recorded measurement."""
    import csv
    import glob
    import re

    def version(path):          # r02 ... _v3.csv after r01 ... _v11.csv
        b = os.path.basename(path)
        r = re.search(r"r(\d+)_", b)
        m = re.search(r"_v(\d+)", b)
        return (int(r.group(1)) if r else 0, int(m.group(1)) if m else -1, path)
    pat = "*pmc_hbm_traffic*.csv" if workload == "sac_depth" else "*pmc_hbm_traffic_%s*.csv" % workload
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)), key=version)
    if workload == "sac_depth":
        files = [f for f in files if not re.search(r"traffic_(sac_rgbd|sac_nature|bdq_per|ae_train)", f)]
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


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_baseline_sac(wl, seconds=12.0):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores."""
    import torch
    from oracle import sac as osac
    from grasp_rl import synthetic
    B, A = wl["batch"], wl["act_dim"]
    if wl["extractor"] == "mlp":
        import numpy as np
        D = wl["obs_dim"]
        spec = osac.SacSpec(extractor="mlp", obs_dim=D, act_dim=A, layers=list(wl.get("layers", (64, 64))))
        orc = osac.SacOracle(spec, seed=0)
        tr = synthetic.make_vector_transitions(4096, np.full(D, 0.5), np.full(D, 0.09), A, 0)
        idx, eps = synthetic.make_noise(4000, B, A, 4096, 1)
        cores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(cores)
        for s in range(20):
            orc.step(osac.prepare_batch(spec, {k: tr[k][idx[s]] for k in ("obs", "act", "rew", "next_obs", "done")}, None), eps[s])
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds and n < 3900:
            orc.step(osac.prepare_batch(spec, {k: tr[k][idx[20 + n]] for k in ("obs", "act", "rew", "next_obs", "done")}, None), eps[20 + n])
            n += 1
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "grad-steps/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
                "sample": "%d oracle SAC-MLP updates at batch %d (sampling + PyTorch-CPU fp32 step), %.1f s" % (n, B, dt)}
    st = synthetic.load_obs_stats(wl["kind"])
    C = st["mean"].shape[-1]
    if wl["extractor"] == "augmented":
        spec = osac.SacSpec(extractor="augmented", img_channels=C - 1, n_direct=1, act_dim=A, layers=list(wl.get("layers", (64, 64))))
    else:
        spec = osac.SacSpec(extractor="nature", img_channels=C, n_direct=0, act_dim=A, layers=list(wl.get("layers", (64, 64))))
    orc = osac.SacOracle(spec, seed=0)
    n_tr = 1024 if wl["kind"] == "depth" else 512
    tr = synthetic.make_transitions(n_tr, wl["kind"], A, 0, st)
    idx, eps = synthetic.make_noise(400, B, A, n_tr, 1)
    cores = min(os.cpu_count() or 1, 16)   # threads USED: the small convs stop scaling (and oversubscribe) beyond this;
    # `host_cores` beside it = what the box has
    torch.set_num_threads(cores)
    stats = {"mean": st["mean"], "var": st["var"], "ret_var": st["ret_var"]} if wl["normalize"] else None

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
    return {"value": n / dt, "unit": "grad-steps/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
            "sample": "%d oracle SAC updates at batch %d (sampling + float64 normalisation + PyTorch-CPU fp32 step), %.1f s"
                      % (n, B, dt)}


def cpu_baseline_bdq(seconds=8.0):
    import numpy as np
    import torch
    from oracle import dqn as od
    spec = od.bdq_spec(101, 5, 33, [[64, 64], [32], [32]])
    spec.lr = 1e-4
    orc = od.QOracle(spec, od.init_params(spec, 0))
    rng = np.random.default_rng(0)
    B = 64
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)

    def one():
        f = lambda a: torch.from_numpy(np.asarray(a, np.float32))
        batch = {"obs": f(rng.normal(size=(B, 101))), "next_obs": f(rng.normal(size=(B, 101))),
                 "act": f(rng.integers(0, 33, (B, 5))), "rew": f(rng.normal(size=B)), "done": f(rng.random(B) < 0.07)}
        orc.step(batch, rng.uniform(0.3, 1.0, B).astype(np.float32))
    for _ in range(5):
        one()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        one()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "grad-steps/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
            "sample": "%d oracle BDQ updates at batch 64 (PyTorch-CPU fp32; minibatch handed over, no sum-tree walk), %.1f s" % (n, dt)}


def cpu_baseline_ae(seconds=8.0):
    import numpy as np
    import torch
    from oracle.autoencoder import AeOracle
    from grasp_rl.autoencoder import glorot_uniform_params
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    orc = AeOracle(glorot_uniform_params(0), lr=2e-4)
    x = np.random.default_rng(0).uniform(0, 0.5, (128, 64, 64, 1)).astype(np.float32)
    orc.step(x)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        orc.step(x)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "train-steps/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
            "sample": "%d oracle auto-encoder training steps at batch 128 (PyTorch-CPU fp32), %.1f s" % (n, dt)}


CONFIGS4_GLOBAL_BATCH = 1024  # BASELINE.json configs[4]
VERIFY_WAIT_MS = 5000         # wait bound while a freshly set-up in-graph exchange is verified and timed (grl_allreduce_set_timeout)
OVERLAP_WORTH_MS = 0.040      # the overlapped plan costs ~0.05 ms more than two-shot at world 1 (profiles/r04_dp_overhead.txt)
IN_GRAPH_VARIANTS = {      # --dp -> [(mode, overlap)] tried in this order
    "auto": [("oneshot", False), ("twoshot", False), ("twoshot", True)],
    "ingraph": [("auto", False)], "ingraph-oneshot": [("oneshot", False)], "ingraph-twoshot": [("twoshot", False)],
    "ingraph-overlap": [("twoshot", True)]}


def make_data_parallel(eng, kind, world, rank, device, init, inject_ipc_failure=None):
    """The exchange step for N > 1.  The in-graph all-reduces over IPC-mapped buffers (grasp_rl.parallel.DataParallelInGraph:
    one-shot, two-shot, two-shot with the dense bucket overlapped) are set up, each VERIFIED on three updates (no time-out,
    replicas bit-identical across ranks -- a checksum travels) and, with 'auto', TIMED on 48 updates (max over ranks); the
    fastest verified variant runs.  Otherwise RCCL with one gradient bucket.  Every decision is collective: a rank whose
    local attempt fails still executes the same sequence of collectives as its peers."""
    import torch
    import torch.distributed as dist
    from grasp_rl.parallel import DataParallelInGraph, DataParallelSac

    def agreed(flag):          # every rank takes the same path
        t = torch.tensor([1.0 if flag else 0.0], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t) > 0

    def attempt(fn, what):
        try:
            fn()
            return True
        except Exception as exc:   # noqa: BLE001
            sys.stderr.write("bench[rank %d]: in-graph exchange, %s: %s\n" % (rank, what, exc))
            return False

    chosen = {"oneshot": "ingraph-oneshot", "twoshot": "ingraph-twoshot"}
    fell_back = ""
    if kind in IN_GRAPH_VARIANTS:
        box = {}
        # this rank's update WITHOUT an exchange (max over ranks): what every variant's time is read against -- the exchange
        # costs (variant - plain) per update, and the overlapped plan (two more launches, +27 % at world 1) can only win when
        # that cost exceeds what its side lane hides
        def plain():
            eng.train_device(16)
            eng.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            eng.train_device(48)
            eng.synchronize()
            box["plain"] = time.perf_counter() - t0
        plain_ok = agreed(attempt(plain, "plain update (no exchange)"))
        plain_ms = None
        if plain_ok:
            t = torch.tensor([box["plain"]], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            plain_ms = 1e3 * float(t) / 48
        eng.set_parameters(init)
        eng.reset_optimizer()
        # (DataParallelInGraph votes after each of its set-up phases: it raises on every rank or on none)
        if inject_ipc_failure == rank:          # validation aid (--inject-ipc-failure): this rank's peer mapping "fails"
            def broken(handles, _real=eng.allreduce_connect):
                _real(handles)
                raise RuntimeError("injected: hipIpcOpenMemHandle fails on rank %d (--inject-ipc-failure)" % rank)
            eng.allreduce_connect = broken
        ok = agreed(attempt(lambda: box.setdefault("dp", DataParallelInGraph(eng, mode="twoshot")), "set-up"))
        if inject_ipc_failure == rank:
            del eng.allreduce_connect
        good, ms, skipped = [], {}, []
        if ok:
            dp = box["dp"]
            # a channel that cannot work (flags that never become visible across a link) must cost seconds, not the 120 s a late
            # peer is granted during training: every wait of the verification / timing updates below is bounded by VERIFY_WAIT_MS
            eng.allreduce_set_timeout(VERIFY_WAIT_MS)
            for mode, overlap in IN_GRAPH_VARIANTS[kind]:
                name = mode + ("+overlap" if overlap else "")
                if overlap and kind == "auto" and plain_ms is not None and ms and min(ms.values()) - plain_ms < OVERLAP_WORTH_MS:
                    skipped.append("%s not tried: the cheapest plain exchange costs %.3f ms per update over the %.4f ms update, below the "
                                   "%.3f ms the overlapped plan adds on its own" % (name, min(ms.values()) - plain_ms, plain_ms, OVERLAP_WORTH_MS))
                    continue
                # (switching is itself collective: drain, barrier, switch, barrier; a configuration without a staged plan
                # refuses `overlap` on every rank alike)
                if not agreed(attempt(lambda: dp.set_mode(mode, overlap), name)):
                    continue
                if not agreed(attempt(lambda: (dp.train(3), dp.check()), name + " (3 updates)")):
                    break                          # a time-out poisons the channel on every rank: no further in-graph attempt
                P = eng.get_parameters()
                chk = torch.tensor([float(sum(float(v.astype("float64").sum()) for v in P.values()))], dtype=torch.float64, device=device)
                lo, hi = chk.clone(), chk.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                if float(lo) != float(hi):         # (the same verdict on every rank)
                    sys.stderr.write("bench[rank %d]: replicas differ after the in-graph exchange (%s)\n" % (rank, name))
                    continue
                if len(IN_GRAPH_VARIANTS[kind]) > 1:
                    def timed():
                        dp.train(16)
                        eng.synchronize()
                        dist.barrier()
                        t0 = time.perf_counter()
                        dp.train(48)
                        eng.synchronize()
                        box["t"] = time.perf_counter() - t0
                        dp.check()
                    if not agreed(attempt(timed, name + " (timing)")):
                        break
                    t = torch.tensor([box["t"]], dtype=torch.float64, device=device)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    ms[(mode, overlap)] = 1e3 * float(t) / 48
                good.append((mode, overlap))
        eng.set_parameters(init)                   # the timed run starts from the common initial state
        eng.reset_optimizer()
        if ok:
            eng.allreduce_set_timeout(0)           # back to the training bound (GRL_TUNE dp_timeout_ms, default 120 s)
        if good:
            mode, overlap = min(good, key=lambda v: ms.get(v, 0.0))
            dp.set_mode(mode, overlap)
            note = ""
            if ms:
                note = "; chosen by timing (ms per update, max over ranks): " + ", ".join("%s%s %.4f" % (m, "+overlap" if o else "", ms[(m, o)]) for m, o in good)
            if plain_ms is not None:
                note += "; update without an exchange %.4f ms" % plain_ms
            if skipped:
                note += "; " + "; ".join(skipped)
            resolved = mode if mode != "auto" else ("oneshot" if world <= 2 else "twoshot")
            verified = "; verified on 3 updates under a %d ms wait bound (no time-out, replicas bit-identical): %s" % (
                VERIFY_WAIT_MS, ", ".join(m + ("+overlap" if o else "") for m, o in good))
            return dp, "dp%d, %s all-reduce over IPC-mapped buffers inside the update graph%s%s%s" % (
                world, {"oneshot": "one-shot", "twoshot": "two-shot"}[resolved],
                " (dense bucket on a side lane under the conv backward)" if overlap else "", note, verified), (
                "ingraph-overlap" if overlap else chosen[resolved])
        if kind != "auto":
            raise SystemExit("--dp %s: the in-graph exchange could not be set up / verified on every rank" % kind)
        sys.stderr.write("bench[rank %d]: no in-graph variant verified: falling back to RCCL\n" % rank)
        fell_back = "in-graph exchange %s on some rank -> every rank fell back together; " % ("could not be set up" if not ok else "did not verify")
    dp = DataParallelSac(eng, overlap=(kind == "rccl-overlap"))
    return dp, "dp%d, %s%s all-reduce, %s" % (world, fell_back, "RCCL" if dist.get_backend() == "nccl" else dist.get_backend(),
                                            "two gradient buckets (dense bucket under the conv backward)" if dp.overlap
                                            else "one gradient bucket"), ("rccl-overlap" if dp.overlap else "rccl")


# ------------------------------------------------------------------------------------------------ workloads
def build_sac_engine(wl, replay, rank, device):
    """Engine of one rank for workload `wl`: parameters identical on every rank, this rank's replay shard filled on the device."""
    from grasp_rl import _capi
    from grasp_rl.engine import SacEngine
    from grasp_rl.init import init_parameters
    C = 5 if wl["kind"] == "rgbd" else 2
    if wl["extractor"] == "mlp":
        cfg = _capi.make_config("mlp", obs_dim=wl["obs_dim"], act_dim=wl["act_dim"], layers=wl.get("layers", (64, 64)), batch_size=wl["batch"],
                                replay_capacity=replay, normalize=wl["normalize"], act_batch=16, seed=1234 + rank)
    else:
        cfg = _capi.make_config(wl["extractor"], obs_channels=C, n_direct=1 if wl["extractor"] == "augmented" else 0,
                                act_dim=wl["act_dim"], layers=wl.get("layers", (64, 64)), batch_size=wl["batch"], replay_capacity=replay,
                                normalize=wl["normalize"], act_batch=16, seed=1234 + rank, replay_rgb_u8=wl.get("rgb_u8", False))
    eng = SacEngine(cfg, device=str(device))
    eng.set_parameters(init_parameters(eng.table, seed=0))       # identical on every rank
    if wl["extractor"] == "mlp":
        fill_replay_vectors(eng, replay, 100 + rank, device, wl["obs_dim"], wl["act_dim"])
    else:
        st = fill_replay_on_device(eng, replay, 100 + rank, device, wl["kind"], wl["act_dim"])
        eng.set_obs_stats(st["mean"], st["var"], st["ret_var"])
    return eng


def device_identities(world, rank, device, same_device):
    """What proves that an N-rank record ran on N distinct GPUs: every rank's device (name, PCI address, uuid where the build
    exposes them) and its row of the peer-access matrix (hipDeviceCanAccessPeer towards every other rank's device), gathered
    on all ranks (a collective) and reported by rank 0."""
    import torch
    import torch.distributed as dist
    p = torch.cuda.get_device_properties(device)
    mine = {"rank": rank, "device": str(device), "name": p.name, "host_pid": os.getpid()}
    for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"):
        if hasattr(p, k):
            mine[k] = int(getattr(p, k))
    if all(k in mine for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        mine["pci"] = "%04x:%02x:%02x.0" % (mine["pci_domain_id"], mine["pci_bus_id"], mine["pci_device_id"])
    if hasattr(p, "uuid"):
        mine["uuid"] = str(p.uuid)
    if same_device:
        mine["peer_access"] = "all ranks share cuda:0 (--same-device)"
    else:
        n_dev = torch.cuda.device_count()
        mine["peer_access"] = [None if j == device.index else (bool(torch.cuda.can_device_access_peer(device.index, j)) if j < n_dev else None)
                               for j in range(world)]      # (one rank per GPU of the node: rank j runs on device j)
    rows = [None] * world
    dist.all_gather_object(rows, mine)
    ids = {r.get("uuid") or r.get("pci") or r["device"] for r in rows}
    return {"ranks": rows, "distinct_devices": len(ids)}


def run_sac(args, wl_name, world, rank, device):
    import numpy as np
    import torch
    import torch.distributed as dist
    from grasp_rl import _capi
    from grasp_rl.engine import SacEngine
    from grasp_rl.init import init_parameters
    from grasp_rl.parallel import DataParallelSac
    wl = dict(WORKLOADS[wl_name])
    strong = args.global_batch is not None
    if strong:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be a multiple of the number of GPUs")
        wl["batch"] = args.global_batch // world
    replay = args.replay or wl["replay"]
    eng = build_sac_engine(wl, replay, rank, device)
    dp, dp_kind, dp_chosen, devices = None, "dp%d" % world, None, None
    if world > 1:
        devices = device_identities(world, rank, device, args.same_device)
        dp, dp_kind, dp_chosen = make_data_parallel(eng, args.dp, world, rank, device, init_parameters(eng.table, seed=0),
                                                    inject_ipc_failure=args.inject_ipc_failure)

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

    times, calls = timed_blocks(run, barrier, args.steps, args.warmup, args.repeats, world, device)
    dt = float(np.median(times))
    metrics = eng.metrics()
    out = {"metric": wl["metric"] if not strong else "SAC grad-steps/sec (64x64 depth, global batch %d)" % args.global_batch,
           "value": None, "unit": "grad-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
           "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
    if strong:      # one global step consumes the whole global batch: `value` = global updates / s
        out["value"] = round(args.steps / dt, 2)
    else:           # per-GPU batch fixed: batch-sized gradient computations / s over the whole job
        out["value"] = round(world * args.steps / dt, 2)
    out["config"] = {"workload": wl["name"] % (wl["batch"], replay) + (" [configs[4]: global batch %d over %d GPU(s)]"
                                                                        % (args.global_batch, world) if strong else ""),
                     "global_batch": wl["batch"] * world, "parallelism": dp_kind,
                     "global_steps_per_s": round(args.steps / dt, 2)}
    out["repeats"] = {"n": args.repeats, "block_ms": [round(1e3 * t, 3) for t in times], "value_is": "median block", **block_note(calls, args.steps)}
    if args.same_device and world > 1:
        out["config"]["validation_only"] = "all %d ranks share cuda:0 (--same-device): not a scaling measurement" % world
    out["losses"] = {k: round(float(v), 6) for k, v in metrics.items()}
    if rank == 0 and not args.no_profile:
        n = min(args.steps, 50)
        prof = profile_pass(eng, eng.train_device, n)
        out["roofline"] = roofline_of(prof, n, dt / args.steps, wl_name if not strong else wl_name + "_gb%d" % args.global_batch)
    else:
        out["roofline"] = None
    if wl_name == "sac_rgbd" and rank == 0 and world == 1:
        # "+ pretrained autoencoder features" (configs[3]): latency of the encoder forward the env side calls per step
        from grasp_rl.autoencoder import glorot_uniform_params
        P = glorot_uniform_params(0)
        eng.load_encoder([P[k] for k in list(P)[:8]])
        frames = np.random.default_rng(0).uniform(0, 0.5, (16, 64, 64, 1)).astype(np.float32)
        eng.encode(frames)
        t0 = time.perf_counter()
        for _ in range(200):
            eng.encode(frames)
        out["ae_encode_us_per_call_16_frames"] = round(1e6 * (time.perf_counter() - t0) / 200, 1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_sac(wl)
    out["cpu_baseline"] = cpu
    if devices is not None:
        out["config"]["devices"] = devices
        out["config"]["exchange"] = dp_chosen
    if dp is not None and hasattr(dp, "close"):
        dp.close(disconnect=True)      # collective: drained everywhere before any rank unmaps its peers / frees its exchange memory
    eng.close()
    if world > 1 and not strong and wl_name == "sac_depth" and CONFIGS4_GLOBAL_BATCH % world == 0 and not args.no_configs4:
        # BASELINE configs[4] is a GLOBAL batch of 1024 (128 per GPU at 8): the driver's one command, `--gpus N`, measures weak
        # scaling at 256 per GPU -- the same line carries the configs[4] shape too (a second engine per rank, the exchange
        # variant that was chosen above, the same timed-block protocol), so that an N-GPU record holds both
        wl4 = dict(WORKLOADS[wl_name], batch=CONFIGS4_GLOBAL_BATCH // world)
        eng4 = build_sac_engine(wl4, replay, rank, device)
        try:
            dp4, kind4, _ = make_data_parallel(eng4, dp_chosen if dp_chosen in IN_GRAPH_VARIANTS else "rccl", world, rank, device,
                                               init_parameters(eng4.table, seed=0))
        except SystemExit as exc:       # (collective: raised on every rank after the same votes) -> the collective library
            sys.stderr.write("bench[rank %d]: configs[4] pass: %s -- falling back to RCCL\n" % (rank, exc))
            dp4, kind4, _ = make_data_parallel(eng4, "rccl", world, rank, device, init_parameters(eng4.table, seed=0))

        def barrier4():
            eng4.synchronize()
            torch.cuda.synchronize(device)
            dist.barrier()
        t4, calls4 = timed_blocks(dp4.train, barrier4, args.steps, args.warmup, args.repeats, world, device)
        d4 = float(np.median(t4))
        out["config"]["configs4"] = {
            "workload": wl4["name"] % (wl4["batch"], replay) + " [configs[4]: global batch %d over %d GPU(s)]" % (CONFIGS4_GLOBAL_BATCH, world),
            "metric": "SAC grad-steps/sec (64x64 depth, global batch %d)" % CONFIGS4_GLOBAL_BATCH, "scaling": "strong",
            "global_batch": CONFIGS4_GLOBAL_BATCH, "per_gpu_batch": wl4["batch"], "value": round(args.steps / d4, 2),
            "unit": "grad-steps/s (global updates: each consumes the whole global batch)", "ms_per_step": round(1e3 * d4 / args.steps, 4),
            "parallelism": kind4,
            "repeats": {"n": args.repeats, "block_ms": [round(1e3 * t, 3) for t in t4], "value_is": "median block", **block_note(calls4, args.steps)}}
        if hasattr(dp4, "close"):
            dp4.close(disconnect=True)
        eng4.close()
    if wl_name == "sac_depth" and rank == 0 and world == 1 and not strong and not args.no_learn_loop:
        from grasp_rl import synthetic
        try:
            out["learn_loop"] = {
                "what": "SAC.learn on an env built as the reference's script builds it -- DummyVecEnv([one factory]) + VecNormalize -- "
                        "fanned out to 16 worker processes by GRL_NUM_ENVS=16 (free synthetic env; host loop cost: "
                        "pipes, running statistics, staging, act); default = one update per environment step, i.e. 16 "
                        "updates per loop iteration (stable-baselines' ratio on its single env); not part of `value`",
                # default SAC(...) arguments: strict step -> store -> update order; VecNormalize statistics on the device
                # because no callback reads the observations (device_norm="auto")
                "strict_order": synthetic.learn_loop_rate(16, args.learn_iters, 60, overlap=False, device=str(device)),
                "strict_order_host_statistics": synthetic.learn_loop_rate(16, args.learn_iters, 60, overlap=False,
                                                                          device=str(device), device_norm=False),
                "overlap_env_step": synthetic.learn_loop_rate(16, args.learn_iters, 60, overlap=True, device=str(device)),
                "overlap_env_step_host_statistics": synthetic.learn_loop_rate(16, args.learn_iters, 60, overlap=True,
                                                                              device=str(device), device_norm=False),
                "one_update_per_iteration": synthetic.learn_loop_rate(16, args.learn_iters, 60, overlap=True,
                                                                      device=str(device), gradient_steps=1),
                "one_update_per_iteration_host_statistics": synthetic.learn_loop_rate(
                    16, args.learn_iters, 60, overlap=True, device=str(device), gradient_steps=1, device_norm=False),
                "one_update_per_iteration_4_workers": synthetic.learn_loop_rate(
                    16, args.learn_iters, 60, overlap=True, device=str(device), gradient_steps=1, envs_per_worker=4)}
            out["learn_loop_updates_per_s"] = out["learn_loop"]["strict_order"]["updates_per_s"]
            out["learn_loop_steps_per_s"] = out["learn_loop"]["overlap_env_step"]["env_steps_per_s"]
        except Exception as e:      # the headline number must not depend on process spawning
            out["learn_loop"] = {"error": repr(e)}
    if wl_name == "sac_depth" and rank == 0 and world == 1 and not strong and not args.no_success:
        # the metric's second half ("grasp success"): the reference's pipeline (VecNormalize -> model.learn -> deterministic
        # evaluation, utils.py:10-44) on the learnable surrogate task, 16 envs, batch 256 -- see tests/test_gpu_learning.py
        from grasp_rl import synthetic
        try:
            r = synthetic.learn_reach("sac", "depth", total_timesteps=args.success_steps, n_envs=16, device=str(device))
            out["success_rate"] = {"task": "grasp_rl.synthetic.ReachGraspEnv (depth 64x64x2, 15-step episodes; random policy 0.07)",
                                   "env_steps": r["env_steps"], "updates": r["updates"], "episodes": r["episodes"],
                                   "train_success_last_200_episodes": r["train_success"], "eval_success_200_episodes": r["eval_success"],
                                   "eval_mean_distance": round(r["eval_distance"], 4), "seconds": r["seconds"]}
        except Exception as e:      # noqa: BLE001
            out["success_rate"] = {"error": repr(e)}
    return out


def run_bdq(args, device, prioritized=True):
    import numpy as np
    import torch
    from grasp_rl import _capi
    from grasp_rl.engine import QEngine
    replay = args.replay or 1_000_000
    cfg = _capi.make_q_config("bdq", 101, 5, 33, common=(64, 64), branch_hidden=(32,), value_hidden=(32,), batch_size=64,
                              replay_capacity=replay, lr=1e-4, prioritized=prioritized)
    eng = QEngine(cfg, device=str(device))
    rng = np.random.default_rng(0)
    P = {}
    for name, _, _, shape, _ in eng.table:
        if "/target_q_func/" not in name:
            P[name] = (rng.normal(0.0, 0.1, shape) if len(shape) >= 2 else np.zeros(shape)).astype(np.float32)
    for name, *_ in eng.table:
        if "/target_q_func/" in name:
            P[name] = P[name.replace("/target_q_func", "")].copy()
    eng.set_parameters(P)
    g = torch.Generator(device=device)
    g.manual_seed(0)
    for k0 in range(0, replay, 65536):
        m = min(65536, replay - k0)
        with torch.cuda.stream(eng.be.stream):
            eng.replay_add_device(torch.randn((m, 101), generator=g, device=device), torch.randint(0, 33, (m, 5), generator=g, device=device).float(),
                                  torch.randn(m, generator=g, device=device), torch.randn((m, 101), generator=g, device=device),
                                  (torch.rand(m, generator=g, device=device) < 1.0 / 15.0).float())
        eng.be.stream.synchronize()
    go = (lambda n: eng.train_per(n, beta=0.4)) if prioritized else (lambda n: eng.train_device(n))

    def barrier():
        eng.synchronize()
        torch.cuda.synchronize(device)
    times, calls = timed_blocks(go, barrier, args.steps, args.warmup, args.repeats, 1, device)
    dt = float(np.median(times))
    out = {"metric": "BDQ grad-steps/sec (101-d observations, 5 x 33 bins, batch 64, %s replay)" % ("prioritised" if prioritized else "uniform"),
           "value": round(args.steps / dt, 2), "unit": "grad-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[2]: gripper_grasp.yaml --algo BDQ (layers [[64,64],[32],[32]], num_actions_pad 33, batch "
                                  "64, lr 1e-4, %s) on 101-d auto-encoder observations, %d-transition ring%s in HBM, device RNG"
                                  % ("prioritized_replay: True" if prioritized else "prioritized_replay: False as gripper_grasp.yaml:106 ships it", replay,
                                     " + priorities" if prioritized else "")},
           "repeats": {"n": args.repeats, "block_ms": [round(1e3 * t, 3) for t in times], "value_is": "median block", **block_note(calls, args.steps)},
           "losses": {k: round(float(v), 6) for k, v in eng.metrics().items()}}
    if not args.no_profile:
        prof = profile_pass(eng, go, 50)
        out["roofline"] = roofline_of(prof, 50, dt / args.steps, "bdq_per" if prioritized else "bdq_uniform")
        out["roofline"]["note"] = "every launch of this update is latency-bound (0.01 GFLOP, < 5 MB): fractions are informational"
    out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline_bdq()
    eng.close()
    return out


def run_ae(args, device):
    import numpy as np
    import torch
    from grasp_rl.autoencoder import AeEngine, glorot_uniform_params
    eng = AeEngine(128, 2e-4, device=str(device))
    eng.set_parameters(glorot_uniform_params(0))
    x = torch.from_numpy(np.random.default_rng(0).uniform(0, 0.5, (128 * 8, 4096)).astype(np.float32)).to(device)
    import ctypes as C
    from grasp_rl._capi import check

    def go(n):          # n training steps on device-resident minibatches (8 distinct ones, cycled)
        done = 0
        while done < n:
            m = min(8, n - done)
            check(eng.lib, eng.lib.grl_ae_train_step(eng.h, C.c_void_p(x.data_ptr()), m))
            done += m

    def barrier():
        eng.synchronize()
        torch.cuda.synchronize(device)
    eng.be.stream.wait_stream(torch.cuda.current_stream(device))
    times, calls = timed_blocks(go, barrier, args.steps, args.warmup, args.repeats, 1, device)
    dt = float(np.median(times))
    out = {"metric": "auto-encoder train-steps/sec (64x64x1 depth, batch 128)", "value": round(args.steps / dt, 2),
           "unit": "train-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "SURVEY 8f-3: SimpleAutoEncoder fit step (encoders.py:40-50,70-136; config/encoder.yaml: batch 128, "
                                  "lr 2e-4): encoder + decoder forward, MSE, backward, Keras-Adam; minibatches resident in HBM"},
           "repeats": {"n": args.repeats, "block_ms": [round(1e3 * t, 3) for t in times], "value_is": "median block", **block_note(calls, args.steps)},
           "losses": {"reconstruction_mse": round(float(eng.metrics()["policy_loss"]), 6)}}
    if not args.no_profile:
        prof = profile_pass(eng, go, 16)
        out["roofline"] = roofline_of(prof, 16, dt / args.steps, "ae_train")
    out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline_ae()
    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps updates; value = median block")
    ap.add_argument("--workload", default="sac_depth", choices=["sac_depth", "sac_rgbd", "sac_depth_128", "sac_nature", "sac_mlp", "bdq_per", "bdq_uniform", "ae_train"])
    ap.add_argument("--global-batch", type=int, default=None, help="fix the GLOBAL batch (strong scaling); per-GPU batch = G / N")
    ap.add_argument("--replay", type=int, default=None)
    ap.add_argument("--learn-iters", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-learn-loop", action="store_true")
    ap.add_argument("--dp", default="auto", choices=["auto", "ingraph", "ingraph-oneshot", "ingraph-twoshot", "ingraph-overlap", "rccl", "rccl-overlap"],
                    help="exchange step for N > 1 (auto: the fastest in-graph IPC all-reduce that verifies -- one-shot, two-shot, two-shot overlapped -- else RCCL one bucket)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="validation aid: gloo lets the N-rank path run where RCCL cannot (ranks sharing one GPU)")
    ap.add_argument("--same-device", action="store_true",
                    help="validation aid: every rank uses cuda:0 (a 1-GPU box); the reported rate is then NOT a scaling number")
    ap.add_argument("--no-success", action="store_true", help="skip the learning run behind `success_rate`")
    ap.add_argument("--inject-ipc-failure", type=int, default=None, metavar="RANK",
                    help="validation aid: the in-graph exchange's peer mapping fails on this rank -> every rank must fall back together")
    ap.add_argument("--no-configs4", action="store_true", help="N > 1: skip the second measurement at BASELINE configs[4]'s global batch of 1024")
    ap.add_argument("--success-steps", type=int, default=80_000)
    args = ap.parse_args()

    if os.environ.get("GRL_LIBRARY"):
        raise SystemExit("bench.py measures the in-tree libgrl.so: unset GRL_LIBRARY (a test-only override)")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (exchange buffers): must be set before HIP initialises
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if world > 1 and not args.workload.startswith("sac_"):
        raise SystemExit("multi-GPU runs are defined for the SAC workloads")
    device = torch.device("cuda", 0 if args.same_device else local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from grasp_rl.parallel import loopback_gloo
        loopback_gloo()            # (single node: gloo on the loopback interface, no host-name lookups)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    if args.workload.startswith("sac_"):
        out = run_sac(args, args.workload, world, rank, device)
    elif args.workload in ("bdq_per", "bdq_uniform"):
        out = run_bdq(args, device, prioritized=args.workload == "bdq_per")
    else:
        out = run_ae(args, device)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
