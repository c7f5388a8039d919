#!/usr/bin/env python
"""Evidence for the two-bucket data-parallel schedule on ONE GPU: run grasp_rl.parallel.DataParallelSac(overlap=True)
with a one-rank RCCL group, with the all-reduce of each bucket range replaced by a stand-in of the same byte count
on the same stream (a one-rank all-reduce moves nothing; the stand-in is a device copy + scale that keeps the
exchange stream busy for a few microseconds), under `rocprofv3 --kernel-trace`.  `--analyse <trace.csv>` then
reports, for every stand-in kernel, which engine kernels ran at the same time on the other stream.

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python scripts/dp_overlap_trace.py
    python scripts/dp_overlap_trace.py --analyse out/.../t_kernel_trace.csv
"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd")):
    sys.path.insert(0, p)


def analyse(path):
    rows = list(csv.DictReader(open(path)))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))) for r in rows]
    comm = [e for e in ev if "elementwise" in e[2] and "MulFunctor" in e[2]]
    eng = [e for e in ev if "grl::" in e[2]]
    n_ov, tot, hid = 0, 0.0, 0.0
    names = {}
    for s, e, _, _ in comm:
        tot += e - s
        ov = [(max(s, a), min(e, b), n) for a, b, n, _ in eng if a < e and b > s]
        if ov:
            n_ov += 1
            hid += sum(b - a for a, b, _ in ov)
            for _, _, n in ov:
                k = n.replace("void ", "").split("(")[0][:60]
                names[k] = names.get(k, 0) + 1
    print("stand-in exchange kernels: %d, overlapping an engine kernel: %d; exchange time %.1f us, of which concurrent with engine kernels %.1f us"
          % (len(comm), n_ov, tot / 1e3, hid / 1e3))
    for k, v in sorted(names.items(), key=lambda kv: -kv[1]):
        print("   overlapped %4d x %s" % (v, k))


def main():
    import torch
    import torch.distributed as dist
    import bench
    from grasp_rl import _capi, parallel
    from grasp_rl.engine import SacEngine
    from grasp_rl.init import init_parameters
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    cfg = _capi.make_config("augmented", obs_channels=2, n_direct=1, act_dim=5, layers=(64, 64), batch_size=128,
                            replay_capacity=20000, normalize=True, act_batch=16, seed=1)      # 128 / GPU: configs[4]'s per-rank batch
    eng = SacEngine(cfg, device=str(dev))
    eng.set_parameters(init_parameters(eng.table, seed=0))
    st = bench.fill_replay_on_device(eng, 20000, 100, dev)
    eng.set_obs_stats(st["mean"], st["var"], st["ret_var"])
    scratch = torch.empty(2_000_000, device=dev)

    def stand_in(flat, group=None):          # same bytes as the bucket range, on whatever stream is current
        for _ in range(4):
            torch.mul(flat, 1.0, out=scratch[: flat.numel()])
        return flat
    parallel.allreduce_flat_ = stand_in
    parallel.allreduce_ranges_ = lambda views, group=None: [stand_in(v) for v in views]
    dp = parallel.DataParallelSac(eng, overlap=True)
    assert dp.overlap
    dp.train(60)
    eng.synchronize()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("done")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        analyse(sys.argv[2])
    else:
        main()
