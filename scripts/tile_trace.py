#!/usr/bin/env python
"""Schedule of the implicit-GEMM launches of one SAC update as the hardware ran them (measurement aid).

Needs the trace build of the library (scripts/tile_trace.sh) selected with GRL_LIBRARY=.../libgrl_trace.so.  Runs a
few SAC updates at the headline shape, reads the per-workgroup records {start, first barrier, loop done, end, CU, XCD,
tile, slabs} of every igemm2 launch of the last update and prints per launch: makespan, phase times, reduction slabs per
CU (average / fullest), the time the fullest CU needs at the MFMA rate.  `--detail TAG` adds the per-problem and per-CU
tables for one launch (default wgrad_conv)."""
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd")):
    sys.path.insert(0, p)
import numpy as np
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
st = bench.fill_replay_on_device(eng, 20000, 100, dev)
eng.set_obs_stats(st["mean"], st["var"], st["ret_var"])
with torch.cuda.stream(eng.be.stream):
    eng.train_device(50)
    eng.synchronize()
detail = sys.argv[sys.argv.index("--detail") + 1] if "--detail" in sys.argv else "wgrad_conv"
TAGS = ["conv2_fwd", "conv3_fwd", "fc_fwd", "heads_l0", "heads_dfeat", "fc_bwd", "conv3_bwd", "conv2_bwd", "wgrad_conv",
        "wgrad_dense", "wgrad_small", "heads_fwd", "heads_bwd"]
SLAB_US = 1024 / 2400.0      # 16 dependent 32x32x2 MFMAs (64 cycles each) per wave and 32-deep slab at 2.4 GHz (64x64, 128x32
                             # shapes); the 32x64 shapes split the reduction over wave pairs / quads: 8 MFMAs per wave and slab


def load(name):
    raw = eng.fetch(name)
    rec = np.frombuffer(raw.tobytes(), np.uint64).reshape(-1, 6)
    return rec[rec[:, 1] != 0]


print("%-12s %5s %8s | %-17s %-13s %-17s %8s | %-15s %9s" % ("launch", "tiles", "makespan", "start->barrier", "loop/slab", "epilogue", "slabs", "per CU avg/max", "max@MFMA"))
for seq in range(40):
    for tag in TAGS:
        try:
            rec = load("trace%d_%s" % (seq, tag))
        except Exception:
            continue
        if rec.shape[0] == 0:
            continue
        n = rec.shape[0]
        t0 = rec[:, 0].astype(np.int64); t1 = rec[:, 1].astype(np.int64)
        tb = rec[:, 4].astype(np.int64); tl_ = rec[:, 5].astype(np.int64)
        base = t0.min()
        us = lambda x: (x - base) / 100.0
        hw = (rec[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
        xcc = (rec[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
        nslab = (rec[:, 2] >> np.uint64(40)).astype(np.int64) & 0xffff
        cfg = int((rec[0, 2] >> np.uint64(56)) & np.uint64(0xf))
        slab_us = SLAB_US * (0.5 if cfg in (2, 3) else 1.0)
        cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
        cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        prob = (rec[:, 3] & np.uint64(0xffff)).astype(np.int64)
        dur = (t1 - t0) / 100.0
        st = (tb - t0) / 100.0; lp = (tl_ - tb) / 100.0; ep = (t1 - tl_) / 100.0
        cu_slabs = collections.Counter()
        per_cu = collections.defaultdict(list)
        for i in range(n):
            cu_slabs[int(cuid[i])] += int(nslab[i])
            per_cu[int(cuid[i])].append((us(t0[i]), us(t1[i])))
        mx = max(cu_slabs.values())
        print("%-12s %5d %7.1f  | %4.1f (%4.1f..%4.1f)  %5.1f / %.2f  %4.1f (%4.1f..%4.1f) %8d | %5.1f / %3d (%3d CUs) %6.1f us" %
              (tag, n, us(t1.max()), st.mean(), st.min(), st.max(), lp.mean(), (lp / np.maximum(1, nslab)).mean(), ep.mean(), ep.min(), ep.max(),
               nslab.sum(), nslab.sum() / 256.0, mx, len(cu_slabs), mx * slab_us))
        if tag != detail:
            continue
        print("  late starts: last start %.1f us" % us(t0.max()))
        print("  per problem (launch order of first tile):")
        seen = []
        for p_ in prob:
            if p_ not in seen:
                seen.append(p_)
        for p_ in seen[:12]:
            m = prob == p_
            print("    prob %2d  slabs %3d  tiles %4d  dur mean %5.1f  min %5.1f  max %5.1f  end max %5.1f" %
                  (p_, nslab[m].max(), m.sum(), dur[m].mean(), dur[m].min(), dur[m].max(), us(t1[m]).max()))
        print("  per CU: slabs held -> last end (us)")
        rows = sorted((cu_slabs[c], max(e for _, e in per_cu[c]), len(per_cu[c])) for c in per_cu)
        for lo in range(0, len(rows), 32):
            ch = rows[lo:lo + 32]
            sl = np.array([r[0] for r in ch]); en = np.array([r[1] for r in ch])
            print("    CUs %3d..%3d  slabs %3d..%3d  tiles %s  end mean %5.1f  MFMA-bound %5.1f  ratio %.2f" %
                  (lo, lo + len(ch) - 1, sl.min(), sl.max(), sorted(set(r[2] for r in ch)), en.mean(), slab_us * sl.mean(), slab_us * sl.mean() / en.mean()))
        grid = np.arange(0, us(t1.max()), 2.0)
        print("  running workgroups over time (2 us steps): " + " ".join("%d" % int(((us(t0) <= g) & (us(t1) > g)).sum()) for g in grid))
