#!/usr/bin/env python
"""Schedule of the merged weight-gradient launch as the hardware ran it (measurement aid).

Needs the trace build of the library (scripts/tile_trace.sh) selected with GRL_LIBRARY=.../libgrl_trace.so.  Runs a
few SAC updates at the headline shape, reads the per-workgroup records {start, end, CU, XCD, tile} of the last
wgrad_conv launch and prints: makespan, per-problem tile durations, per-XCD and per-CU occupancy, start skew."""
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
lib = eng.lib
lib.grl_debug_tile_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(6 * 4096, np.uint64)
rc = lib.grl_debug_tile_trace(buf.ctypes.data, buf.size)
assert rc == 0, rc
rec = buf.reshape(-1, 6)
n = int(rec[0, 3] >> np.uint64(48))
rec = rec[:n]
t0 = rec[:, 0].astype(np.int64); t1 = rec[:, 1].astype(np.int64)
base = t0.min()
us = lambda x: (x - base) / 100.0                    # 100 MHz clock
hw = (rec[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
xcc = (rec[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
nslab = (rec[:, 2] >> np.uint64(40)).astype(np.int64) & 0xffff
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
prob = (rec[:, 3] & np.uint64(0xffff)).astype(np.int64)
dur = (t1 - t0) / 100.0
print("tiles %d   makespan %.1f us   (first start 0, last start %.1f, last end %.1f)" % (n, us(t1.max()), us(t0.max()), us(t1.max())))
tab = collections.Counter(zip((np.arange(n) % 8).tolist(), xcc.tolist()))
print("block b %% 8 -> XCC_ID (count):", {k: v for k, v in sorted(tab.items())})
print("distinct CUs used: %d" % len(set(cuid.tolist())))
sclk = (rec[:, 5].astype(np.int64) - rec[:, 4].astype(np.int64)) / np.maximum(1, (t1 - t0)) * 100.0
print("s_memtime ticks per us over the tiles (shader clock if s_memtime counts core cycles): mean %.0f  min %.0f  max %.0f" % (sclk.mean(), sclk.min(), sclk.max()))
x0 = np.where(xcc == xcc[0])[0]
print("first XCD, CU of its workgroups in block order:", cuid[x0][:96].tolist())
print("\nper problem (list order = launch order of first tile):")
seen = []
for p_ in prob:
    if p_ not in seen:
        seen.append(p_)
for p_ in seen:
    m = prob == p_
    print("  prob %2d  slabs %3d  tiles %4d  start %5.1f..%5.1f  dur mean %5.1f  min %5.1f  max %5.1f  end max %5.1f" %
          (p_, nslab[m].max(), m.sum(), us(t0[m]).min(), us(t0[m]).max(), dur[m].mean(), dur[m].min(), dur[m].max(), us(t1[m]).max()))
print("\nper XCD: tiles, busy-sum us, last end")
for x in range(8):
    m = xcc == x
    print("  xcd %d  tiles %4d  sum dur %7.1f  last end %5.1f  CUs %d" % (x, m.sum(), dur[m].sum(), us(t1[m]).max() if m.any() else 0, len(set(cuid[m].tolist()))))
per_cu = collections.defaultdict(list)
cu_slabs = collections.Counter()
for i in range(n):
    per_cu[int(cuid[i])].append((us(t0[i]), us(t1[i])))
    cu_slabs[int(cuid[i])] += int(nslab[i])
print("\nper CU: slabs held -> last end (us), MFMA-bound time = slabs x 0.427 us (16 MFMAs x 64 cycles per wave and slab at 2.4 GHz):")
rows = sorted((cu_slabs[c], max(e for _, e in per_cu[c]), len(per_cu[c])) for c in per_cu)
for lo in range(0, len(rows), 32):
    ch = rows[lo:lo + 32]
    sl = np.array([r[0] for r in ch]); en = np.array([r[1] for r in ch])
    print("  CUs %3d..%3d  slabs %3d..%3d  tiles %s  end mean %5.1f  MFMA-bound %5.1f  ratio %.2f" %
          (lo, lo + len(ch) - 1, sl.min(), sl.max(), sorted(set(r[2] for r in ch)), en.mean(), 0.427 * sl.mean(), 0.427 * sl.mean() / en.mean()))
print("total slabs %d -> %.1f us if spread evenly over 256 CUs" % (nslab.sum(), 0.427 * nslab.sum() / 256))
cnt = collections.Counter(len(v) for v in per_cu.values())
print("\ntiles per CU histogram:", dict(sorted(cnt.items())))
ends = sorted(max(e for _, e in v) for v in per_cu.values())
print("per-CU last end: min %.1f  median %.1f  p90 %.1f  max %.1f" % (ends[0], ends[len(ends) // 2], ends[int(0.9 * len(ends))], ends[-1]))
# how many workgroups are running at time t
grid = np.arange(0, us(t1.max()), 2.0)
print("\nrunning workgroups over time (2 us steps):")
print("  " + " ".join("%d" % int(((us(t0) <= g) & (us(t1) > g)).sum()) for g in grid))
