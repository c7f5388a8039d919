"""Auto-encoder training throughput (config/encoder.yaml: batch 128) -- development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd")):
    sys.path.insert(0, p)
import numpy as np
from grasp_rl.autoencoder import AeEngine, glorot_uniform_params
eng = AeEngine(128, 2e-4)
eng.set_parameters(glorot_uniform_params(0))
x = np.random.default_rng(0).uniform(0, 0.5, (128 * 4, 64, 64, 1)).astype(np.float32)
eng.train_batches(x); eng.synchronize()
t0 = time.perf_counter(); n = 5
for _ in range(n): eng.train_batches(x)
eng.synchronize(); dt = time.perf_counter() - t0
print("ae train: %.3f ms / step (batch 128), %.0f img/s, loss %.5f" % (1e3 * dt / (4 * n), 128 * 4 * n / dt, eng.metrics()["policy_loss"]))
eng.profile(True); eng.train_batches(x[:128]); eng.synchronize()
for k, v in sorted(eng.profile_dump().items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"]):
    print("  %-20s %8.3f ms x %d" % (k, v["avg_ms"], v["launches"]))
