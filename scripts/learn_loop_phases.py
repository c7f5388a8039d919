#!/usr/bin/env python
"""Where one iteration of SAC.learn goes on the host (strict order, statistics on the device, 16 SubprocVecEnv workers of the
free synthetic env): wall-clock stamps around the calls of the loop (perf_counter; no profiler), medians over the run.
`act` includes the wait for the 16 updates queued before it (the stream is in order): its excess over 16 x the engine's update
time is the act latency."""
import functools
import os
import sys
import time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd")):
    sys.path.insert(0, p)
import numpy as np


def main():
    from grasp_rl.synthetic import SyntheticGraspEnv
    from grasp_rl.sb.callbacks import BaseCallback
    from grasp_rl.sb.policies import AugmentedNatureCnn, SacCnnPolicy
    from grasp_rl.sb.sac import SAC
    from grasp_rl.sb.vec_env import SubprocVecEnv, VecNormalize
    n_envs, iters, warm = 16, 400, 60
    venv = SubprocVecEnv([functools.partial(SyntheticGraspEnv, "depth", 15, s) for s in range(n_envs)])
    try:
        env = VecNormalize(venv, norm_obs=True, norm_reward=True, clip_obs=10.0)
        model = SAC(SacCnnPolicy, env, policy_kwargs={"layers": [64, 64], "cnn_extractor": AugmentedNatureCnn(1)}, buffer_size=20000,
                    batch_size=256, learning_starts=256, overlap_env_step=False, device="cuda:0", device_norm=True)
        acc = {}

        def timed(obj, name, label=None):
            f = getattr(obj, name)

            def g(*a, **k):
                t0 = time.perf_counter()
                r = f(*a, **k)
                acc.setdefault(label or name, []).append(time.perf_counter() - t0)
                return r
            setattr(obj, name, g)
        eng = model.engine
        for name in ("act", "observe", "replay_add_observed", "train", "set_ret_var", "metrics"):
            timed(eng, name)
        timed(venv, "step_async", "venv.step_async")
        timed(venv, "step_wait", "venv.step_wait (workers step)")
        timed(env, "step_wait", "VecNormalize.step_wait (incl. venv + observe)")

        class Clock(BaseCallback):
            def _on_step(self):
                if self.n_calls == warm:
                    for v in acc.values():
                        v.clear()
                    self.model.engine.synchronize()
                    self.t0 = time.perf_counter()
                return True
        c = Clock()
        model.learn(total_timesteps=n_envs * (warm + iters), callback=c)
        eng.synchronize()
        dt = (time.perf_counter() - c.t0) / iters
        print("strict order, device statistics: %.1f us per iteration (16 updates)" % (1e6 * dt))
        tot = 0.0
        for k, v in acc.items():
            if not v:
                continue
            med, mean = float(np.median(v)), float(np.mean(v))
            per_iter = mean * len(v) / iters
            print("  %-48s calls/iter %.2f  median %7.1f us  mean %7.1f us  -> %7.1f us per iteration" % (k, len(v) / iters, 1e6 * med, 1e6 * mean, 1e6 * per_iter))
            if not k.startswith("venv.") and k != "observe":
                tot += per_iter
        print("  accounted (act + VecNormalize.step_wait + replay + train + small): %.1f us; rest = Python of the loop, callbacks" % (1e6 * tot))
    finally:
        venv.close()


if __name__ == "__main__":
    main()
