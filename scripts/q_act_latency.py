"""development: latency of grl_act on a BDQ handle (Q-values of ONE observation, configs[2] shape) with the polled completion counter\nand with GRL_TUNE=act_poll=0 (device-to-host copy + stream synchronisation).  Uses the test utilities for the set-up.  Run on the GPU box."""
import sys, time, os
sys.path.insert(0, "tests"); sys.path.insert(0, "deep-rl-grasping_amd"); sys.path.insert(0, ".")
import numpy as np
import q_parity_util as qu
def run(tune):
    os.environ["GRL_TUNE"] = tune
    case = qu.make_q_case(**qu.CASES["bdq_baseline_config3"])
    eng = qu.q_engine_setup(case)
    obs = case["tr"]["obs"][:1]
    for _ in range(200): eng.q_values(obs)
    t0 = time.perf_counter()
    for _ in range(2000): eng.q_values(obs)
    dt = (time.perf_counter() - t0) / 2000
    eng.close()
    return dt * 1e6
print("q act poll: %.1f us  sync: %.1f us" % (run(""), run("act_poll=0")))
print("q act poll: %.1f us  sync: %.1f us" % (run(""), run("act_poll=0")))
