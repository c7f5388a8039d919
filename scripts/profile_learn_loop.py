import cProfile, pstats, sys, os, io
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "deep-rl-grasping_amd")):
    sys.path.insert(0, p)
"""cProfile of SAC.learn's host loop (bench.py `learn_loop`): where the Python time of an iteration goes."""


def main():
    from grasp_rl import synthetic
    overlap = "--strict" not in sys.argv
    dn = "--device-norm" in sys.argv
    synthetic.learn_loop_rate(16, 50, 20, overlap=overlap, device="cuda:0", device_norm=dn)   # warm
    pr = cProfile.Profile()
    pr.enable()
    r = synthetic.learn_loop_rate(16, 300, 20, overlap=overlap, device="cuda:0", device_norm=dn)
    pr.disable()
    print(r)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue()[:6000])


if __name__ == "__main__":       # SubprocVecEnv uses the forkserver start method: the main module must be importable
    main()
