"""SubprocVecEnv (BASELINE configs 2 / 5: N simulators on N host cores feeding one engine): same
observations, rewards, dones and auto-reset behaviour as DummyVecEnv on identically seeded envs."""
import functools

import numpy as np

from fake_env import FakeGraspEnv
from stable_baselines.common.vec_env import DummyVecEnv, SubprocVecEnv, VecNormalize


def _make(seed):
    return FakeGraspEnv(seed=seed, vector_dim=12, act_dim=3, episode_len=5)


def test_subproc_matches_dummy_and_supports_the_reference_calls():
    fns = [functools.partial(_make, s) for s in range(3)]
    ref = DummyVecEnv(fns)
    sub = SubprocVecEnv(fns)
    try:
        assert sub.num_envs == 3 and sub.observation_space.shape == (12,)
        o_ref, o_sub = ref.reset(), sub.reset()
        assert np.array_equal(o_ref, o_sub)
        rng = np.random.default_rng(0)
        saw_done = False
        for _ in range(12):
            a = rng.uniform(-1, 1, (3, 3)).astype(np.float32)
            r1, r2 = ref.step(a), sub.step(a)
            for x, y in zip(r1[:3], r2[:3]):
                assert np.array_equal(x, y)
            for i1, i2 in zip(r1[3], r2[3]):
                assert ("terminal_observation" in i1) == ("terminal_observation" in i2)
                if "terminal_observation" in i1:
                    saw_done = True
                    assert np.array_equal(i1["terminal_observation"], i2["terminal_observation"])
        assert saw_done
        # attribute / method access used by sb_helper.py:42-45 and the callbacks
        assert sub.get_attr("episode_len") == [5, 5, 5]
        sub.set_attr("episode_len", 9, indices=1)
        assert sub.get_attr("episode_len") == [5, 9, 5]
        assert sub.env_method("is_simplified") == [False, False, False]
        # VecNormalize wraps it like any VecEnv
        vn = VecNormalize(sub, norm_obs=True, norm_reward=True, clip_obs=10.)
        o = vn.reset()
        assert o.shape == (3, 12) and np.all(np.isfinite(o))
        o, r, d, info = vn.step(rng.uniform(-1, 1, (3, 3)).astype(np.float32))
        assert o.shape == (3, 12) and r.shape == (3,) and len(info) == 3
    finally:
        sub.close()
        ref.close()


def test_several_envs_per_worker_behave_like_one_each():
    """envs_per_worker=2 (5 envs on 3 worker processes): identical observations / rewards / resets / attribute routing."""
    fns = [functools.partial(_make, s) for s in range(5)]
    ref = DummyVecEnv(fns)
    sub = SubprocVecEnv(fns, envs_per_worker=2)
    try:
        assert sub.num_envs == 5 and len(sub.processes) == 3
        assert np.array_equal(ref.reset(), sub.reset())
        rng = np.random.default_rng(1)
        for _ in range(11):
            a = rng.uniform(-1, 1, (5, 3)).astype(np.float32)
            r1, r2 = ref.step(a), sub.step(a)
            for x, y in zip(r1[:3], r2[:3]):
                assert np.array_equal(x, y)
            for i1, i2 in zip(r1[3], r2[3]):
                assert ("terminal_observation" in i1) == ("terminal_observation" in i2)
        sub.set_attr("episode_len", 7, indices=[3, 0])
        assert sub.get_attr("episode_len") == [7, 5, 5, 7, 5]
        assert sub.get_attr("episode_len", indices=[4, 3]) == [5, 7]
        assert sub.env_method("is_simplified", indices=2) == [False]
    finally:
        sub.close()
        ref.close()


def test_a_dead_worker_raises_instead_of_hanging_and_close_still_releases_the_rest():
    """The per-step exchange reads the workers' pipes directly (one byte each way): a worker that died must surface as an
    exception in step_async / step_wait, not as a read that never returns, and close() must still join the others."""
    import pytest
    sub = SubprocVecEnv([functools.partial(_make, s) for s in range(3)])
    try:
        sub.reset()
        sub.step(np.zeros((3, 3), np.float32))
        sub.processes[1].terminate()
        sub.processes[1].join()
        with pytest.raises((EOFError, BrokenPipeError, ConnectionResetError)):
            for _ in range(3):          # (the first write after the death may still land in the pipe's buffer)
                sub.step_async(np.zeros((3, 3), np.float32))
                sub.step_wait()
    finally:
        sub.waiting = False
        sub.close()
    assert sub.closed


def test_workers_import_what_the_script_imported_when_it_runs_under_runpy(tmp_path):
    """`python -m grasp_rl.dp_run train_stable_baselines.py ...` executes the script with runpy.  Its factory -- `lambda:
    gym.make('gripper-env-v0', ...)` -- needs the registration that `import manipulation_main` performs at the script's top
    (train_stable_baselines.py:10) in every forkserver worker: the workers re-import the script (multiprocessing, from its file)
    and, belt and braces, the script's top-level modules (vec_env._main_module_imports).  The whole flow -- one factory,
    `fan_out(3)`, forkserver, runpy -- in a child interpreter."""
    import os
    import subprocess
    import sys
    import textwrap
    (tmp_path / "regmod.py").write_text("import fakegym\nfakegym.REGISTERED.add('gripper-env-v0')\n")
    (tmp_path / "fakegym.py").write_text(textwrap.dedent("""
        REGISTERED = set()
        def make(name):
            if name not in REGISTERED:
                raise KeyError("no registered env with id: " + name)
            from fake_env import FakeGraspEnv
            return FakeGraspEnv(seed=None, vector_dim=6, act_dim=2, episode_len=3)
        """))
    (tmp_path / "script.py").write_text(textwrap.dedent("""
        import numpy as np
        import fakegym
        import regmod                      # side effect: registers the env (like `import manipulation_main`)
        from stable_baselines.common.vec_env import DummyVecEnv
        if __name__ == "__main__":
            env = DummyVecEnv([lambda: fakegym.make('gripper-env-v0')])
            env.fan_out(3, start_method="forkserver")
            obs = env.reset()
            obs, rew, done, info = env.step(np.zeros((3, 2), np.float32))
            env.close()
            open("ok.txt", "w").write("%d %s" % (env.num_envs, obs.shape))
        """))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), os.path.join(repo, "deep-rl-grasping_amd"), os.path.join(repo, "tests")]))
    runner = "import runpy, sys; sys.argv = ['script.py']; runpy.run_path('script.py', run_name='__main__')"
    r = subprocess.run([sys.executable, "-c", runner], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / "ok.txt").read_text() == "3 (3, 6)"
