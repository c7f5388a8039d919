"""Seeded synthetic replay contents with the statistics of the reference's real runs.

Per-pixel observation statistics come from the VecNormalize pickles the reference ships
(``trained_models/SAC_depth_1mbuffer/best_model/vecnormalize.pkl`` for depth,
``trained_models/SAC_full_rgbd/vecnormalize.pkl`` for RGB-D); ``scripts/make_golden.py`` extracted
them into ``data/obs_stats_{depth,rgbd}.npz``.  Rewards mimic the shaped reward of
``config/gripper_grasp.yaml:39-47``.
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_obs_stats(kind="depth"):
    """dict(mean[64,64,C], var[64,64,C] float64, ret_var, count) in the env's HWC layout."""
    z = np.load(os.path.join(_DATA, "obs_stats_%s.npz" % kind))
    return {"mean": z["mean"], "var": z["var"], "ret_var": float(z["ret_var"]), "count": float(z["count"])}


def make_transitions(n, kind="depth", act_dim=5, seed=0, stats=None):
    """n raw (un-normalised) transitions in env layout: obs [n,64,64,C] float32 etc."""
    stats = stats or load_obs_stats(kind)
    rng = np.random.default_rng(seed)
    mean, var = stats["mean"], stats["var"]
    C = mean.shape[-1]

    def draw():
        o = rng.normal(mean, np.sqrt(var), (n,) + mean.shape).astype(np.float32)
        if kind == "rgbd":
            o[..., :3] = np.clip(np.round(o[..., :3]), 0, 255)
            o[..., 3] = np.clip(o[..., 3], 0.02, 2.0)
        else:
            o[..., 0] = np.clip(o[..., 0], 0.02, 2.0)
        o[..., C - 1] = 0.0                                   # sensor pad channel (robot.py:199-204)
        o[:, 0, 0, C - 1] = rng.uniform(0.0, 1.0, n)          # gripper width / 0.1 in pixel [0,0]
        return o

    obs, nxt = draw(), draw()
    act = rng.uniform(-1, 1, (n, act_dim)).astype(np.float32)
    u = rng.random(n)
    rew = np.where(u < 0.80, -200.0,
                   np.where(u < 0.99, -100.0 + 1000.0 * rng.uniform(-0.03, 0.03, n), 10000.0)).astype(np.float32)
    done = (rng.random(n) < 1.0 / 15.0).astype(np.float32)
    return {"obs": obs, "act": act, "rew": rew, "next_obs": nxt, "done": done}


def make_vector_transitions(n, mean, var, act_dim=5, seed=0):
    """Vector-observation variant (auto-encoder features, sacMlp)."""
    rng = np.random.default_rng(seed)
    obs = rng.normal(mean, np.sqrt(var), (n,) + mean.shape).astype(np.float32)
    nxt = rng.normal(mean, np.sqrt(var), (n,) + mean.shape).astype(np.float32)
    act = rng.uniform(-1, 1, (n, act_dim)).astype(np.float32)
    rew = rng.normal(-150.0, 300.0, n).astype(np.float32)
    done = (rng.random(n) < 1.0 / 15.0).astype(np.float32)
    return {"obs": obs, "act": act, "rew": rew, "next_obs": nxt, "done": done}


def make_noise(n_steps, batch, act_dim, replay_size, seed=1):
    """Explicit minibatch indices (with replacement, A.1 step 5) and policy noise streams."""
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, replay_size, (n_steps, batch), dtype=np.int64)
    eps = rng.standard_normal((n_steps, batch, act_dim)).astype(np.float32)
    return idx, eps


class SyntheticGraspEnv:
    """Stand-in for the reference's PyBullet 'gripper-env-v0' (manipulation_main/gripperEnv/robot.py) for
    throughput measurements of the learn loop: the same observation / action spaces (robot.py:207-228,
    actuator.py:54-89) and the attributes sb_helper.py reads, observations drawn from the per-pixel statistics of
    the reference's real runs, no physics.  One instance per worker process of a ``SubprocVecEnv``; it measures
    what the HOST LOOP around the engine costs (pipes, VecNormalize, staging), not PyBullet."""

    def __init__(self, kind="depth", episode_len=15, seed=0, act_dim=5, pool=64):
        from .sb.spaces import Box
        self.stats = load_obs_stats(kind)
        C = self.stats["mean"].shape[-1]
        self.kind = kind
        self.observation_space = Box(0, 255, shape=(64, 64, C), dtype=np.float32)
        self.action_space = Box(-1.0, 1.0, shape=(act_dim,), dtype=np.float32)
        self.depth_obs, self.full_obs = (kind == "depth"), (kind == "rgbd")
        self.episode_len, self.episode_step, self.episode_rewards = episode_len, 0, 0.0
        self.history, self.sr_mean = [], 0.0
        self.curriculum = type("Curriculum", (), {"_lambda": 0.0})()
        rng = np.random.default_rng(seed)
        m, v = self.stats["mean"], self.stats["var"]
        self._pool = rng.normal(m, np.sqrt(v), (pool,) + m.shape).astype(np.float32)   # pre-drawn frames: the env is free
        if kind == "rgbd":
            self._pool[..., :3] = np.clip(np.round(self._pool[..., :3]), 0, 255)
        self._pool[..., C - 1] = 0.0
        self._pool[:, 0, 0, C - 1] = rng.uniform(0, 1, pool)
        self._k = 0
        self._rng = rng

    def is_simplified(self):
        return False

    def _obs(self):
        self._k = (self._k + 1) % self._pool.shape[0]
        return self._pool[self._k]

    def reset(self):
        self.episode_step, self.episode_rewards = 0, 0.0
        return self._obs()

    def step(self, action):
        self.episode_step += 1
        r = float(-200.0 + 100.0 * np.tanh(np.sum(action)))
        self.episode_rewards += r
        done = self.episode_step >= self.episode_len
        if done:
            self.history.append(1)
            self.sr_mean = 1.0
        return self._obs(), r, done, {"is_success": done, "episode_step": self.episode_step,
                                      "episode_rewards": self.episode_rewards, "status": 1}

    def close(self):
        pass


class ReachGraspEnv:
    """LEARNABLE surrogate of the reference's grasping task, for evidence that the update path learns (the metric's
    second half, "grasp success": `manipulation_main/utils.py:10-44` prints `Mean success rate` from `info['is_success']`;
    PyBullet is not available here, so the physics is replaced by a task with the same interface and a known optimum).

    An object ("blob", a disc 0.3 m closer to the camera than the table) sits at p in [-0.8, 0.8]^2 of the camera image;
    the first two action dimensions are the commanded x / y of the gripper (the reference's action = [dx, dy, dz, dyaw,
    open], actuator.py:54-89).  Reward per step = 1 - |a_xy - p| (dense, like the shaped reward of gripper_grasp.yaml:
    39-47 in spirit; centred so that a random policy's return is ~0); the object stays put for one 15-step episode;
    the gripper closes linearly over the episode, so the width feature (pixel [0, 0] of the pad channel, robot.py:
    199-204) is 1 - t/T -- with a fixed horizon and an unobservable clock the bootstrapped targets carry a time-to-go
    term 100x the variance of the action-dependent reward (measured with the CPU oracle as the learner,
    scripts/learn_check_oracle.py: no learning in 40 000 updates without it, 0.89 success after 20 000 with it).
    `is_success` = the last command of the episode is within `tol` of the object.  A uniformly random policy succeeds in
    ~7 % of the episodes (tol 0.3).

    kind 'depth'  -> obs [64, 64, 2] float32 exactly as robot.py:183-205 builds it (depth image + pad channel carrying the
                     gripper width in pixel [0, 0]), Box(0, 255) so the SB CNN policies accept it;
    kind 'vector' -> obs [101]: the 100-d 'auto-encoder feature' stand-in (two 50-bin Gaussian bumps at p_x, p_y) + the
                     width, the shape sensor.py:220-222 hands to the MLP policies and to DQN / BDQ.
    action 'box' (SAC, BDQ: Box(-1, 1, [5])) or 'discrete' (DQN: Discrete(n); action k commands x = -1 + 2k/(n-1) and the
    reward only looks at p_x)."""

    def __init__(self, kind="depth", episode_len=15, seed=0, act_dim=5, action="box", n_discrete=12, tol=0.3):
        from .sb.spaces import Box, Discrete
        self.kind, self.episode_len, self.tol = kind, int(episode_len), float(tol)
        if kind == "depth":
            self.observation_space = Box(0, 255, shape=(64, 64, 2), dtype=np.float32)
        else:
            self.observation_space = Box(-np.inf, np.inf, shape=(101,), dtype=np.float32)
        self.discrete = action == "discrete"
        self.action_space = Discrete(n_discrete) if self.discrete else Box(-1.0, 1.0, shape=(act_dim,), dtype=np.float32)
        self.n_discrete = n_discrete
        self.depth_obs, self.full_obs = (kind == "depth"), False
        self.episode_step, self.episode_rewards = 0, 0.0
        self.history, self.sr_mean = [], 0.0
        self.curriculum = type("Curriculum", (), {"_lambda": 0.0})()
        self._rng = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:64, 0:64].astype(np.float32)
        self._yy, self._xx = yy, xx
        self._centres = np.linspace(-1.0, 1.0, 50, dtype=np.float32)
        self._p = np.zeros(2, np.float32)
        self._width = 0.5

    def is_simplified(self):
        return False

    def _obs(self):
        if self.kind == "depth":
            cx, cy = 31.5 + 30.0 * self._p[0], 31.5 + 30.0 * self._p[1]
            img = 0.6 + 0.01 * self._rng.standard_normal((64, 64)).astype(np.float32)
            img[(self._xx - cx) ** 2 + (self._yy - cy) ** 2 <= 36.0] = 0.3
            o = np.zeros((64, 64, 2), np.float32)
            o[..., 0] = img
            o[0, 0, 1] = self._width
            return o
        o = np.empty(101, np.float32)
        o[:50] = np.exp(-0.5 * ((self._centres - self._p[0]) / 0.08) ** 2)
        o[50:100] = np.exp(-0.5 * ((self._centres - self._p[1]) / 0.08) ** 2)
        o[100] = self._width
        return o + 0.01 * self._rng.standard_normal(o.shape).astype(np.float32)

    def reset(self):
        self.episode_step, self.episode_rewards = 0, 0.0
        self._p = self._rng.uniform(-0.8, 0.8, 2).astype(np.float32)
        self._width = 1.0
        return self._obs()

    def distance(self, action):
        if self.discrete:
            return abs(-1.0 + 2.0 * int(action) / (self.n_discrete - 1) - float(self._p[0]))
        a = np.clip(np.asarray(action, np.float32).reshape(-1)[:2], -1.0, 1.0)
        return float(np.sqrt(np.sum((a - self._p) ** 2)))

    def step(self, action):
        self.episode_step += 1
        d = self.distance(action)
        r = 1.0 - d
        self.episode_rewards += r
        done = self.episode_step >= self.episode_len
        self._width = 1.0 - self.episode_step / float(self.episode_len)
        ok = bool(d < self.tol)
        if done:
            self.history.append(int(ok))
            self.sr_mean = float(np.mean(self.history[-100:]))
        return self._obs(), r, done, {"is_success": ok, "episode_step": self.episode_step,
                                      "episode_rewards": self.episode_rewards, "status": int(ok), "distance": d}

    def close(self):
        pass


def learn_loop_rate(n_envs=16, iterations=300, warm=60, batch_size=256, buffer_size=20000, overlap=True, device="cuda:0",
                    gradient_steps=None, device_norm=None, envs_per_worker=1, via="cli"):
    """Env-steps / second and updates / second of ``SAC.learn`` with `n_envs` SyntheticGraspEnv worker processes behind
    SubprocVecEnv + VecNormalize -- BASELINE configs[1]: "16 vectorised PyBullet envs feed a single GPU", with the
    simulator replaced by a free one.  via="cli" (default): the env is built the way the reference's unmodified script
    builds it -- ``DummyVecEnv([ONE factory])`` (train_stable_baselines.py:52-54) wrapped in VecNormalize
    (sb_helper.py:117-119) -- and reaches its `n_envs` workers through GRL_NUM_ENVS when the model is constructed;
    via="subproc": a ``SubprocVecEnv`` built by hand.  gradient_steps None = one update per ENVIRONMENT step (stable-baselines' ratio
    with train_freq 1 / gradient_steps 1 on its single env, sb_helper.py:120-128: n_envs updates per loop iteration);
    1 = one update per loop iteration.  Returns a dict."""
    import functools
    import time
    from .sb.callbacks import BaseCallback
    from .sb.policies import AugmentedNatureCnn, SacCnnPolicy
    from .sb.sac import SAC
    from .sb.vec_env import DummyVecEnv, SubprocVecEnv, VecNormalize

    class Clock(BaseCallback):
        reads_observations = False

        def __init__(self, warm):
            super().__init__()
            self.warm, self.t0, self.u0 = warm, None, 0

        def _on_step(self):
            if self.n_calls == self.warm:
                self.model.engine.synchronize()
                self.t0, self.u0 = time.perf_counter(), self.model.n_updates
            return True

    saved = {k: os.environ.get(k) for k in ("GRL_NUM_ENVS", "GRL_ENVS_PER_WORKER")}
    if via == "cli":
        venv = DummyVecEnv([functools.partial(SyntheticGraspEnv, "depth", 15, None)])     # seed None: every worker draws its own
        os.environ["GRL_NUM_ENVS"], os.environ["GRL_ENVS_PER_WORKER"] = str(n_envs), str(envs_per_worker)
    else:
        venv = SubprocVecEnv([functools.partial(SyntheticGraspEnv, "depth", 15, s) for s in range(n_envs)],
                             envs_per_worker=envs_per_worker)
        os.environ.pop("GRL_NUM_ENVS", None)
    try:
        env = VecNormalize(venv, norm_obs=True, norm_reward=True, clip_obs=10.0)
        model = SAC(SacCnnPolicy, env, policy_kwargs={"layers": [64, 64], "cnn_extractor": AugmentedNatureCnn(1)},
                    buffer_size=buffer_size, batch_size=batch_size, learning_starts=max(batch_size, n_envs),
                    overlap_env_step=overlap, device=device, gradient_steps=gradient_steps, device_norm=device_norm)
        assert model.n_envs == n_envs and env.num_envs == n_envs
        clock = Clock(warm)
        model.learn(total_timesteps=n_envs * (warm + iterations), callback=clock)
        model.engine.synchronize()
        dt = time.perf_counter() - clock.t0
        return {"n_envs": n_envs, "iterations": iterations, "overlap_env_step": bool(overlap),
                "gradient_steps": n_envs if gradient_steps is None else gradient_steps, "device_norm": "auto (on: no callback reads the observations)" if device_norm is None else bool(device_norm),
                "worker_processes": (n_envs + envs_per_worker - 1) // envs_per_worker,
                "env_built_as": "DummyVecEnv([fn]) + GRL_NUM_ENVS" if via == "cli" else "SubprocVecEnv",
                "env_steps_per_s": round(n_envs * iterations / dt, 1),
                "updates_per_s": round((model.n_updates - clock.u0) / dt, 1),
                "ms_per_iteration": round(1e3 * dt / iterations, 3)}
    finally:
        venv.close()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# ------------------------------------------------------------------------------------------------ learning evidence
def evaluate_success(model, kind="depth", action="box", episodes=200, seed=10_000, n_discrete=12, vec_normalize=None):
    """`Mean success rate` of the reference's evaluation loop (manipulation_main/utils.py:10-44: deterministic
    `model.predict`, `info['is_success']` of the last step) over fresh ReachGraspEnv episodes.  Observations are
    normalised with the frozen statistics of `vec_normalize` (what `run(args)` does with the saved vecnormalize.pkl)."""
    env = ReachGraspEnv(kind, seed=seed, action=action, n_discrete=n_discrete)
    ok, dist = [], []
    for _ in range(episodes):
        obs, done = env.reset(), False
        while not done:
            o = vec_normalize.normalize_obs(obs[None])[0] if vec_normalize is not None else obs
            act, _ = model.predict(o, deterministic=True)
            obs, _, done, info = env.step(act)
        ok.append(float(info["is_success"]))
        dist.append(info["distance"])
    return float(np.mean(ok)), float(np.mean(dist))


def learn_reach(algo="sac", kind="depth", total_timesteps=30_000, n_envs=16, device="cuda:0", seed=0, engine_factory=None,
                eval_episodes=200, **model_kwargs):
    """Train `algo` ('sac' | 'dqn' | 'bdq') on ReachGraspEnv through the stable-baselines surface the reference drives
    (sb_helper.py:104-128 / 159-165 / 210-224: VecNormalize(norm_obs, norm_reward, clip_obs 10) + model.learn), then run the
    reference's evaluation.  Returns dict(train_success = mean is_success of the last 200 TRAINING episodes (the
    reference's monitor column `s`, trained_models/*/log_file.monitor.csv), eval_success, eval_distance, updates, seconds)."""
    import time
    from .sb.callbacks import BaseCallback
    from .sb.dqn import BDQ, DQN
    from .sb.policies import AugmentedNatureCnn, SacCnnPolicy, SacMlpPolicy
    from .sb.sac import SAC
    from .sb.vec_env import DummyVecEnv, VecNormalize

    class Successes(BaseCallback):
        def __init__(self):
            super().__init__()
            self.s = []

        def _on_step(self):
            for d, info in zip(np.atleast_1d(self.locals["done"]), self.locals["info"]):
                if d:
                    self.s.append(float(info["is_success"]))
            return True

    action = "discrete" if algo == "dqn" else "box"
    N = n_envs if algo == "sac" else 1
    venv = DummyVecEnv([(lambda s=s: ReachGraspEnv(kind, seed=seed * 1000 + s, action=action)) for s in range(N)])
    env = VecNormalize(venv, norm_obs=True, norm_reward=True, clip_obs=10.0)
    if algo == "sac":
        if kind == "depth":
            policy, pk = SacCnnPolicy, {"layers": [64, 64], "cnn_extractor": AugmentedNatureCnn(1)}
        else:
            policy, pk = SacMlpPolicy, {"layers": [64, 64]}
        kw = dict(buffer_size=100_000, batch_size=256, learning_starts=max(256, N), seed=seed)
        kw.update(model_kwargs)
        cls = SAC
        args = (policy, env)
        kw["policy_kwargs"] = pk
    elif algo == "dqn":     # config/gripper_grasp.yaml DQN block: lr 1e-3, batch 32, prioritized_replay
        kw = dict(learning_rate=1e-3, batch_size=32, prioritized_replay=True, seed=seed)
        kw.update(model_kwargs)
        cls, args = DQN, ("MlpPolicy", env)
    else:                   # ... BDQ block: lr 1e-4, batch 64, layers [[64,64],[32],[32]], 33 bins, eps 0.3 -> 0.1
        kw = dict(learning_rate=1e-4, batch_size=64, buffer_size=100_000, num_actions_pad=33, learning_starts=1000,
                  target_network_update_freq=1000, exploration_fraction=0.3, exploration_final_eps=0.1,
                  prioritized_replay=True, policy_kwargs={"layers": [[64, 64], [32], [32]]}, seed=seed)
        kw.update(model_kwargs)
        cls, args = BDQ, ("MlpActPolicy", env)
    old = cls.__dict__.get("_engine_factory")          # (the staticmethod object itself, not the unwrapped function)
    if engine_factory is not None:
        cls._engine_factory = staticmethod(engine_factory)
    try:
        model = cls(*args, device=device, **kw)
    finally:
        if engine_factory is not None:
            if old is not None:
                cls._engine_factory = old
            else:
                del cls._engine_factory
    cb = Successes()
    t0 = time.perf_counter()
    model.learn(total_timesteps=total_timesteps, callback=cb)
    model.engine.synchronize()
    secs = time.perf_counter() - t0
    env.training = False
    ev, dist = evaluate_success(model, kind, action, eval_episodes, vec_normalize=env)
    out = {"algo": algo, "kind": kind, "env_steps": int(model.num_timesteps), "updates": int(model.n_updates),
           "episodes": len(cb.s), "train_success": float(np.mean(cb.s[-200:])) if cb.s else 0.0,
           "eval_success": ev, "eval_distance": dist, "seconds": round(secs, 2), "metrics": model.engine.metrics()}
    model.engine.close()
    return out
