"""``DQN`` and ``BDQ`` with the constructor / learn / predict / save / load surface the reference uses
(/root/reference/manipulation_main/training/sb_helper.py:159-165 ``sb.DQN(DQNMlpPolicy, env, verbose,
gamma, batch_size, prioritized_replay, tensorboard_log)``; :202-226 ``sb.BDQ(MlpActPolicy, env, ...,
policy_kwargs={'layers': [[common],[branch],[value]]}, epsilon_greedy, exploration_fraction,
exploration_final_eps, num_actions_pad, learning_starts, target_network_update_freq,
prioritized_replay)``; loaders at train_stable_baselines.py:101-104).

The network update (dueling towers, double-Q target, Huber / squared TD loss, per-variable gradient
clipping, Adam, hard target copy) runs in the HIP engine (``QEngine``); this file is the host loop:
epsilon-greedy exploration, replay bookkeeping and -- when ``prioritized_replay=True`` -- the
proportional prioritised sampler (sum-tree on the host, as stable-baselines does; a device-side tree
is SURVEY.md 8f row 4).  DQN follows stock stable-baselines 2.10.1; BDQ follows the decisions
documented in SURVEY.md A.6 / oracle/dqn.py (the fork's source is unavailable: parity unpinned).

Data parallelism (BASELINE north_star: "SAC / BDQ / DQN update ... optionally sharded"):
``data_parallel="auto"`` (or GRL_DATA_PARALLEL=auto) under a torchrun launch trains one replica per rank, as ``sb.SAC``
does -- every rank steps its OWN environment and fills its own replay ring, the global minibatch is split evenly over
the ranks, the gradient SUM is exchanged inside the update's graph (``grasp_rl.parallel.DataParallelInGraph``; a
collective library as fallback), clipped per variable as the mean of the replicas and applied identically everywhere.
Counters (``num_timesteps``, exploration and beta schedules, ``learning_starts``, ``train_freq``,
``target_network_update_freq``) count environment steps of the JOB: W per loop iteration.  Callbacks, logging and
the stop decision are rank 0's.  Prioritised replay keeps one priority tree per rank (importance weights against the
rank's own total and minimum) and needs the in-graph exchange.
"""
import time

import numpy as np

from .. import _capi
from ..engine import QEngine
from . import logger
from . import policies as pol
from . import save_util
from . import spaces as sp
from .callbacks import as_callback
from .vec_env import DummyVecEnv, VecEnv, unwrap_vec_normalize


class LinearSchedule:
    def __init__(self, schedule_timesteps, final_p, initial_p=1.0):
        self.schedule_timesteps, self.final_p, self.initial_p = max(1, int(schedule_timesteps)), final_p, initial_p

    def value(self, step):
        frac = min(float(step) / self.schedule_timesteps, 1.0)
        return self.initial_p + frac * (self.final_p - self.initial_p)


class _QModel:
    """Shared host loop of DQN and BDQ."""
    _engine_factory = staticmethod(lambda cfg, device: QEngine(cfg, device=device))
    algo = "dqn"

    def __init__(self, policy, env, gamma=0.99, learning_rate=5e-4, buffer_size=50000, exploration_fraction=0.1,
                 exploration_final_eps=0.02, exploration_initial_eps=1.0, train_freq=1, batch_size=32, double_q=True,
                 learning_starts=1000, target_network_update_freq=500, prioritized_replay=False,
                 prioritized_replay_alpha=0.6, prioritized_replay_beta0=0.4, prioritized_replay_beta_iters=None,
                 prioritized_replay_eps=1e-6, param_noise=False, n_cpu_tf_sess=None, verbose=0, tensorboard_log=None,
                 _init_setup_model=True, policy_kwargs=None, full_tensorboard_log=False, seed=None, device="cuda:0",
                 data_parallel=None, dp_exchange="ingraph"):
        if param_noise:
            raise NotImplementedError("param_noise is not implemented")
        import os
        if data_parallel is None:
            data_parallel = os.environ.get("GRL_DATA_PARALLEL") or None
        self.data_parallel, self.dp_exchange = data_parallel, dp_exchange
        self._dp_rt = self._dp = None
        self.policy, self.policy_kwargs = policy, ({} if policy_kwargs is None else dict(policy_kwargs))
        self.gamma, self.learning_rate, self.buffer_size = gamma, learning_rate, int(buffer_size)
        self.exploration_fraction, self.exploration_final_eps = exploration_fraction, exploration_final_eps
        self.exploration_initial_eps, self.train_freq, self.batch_size = exploration_initial_eps, train_freq, int(batch_size)
        self.double_q, self.learning_starts = double_q, learning_starts
        self.target_network_update_freq = target_network_update_freq
        self.prioritized_replay, self.prioritized_replay_alpha = prioritized_replay, prioritized_replay_alpha
        self.prioritized_replay_beta0, self.prioritized_replay_beta_iters = prioritized_replay_beta0, prioritized_replay_beta_iters
        self.prioritized_replay_eps, self.param_noise = prioritized_replay_eps, param_noise
        self.verbose, self.tensorboard_log, self.seed, self.device = verbose, tensorboard_log, seed, device
        self.n_cpu_tf_sess, self.full_tensorboard_log = n_cpu_tf_sess, full_tensorboard_log
        self.num_timesteps, self.n_updates = 0, 0
        self.env = self.observation_space = self.action_space = None
        self.n_envs, self._vec_normalize_env, self.engine = 1, None, None
        self._rng = np.random.default_rng(seed)
        if self._dp_runtime() is not None:        # exploration differs per replica
            self._rng = np.random.default_rng([0 if seed is None else int(seed), self._dp_rt.rank])
        self.exploration = None
        if env is not None:
            self.set_env(env)
        if _init_setup_model and self.observation_space is not None:
            self.setup_model()

    # ------------------------------------------------------------------ env
    def set_env(self, env):
        if env is not None and not isinstance(env, VecEnv) and not hasattr(env, "num_envs"):
            env = DummyVecEnv([lambda: env])
        if env is not None:
            if env.num_envs != 1:
                raise ValueError("%s trains on a single environment (as stable-baselines does)" % type(self).__name__)
            import os
            if os.environ.get("GRL_NUM_ENVS", "1").strip() not in ("", "1"):
                logger.warn("GRL_NUM_ENVS is ignored by %s: it trains on the single environment it is given, like "
                            "stable-baselines' (the fan-out is SAC's)" % type(self).__name__)
            self.observation_space, self.action_space = env.observation_space, env.action_space
            self._vec_normalize_env = unwrap_vec_normalize(env)
        self.env = env

    def get_env(self):
        return self.env

    def get_vec_normalize_env(self):
        return self._vec_normalize_env

    # ------------------------------------------------------------------ data parallel
    def _dp_runtime(self):
        if self._dp_rt is not None:
            return self._dp_rt
        if self.data_parallel in (None, False, "", "0", "off"):
            return None
        from ..parallel import runtime_for
        self._dp_rt = runtime_for(self.data_parallel)
        return self._dp_rt

    # ------------------------------------------------------------------ model
    def _towers(self):
        raise NotImplementedError

    def setup_model(self):
        if getattr(self.policy, "layer_norm", False) or self.policy_kwargs.get("layer_norm", False):
            raise NotImplementedError("layer_norm policies are not implemented in the HIP engine")
        if getattr(self.policy, "feature_extraction", "mlp") == "cnn":
            raise NotImplementedError("%s on image observations is not implemented (the reference trains it on "
                                      "auto-encoder features)" % type(self).__name__)
        obs_shape = tuple(self.observation_space.shape)
        if len(obs_shape) != 1:
            raise ValueError("vector observations expected, got %s" % (obs_shape,))
        D, bins, common, branch, value = self._towers()
        vn = self._vec_normalize_env
        kw = {}
        if vn is not None:
            kw = dict(clip_obs=vn.clip_obs, clip_reward=vn.clip_reward, norm_eps=vn.epsilon)
        lr = float(self.learning_rate(1.0)) if callable(self.learning_rate) else float(self.learning_rate)
        rt = self._dp_runtime()
        self._local_batch = self.batch_size if rt is None else rt.shard(self.batch_size, "minibatch rows")
        engine_seed = 0 if self.seed is None else int(self.seed)
        if rt is not None:
            engine_seed += 7919 * rt.rank             # every replica draws its own replay indices
            self.device = rt.device
        cfg = _capi.make_q_config(self.algo, obs_shape[0], D, bins, common, branch, value, batch_size=self._local_batch,
                                  act_batch=1, replay_capacity=self.buffer_size, normalize=0 if vn is None else _capi.norm_mode(vn), gamma=self.gamma,
                                  lr=lr, double_q=self.double_q, seed=engine_seed,
                                  prioritized=bool(self.prioritized_replay), per_alpha=self.prioritized_replay_alpha,
                                  per_eps=self.prioritized_replay_eps, **kw)
        self.engine = self._engine_factory(cfg, self.device)
        self.D, self.bins = D, bins
        self._init_weights()
        if rt is not None:
            self._dp = rt.make_exchange(self.engine, prefer=self.dp_exchange)
            if self.prioritized_replay and not hasattr(self._dp, "train_per"):
                raise NotImplementedError("prioritized_replay under data parallelism needs the in-graph exchange (every rank draws "
                                          "from its own priority tree inside the update's graph); this job fell back to a collective library")
            self._dp.broadcast_parameters(src=0)
        # prioritised replay lives on the device (csrc/per_kernels.h): sampling, importance weights and the
        # priority write-back never leave HBM
        self._max_priority = 1.0

    def _init_weights(self):
        """tf.contrib.layers.fully_connected defaults: Xavier-uniform weights, zero biases; target = copy."""
        rng = np.random.default_rng(0 if self.seed is None else int(self.seed))
        P = {}
        for name, _, _, shape, _ in self.engine.table:
            if "/target_q_func/" in name:
                continue
            if name.endswith("weights:0"):
                lim = np.sqrt(6.0 / (shape[0] + shape[1]))
                P[name] = rng.uniform(-lim, lim, shape).astype(np.float32)
            elif name.endswith("eps:0"):
                P[name] = np.float32(self.exploration_initial_eps).reshape(())
            else:
                P[name] = np.zeros(shape, np.float32)
        for name, *_ in self.engine.table:
            if "/target_q_func/" in name:
                P[name] = P[name.replace("/target_q_func", "")].copy()
        self.engine.set_parameters(P)

    def _eps_name(self):
        return ("deepq" if self.algo == "dqn" else "bdq") + "/eps:0"

    # ------------------------------------------------------------------ acting
    def _greedy_bins(self, obs):
        return self.engine.q_values(np.asarray(obs, np.float32).reshape(1, -1)).argmax(axis=2)[0]     # [D]

    def _bins_to_env_action(self, bins):
        raise NotImplementedError

    def predict(self, observation, state=None, mask=None, deterministic=True):
        observation = np.asarray(observation)
        single = observation.shape == tuple(self.observation_space.shape)
        obs = observation.reshape((-1,) + tuple(self.observation_space.shape))
        eps = 0.0 if deterministic else float(self.engine.get_parameters()[self._eps_name()])
        acts = []
        for o in obs:
            b = self._greedy_bins(o)
            if eps > 0 and self._rng.random() < eps:
                b = self._rng.integers(0, self.bins, self.D)
            acts.append(self._bins_to_env_action(b))
        acts = np.asarray(acts)
        return (acts[0] if single else acts), None

    # ------------------------------------------------------------------ learn
    def learn(self, total_timesteps, callback=None, log_interval=100, tb_log_name=None, reset_num_timesteps=True,
              replay_wrapper=None):
        if self.env is None:
            raise ValueError("learn() needs an environment")
        total_timesteps = int(total_timesteps)
        if reset_num_timesteps:
            self.num_timesteps = 0
        rt, dp = self._dp_rt, self._dp
        W = 1 if rt is None else rt.world
        lead = rt is None or rt.rank == 0
        callback = as_callback(callback if lead else None)      # data parallel: evaluation / checkpoints / logging on rank 0 only
        callback.init_callback(self)
        eng, vn = self.engine, self._vec_normalize_env
        if rt is not None and vn is not None:       # running statistics merged over the ranks
            from ..parallel import share_running_stats
            share_running_stats(vn, rt.ctrl)
        self.exploration = LinearSchedule(self.exploration_fraction * total_timesteps, self.exploration_final_eps,
                                          self.exploration_initial_eps)
        beta_iters = self.prioritized_replay_beta_iters or total_timesteps
        beta_schedule = LinearSchedule(beta_iters, 1.0, self.prioritized_replay_beta0)
        episode_rewards, episode_successes = [0.0], []
        writer = logger.SummaryWriter(self.tensorboard_log, tb_log_name or type(self).__name__) \
            if getattr(self, "tensorboard_log", None) else None
        obs = self.env.reset()
        obs_ = vn.get_original_obs() if vn is not None else obs
        start = time.time()
        callback.on_training_start(locals(), globals())
        callback.on_rollout_start()
        finished = False
        try:
            self._learn_loop(total_timesteps, callback, log_interval, rt, dp, W, lead, eng, vn, obs, obs_, beta_schedule,
                             episode_rewards, episode_successes, start)
            finished = True
        finally:
            if rt is not None and vn is not None:          # the collective hook must not outlive the collective loop
                for rms in (vn.obs_rms, vn.ret_rms):
                    rms.__dict__.pop("gather", None)
        if dp is not None and hasattr(dp, "check"):
            dp.check()                  # raises if an exchange timed out (replicas out of step)
        if dp is not None and finished and hasattr(dp, "close"):
            dp.close()                  # collective drain: no rank releases its exchange memory while a peer still pulls from it
        P = {self._eps_name(): np.float32(self.exploration.value(self.num_timesteps)).reshape(())}
        eng.set_parameters(P, exact_match=False)        # stable-baselines stores the last epsilon with the model
        callback.on_training_end()
        if writer is not None:
            writer.close()
        return self

    def _learn_loop(self, total_timesteps, callback, log_interval, rt, dp, W, lead, eng, vn, obs, obs_, beta_schedule,
                    episode_rewards, episode_successes, start):
        # stable-baselines: `for _ in range(total_timesteps)` -- a continued run (reset_num_timesteps=False) takes total_timesteps
        # MORE steps from where the counter stands; the schedules keep reading the counter itself
        end = self.num_timesteps + total_timesteps
        while self.num_timesteps < end:
            eps = self.exploration.value(self.num_timesteps)
            if self._rng.random() < eps:
                bins = self._rng.integers(0, self.bins, self.D)
            else:
                bins = self._greedy_bins(obs[0])
            env_action = self._bins_to_env_action(bins)
            new_obs, rew, done, info = self.env.step(np.asarray([env_action]))
            before = self.num_timesteps
            self.num_timesteps += W                 # environment steps of the job
            callback.update_locals(locals())
            stop = callback.on_step() is False
            if rt is not None:
                stop = rt.any(stop)                 # rank 0's callbacks decide for every replica; no rank runs ahead
            if stop:
                break
            new_obs_, rew_ = (vn.get_original_obs(), vn.get_original_reward()) if vn is not None else (new_obs, rew)
            eng.replay_add(np.asarray(obs_, np.float32), bins.astype(np.float32).reshape(1, -1),
                           np.asarray(rew_, np.float32), np.asarray(new_obs_, np.float32), np.asarray(done, np.float32))
            obs, obs_ = new_obs, new_obs_
            episode_rewards[-1] += float(np.asarray(rew_).reshape(-1)[0])
            if done[0]:
                if isinstance(info[0], dict) and info[0].get("is_success") is not None:
                    episode_successes.append(float(info[0]["is_success"]))
                episode_rewards.append(0.0)
            can_sample = eng.replay_size() >= self._local_batch
            # one update per train_freq environment steps of the job: W of them happened in this iteration
            crossed = lambda every: self.num_timesteps // max(1, int(every)) - before // max(1, int(every))
            n_upd = crossed(self.train_freq) if (can_sample and self.num_timesteps > self.learning_starts) else 0
            if n_upd > 0:
                callback.on_rollout_end()
                if vn is not None and eng.cfg.normalize:
                    eng.set_obs_stats(vn.obs_rms.mean, vn.obs_rms.var, float(vn.ret_rms.var))
                if callable(self.learning_rate):    # stable-baselines evaluates the schedule per update: lr(1 - step / total)
                    eng.set_learning_rate(self.learning_rate(1.0 - (self.num_timesteps - 1) / max(1, total_timesteps)))
                if self.prioritized_replay:
                    (eng if dp is None else dp).train_per(n_upd, beta_schedule.value(self.num_timesteps))
                elif dp is not None:
                    dp.train(n_upd)                # the global minibatch: gradients exchanged inside each update's graph
                else:
                    eng.train(n_upd)               # uniform indices from the device RNG, importance weights 1
                self.n_updates += n_upd
                callback.on_rollout_start()
            if can_sample and self.num_timesteps > self.learning_starts and crossed(self.target_network_update_freq) > 0:
                eng.update_target()
            if self.verbose >= 1 and lead and done[0] and log_interval is not None and len(episode_rewards) % log_interval == 0:
                logger.logkv("steps", self.num_timesteps)
                logger.logkv("episodes", len(episode_rewards))
                logger.logkv("mean 100 episode reward", round(float(np.mean(episode_rewards[-101:-1])), 1))
                logger.logkv("% time spent exploring", int(100 * eps))
                if episode_successes:
                    logger.logkv("success rate", float(np.mean(episode_successes[-100:])))
                logger.logkv("fps", int(self.num_timesteps / (time.time() - start + 1e-9)))
                logger.dumpkvs()

    # ------------------------------------------------------------------ persistence
    def get_parameter_list(self):
        return self.engine.param_names()

    def get_parameters(self):
        return self.engine.get_parameters()

    def load_parameters(self, load_path_or_dict, exact_match=True):
        params = load_path_or_dict
        if isinstance(params, str):
            _, params = save_util.load_from_zip(params)
        unknown = [k for k in params if k not in set(self.get_parameter_list())]
        if unknown:
            raise RuntimeError("load_parameters: unknown parameter(s) %s" % unknown[:5])
        self.engine.set_parameters(params, exact_match=exact_match)

    _SAVED = ("double_q", "param_noise", "learning_starts", "train_freq", "prioritized_replay",
              "prioritized_replay_eps", "batch_size", "target_network_update_freq", "prioritized_replay_alpha",
              "prioritized_replay_beta0", "prioritized_replay_beta_iters", "exploration_final_eps",
              "exploration_fraction", "gamma", "verbose", "buffer_size", "n_cpu_tf_sess", "seed", "policy_kwargs")

    def _data(self):
        d = {k: getattr(self, k) for k in self._SAVED}
        d["learning_rate"] = float(self.learning_rate(1.0)) if callable(self.learning_rate) else float(self.learning_rate)
        d.update(observation_space=self.observation_space, action_space=self.action_space, policy=self.policy,
                 n_envs=1, _vectorize_action=True)
        return d

    def save(self, save_path, cloudpickle=False):
        return save_util.save_to_zip(save_path, self._data(), self.get_parameters())

    @classmethod
    def load(cls, load_path, env=None, custom_objects=None, **kwargs):
        data, params = save_util.load_from_zip(load_path)
        model = cls(policy=data.get("policy") if isinstance(data.get("policy"), type) else None, env=None,
                    _init_setup_model=False)
        for k, v in data.items():
            if k in cls._SAVED + ("learning_rate", "num_actions_pad", "epsilon_greedy") and v is not None:
                setattr(model, k, v)
        model.policy_kwargs = dict(model.policy_kwargs or {})
        for k, v in kwargs.items():
            setattr(model, k, v)
        model.observation_space, model.action_space = data.get("observation_space"), data.get("action_space")
        if env is not None:
            model.set_env(env)
        if model.observation_space is None or model.action_space is None:
            raise ValueError("the zip holds no readable spaces; pass env=")
        model._infer_shapes(params)
        model.setup_model()
        model.load_parameters(params)
        return model

    def _infer_shapes(self, params):
        pass


class DQN(_QModel):
    """Stock stable-baselines deepq: two separate dueling towers on the observation."""
    algo = "dqn"

    def _towers(self):
        if not sp.is_discrete(self.action_space):
            raise ValueError("DQN needs a Discrete action space")
        if not self.policy_kwargs.get("dueling", True):
            raise NotImplementedError("only the dueling architecture (stable-baselines default) is implemented")
        layers = tuple(self.policy_kwargs.get("layers", (64, 64)))
        return 1, int(self.action_space.n), (), layers, layers

    def _bins_to_env_action(self, bins):
        return int(bins[0])


class BDQ(_QModel):
    """Branching dueling Q-network of the reference's ``bdq_sb`` fork (decisions: SURVEY.md A.6)."""
    algo = "bdq"

    def __init__(self, policy, env, num_actions_pad=33, epsilon_greedy=True, learning_rate=1e-4, batch_size=64,
                 buffer_size=1000000, target_network_update_freq=1000, exploration_fraction=0.1,
                 exploration_final_eps=0.02, **kwargs):
        self.num_actions_pad, self.epsilon_greedy = int(num_actions_pad), epsilon_greedy
        if not epsilon_greedy:
            raise NotImplementedError("only epsilon-greedy exploration is implemented for BDQ")
        super().__init__(policy, env, learning_rate=learning_rate, batch_size=batch_size, buffer_size=buffer_size,
                         target_network_update_freq=target_network_update_freq,
                         exploration_fraction=exploration_fraction, exploration_final_eps=exploration_final_eps, **kwargs)

    _SAVED = _QModel._SAVED + ("num_actions_pad", "epsilon_greedy")

    def _towers(self):
        if not sp.is_box(self.action_space):
            raise ValueError("BDQ discretises a continuous (Box) action space")
        layers = self.policy_kwargs.get("layers", [[512, 256], [128], [128]])
        if len(layers) != 3:
            raise ValueError("policy_kwargs['layers'] must be [[common...], [branch...], [value...]]")
        return int(np.prod(self.action_space.shape)), self.num_actions_pad, tuple(layers[0]), tuple(layers[1]), tuple(layers[2])

    def _bins_to_env_action(self, bins):
        low, high = np.asarray(self.action_space.low, np.float32), np.asarray(self.action_space.high, np.float32)
        unit = -1.0 + 2.0 * np.asarray(bins, np.float32) / float(self.bins - 1)       # bin k -> [-1, 1]
        return (low + 0.5 * (unit + 1.0) * (high - low)).reshape(self.action_space.shape)
