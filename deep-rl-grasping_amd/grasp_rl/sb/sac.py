"""``SAC`` with the stable-baselines 2.10 constructor / learn / predict / save / load surface that
/root/reference/manipulation_main/training/sb_helper.py:104-128,175-177,228-247,
train_stable_baselines.py:96-104, base_callbacks.py:78-117,139-149 and utils.py:65-76 exercise.
The update itself (replay sampling + normalisation, CNN/MLP forward-backward, losses, Adam, Polyak)
runs in the HIP engine (grasp_rl.engine.SacEngine -> libgrl.so); this file is the host loop that
stable-baselines' ``SAC.learn`` implements: act -> env.step -> callbacks -> store
raw transition -> every ``train_freq`` steps ``gradient_steps`` updates.

Differences to stable-baselines, all opt-in: vectorised envs with ``num_envs > 1`` are accepted
(stable-baselines 2 asserts a single env); ``ent_coef`` must be 'auto' / 'auto_<init>' and
``target_update_interval`` 1 (what the reference uses: zip JSON).

Data parallelism (BASELINE configs[4]: global batch 1024, 64 envs, 8 GPUs).  ``data_parallel="auto"`` (or
GRL_DATA_PARALLEL=auto in the environment, so that the reference's unmodified scripts pick it up) makes a model that was
started by ``torch.distributed.run`` one replica of W: ``batch_size`` is the GLOBAL minibatch (each rank's engine takes
batch_size / W rows from ITS replay shard, which holds the transitions of ITS environments), the gradients are exchanged
inside the update's graph (``grasp_rl.parallel.DataParallelInGraph``; RCCL as fallback), parameters start from rank 0's,
the VecNormalize statistics are merged over the ranks at every env step (host: ``share_running_stats``; with
``device_norm`` on the device, grl_norm_update), ``num_timesteps`` / ``total_timesteps`` / ``learning_starts`` count the
environment steps of ALL ranks, callbacks (evaluation, checkpoints, logging) run on rank 0 only and its stop request ends
``learn`` on every rank.  ``python -m grasp_rl.dp_run <script> ...`` is the launcher-side helper for scripts that create
directories (train_stable_baselines.py:27-29).

Updates per env step.  stable-baselines runs ``gradient_steps`` updates every ``train_freq`` calls of ``env.step``
and its env is a single environment, so its defaults (1, 1) mean ONE UPDATE PER ENVIRONMENT STEP.  With N
sub-environments one ``env.step`` is N environment steps; ``gradient_steps=None`` (the default here) resolves to
``N * train_freq // train_freq = N`` updates per vectorised step and so keeps that ratio (for N = 1 it is
stable-baselines' 1).  An explicit integer is taken literally (``gradient_steps=1`` with 16 envs = one update per
16 environment steps).  The k updates of one loop iteration are ONE ``grl_train_step(k)`` call.
"""
import os
import time
from collections import deque

import numpy as np

from .. import _capi
from ..engine import SacEngine
from ..init import init_parameters
from . import logger
from . import policies as pol
from . import save_util
from . import spaces as sp
from .callbacks import as_callback, may_read_observations
from .vec_env import DummyVecEnv, VecEnv, expand_training_env, unwrap_vec_normalize

_POLICY_NAMES = {"MlpPolicy": pol.SacMlpPolicy, "CnnPolicy": pol.SacCnnPolicy, "LnMlpPolicy": pol.SacLnMlpPolicy,
                 "LnCnnPolicy": pol.SacLnCnnPolicy}


def resolve_device_norm(setting, user_callback, rt, dp):
    """Where the VecNormalize statistics live during ``learn``: True = on the device (``grl_observe`` / ``grl_norm_update``).
    'auto' looks at the callback the USER handed over -- before ``learn`` replaces it by a bare one on ranks > 0 -- and under
    data parallelism the decision is COLLECTIVE (``rt.all``): with the statistics on the device every env step runs an
    in-kernel merge that waits for all peers, with them on the host an all_gather on the control group; replicas that chose
    differently (a script that hands only rank 0 a callback) would wait for each other in two different exchanges."""
    if setting == "auto":
        setting = not (user_callback is not None and may_read_observations(as_callback(user_callback)))
    setting = bool(setting)
    if rt is not None:
        if setting and not hasattr(dp, "check"):
            # the device-side merge of the statistics rides on the in-graph exchange's channel; with the collective
            # fallback (RCCL / gloo) the handle is not connected and every replica would keep its own observation
            # statistics: take the host path, whose moments are gathered over the ranks
            logger.warn("device_norm needs the in-graph exchange; this job fell back to a collective library -> host statistics")
            setting = False
        setting = rt.all(setting)
    return setting


class SAC:
    # tests substitute the g++ emulation build here; the product default needs a HIP device
    _engine_factory = staticmethod(lambda cfg, device: SacEngine(cfg, device=device))

    def __init__(self, policy, env, gamma=0.99, learning_rate=3e-4, buffer_size=50000, learning_starts=100,
                 train_freq=1, batch_size=64, tau=0.005, ent_coef="auto", target_update_interval=1,
                 gradient_steps=None, target_entropy="auto", action_noise=None, random_exploration=0.0, verbose=0,
                 tensorboard_log=None, _init_setup_model=True, policy_kwargs=None, full_tensorboard_log=False,
                 seed=None, n_cpu_tf_sess=None, device="cuda:0", overlap_env_step=None, replay_rgb_u8=False,
                 device_norm=None, data_parallel=None, dp_exchange="ingraph", dp_mode="auto", dp_overlap=False):
        if isinstance(policy, str):
            if policy not in _POLICY_NAMES:
                raise ValueError("unknown policy %r" % policy)
            policy = _POLICY_NAMES[policy]
        self.policy = policy
        self.policy_kwargs = {} if policy_kwargs is None else dict(policy_kwargs)
        self.gamma, self.learning_rate = gamma, learning_rate
        self.buffer_size, self.learning_starts = int(buffer_size), learning_starts
        self.train_freq, self.batch_size, self.tau = train_freq, int(batch_size), tau
        self.ent_coef, self.target_update_interval = ent_coef, target_update_interval
        self.gradient_steps, self.target_entropy = gradient_steps, target_entropy
        self.action_noise, self.random_exploration = action_noise, random_exploration
        self.verbose, self.tensorboard_log = verbose, tensorboard_log
        self.full_tensorboard_log, self.seed, self.n_cpu_tf_sess = full_tensorboard_log, seed, n_cpu_tf_sess
        self.device = device
        # Opt-in pipelining (not stable-baselines semantics): the gradient update of step t is launched while
        # the (subprocess) environments compute step t, i.e. it samples a replay that does not yet hold
        # transition t.  Off by default: the default loop is SB's strict step -> store -> update order.
        if overlap_env_step is None:
            overlap_env_step = os.environ.get("GRL_OVERLAP_ENV_STEP", "0") == "1"
        self.overlap_env_step = bool(overlap_env_step)
        # Opt-in: RGB-D observations whose colour channels are integers in [0, 255] (the reference's camera: uint8,
        # sensor.py:126-145) are stored with byte colours -- half the HBM per transition (grl_config.replay_rgb_u8).
        self.replay_rgb_u8 = bool(replay_rgb_u8)
        # The VecNormalize observation statistics on the device (grl_norm_update / grl_observe) while learning: the wrapper
        # then hands RAW observations to this loop (callbacks would see them as `new_obs`); actions and minibatches are
        # normalised where the engine consumes them, with the same arithmetic -- parameters and the pickled statistics are
        # bit-identical to the host path (tests/test_sb_api_host.py), one upload per env step instead of three, no float64
        # pass over the observations on the host (learn loop 80 % -> 94 % of the engine-only rate in strict order).  The
        # wrapper's host copy is refreshed whenever it is pickled / saved / synchronised into an evaluation env and when
        # learn() returns.  True / False, or "auto" (default; GRL_DEVICE_NORM=0 / 1 / auto): on unless a callback may look at
        # the observations (callbacks.may_read_observations: everything but this package's and the reference's own callbacks).
        if device_norm is None:
            device_norm = {"0": False, "1": True}.get(os.environ.get("GRL_DEVICE_NORM", "auto"), "auto")
        self.device_norm = device_norm if device_norm == "auto" else bool(device_norm)
        # Opt-in: one replica of a data-parallel job (module docstring).  None / False: off; "auto": when this process was
        # launched by torch.distributed.run with WORLD_SIZE > 1; True: required (raises without a launch).
        if data_parallel is None:
            data_parallel = os.environ.get("GRL_DATA_PARALLEL") or None
        self.data_parallel = data_parallel
        self.dp_exchange, self.dp_mode, self.dp_overlap = dp_exchange, dp_mode, bool(dp_overlap)
        self._dp_rt = self._dp = None
        self.num_timesteps = 0
        self.n_updates = 0
        self.env = None
        self.observation_space = self.action_space = None
        self.n_envs = 1
        self._vectorize_action = True
        self._vec_normalize_env = None
        self.engine = None
        self.episode_reward = None
        self._rng = np.random.default_rng(seed)
        if self._dp_runtime() is not None:        # exploration noise differs per replica
            self._rng = np.random.default_rng([0 if seed is None else int(seed), self._dp_rt.rank])
        if env is not None:
            self.set_env(env)
        if _init_setup_model and self.observation_space is not None:
            self.setup_model()

    # ------------------------------------------------------------------ environment
    def set_env(self, env):
        if env is not None and not isinstance(env, VecEnv) and not hasattr(env, "num_envs"):
            env = DummyVecEnv([lambda: env])
        if env is not None:
            # GRL_NUM_ENVS: the env a model is handed is its TRAINING env -- the one the reference's script wraps as
            # DummyVecEnv([one factory]) (train_stable_baselines.py:52-54) fans out to worker processes here (row J3)
            rt = self._dp_runtime()
            env = expand_training_env(env, rank=0 if rt is None else rt.rank, world=1 if rt is None else rt.world)
            if self.observation_space is not None and tuple(env.observation_space.shape) != tuple(self.observation_space.shape):
                raise ValueError("observation space of the new env does not match the model")
            self.observation_space, self.action_space = env.observation_space, env.action_space
            self.n_envs = env.num_envs
            self._vec_normalize_env = unwrap_vec_normalize(env)
            if self.engine is not None and self.n_envs > self.engine.cfg.act_batch:
                raise ValueError("env has more sub-environments than the model was built for")
            self._norm_stamp = None     # a new wrapper: its statistics must reach the device whatever their counts are
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
    def _learning_rate_value(self):
        lr = self.learning_rate
        return float(lr(1.0)) if callable(lr) else float(lr)

    def setup_model(self):
        if not sp.is_box(self.action_space):
            raise ValueError("SAC needs a continuous (Box) action space")
        if getattr(self.policy, "layer_norm", False) or self.policy_kwargs.get("layer_norm", False):
            raise NotImplementedError("layer_norm policies are not implemented in the HIP engine")
        if self.target_update_interval != 1:
            raise NotImplementedError("target_update_interval must be 1 (Polyak update fused into the Adam kernel)")
        ent = self.ent_coef
        if not (isinstance(ent, str) and ent.startswith("auto")):
            raise NotImplementedError("only ent_coef='auto' / 'auto_<init>' is implemented")
        self._ent_init = float(ent.split("_")[1]) if "_" in ent else 1.0
        extractor, n_direct = pol.extractor_from_kwargs(self.policy, self.policy_kwargs)
        layers = tuple(self.policy_kwargs.get("layers", (64, 64)))
        obs_shape = tuple(self.observation_space.shape)
        act_dim = int(np.prod(self.action_space.shape))
        if self.target_entropy == "auto":
            self.target_entropy = -np.prod(self.action_space.shape).astype(np.float32)
        rt = self._dp_runtime()
        self._local_batch = self.batch_size if rt is None else rt.shard(self.batch_size, "minibatch rows")
        engine_seed = 0 if self.seed is None else int(self.seed)
        if rt is not None:
            engine_seed = engine_seed + 7919 * rt.rank       # every replica draws its own replay indices / policy noise
            self.device = rt.device
        kw = dict(act_dim=act_dim, layers=layers, batch_size=self._local_batch, act_batch=max(1, self.n_envs),
                  replay_capacity=self.buffer_size,
                  normalize=0 if self._vec_normalize_env is None else _capi.norm_mode(self._vec_normalize_env),
                  gamma=self.gamma, lr=self._learning_rate_value(), tau=self.tau,
                  target_entropy=float(self.target_entropy), seed=engine_seed)
        if self._vec_normalize_env is not None:
            vn = self._vec_normalize_env
            kw.update(clip_obs=vn.clip_obs, clip_reward=vn.clip_reward, norm_eps=vn.epsilon)
        if extractor == "mlp":
            if len(obs_shape) != 1:
                raise ValueError("MlpPolicy expects vector observations, got shape %s" % (obs_shape,))
            cfg = _capi.make_config("mlp", obs_dim=obs_shape[0], **kw)
        else:
            if len(obs_shape) != 3 or obs_shape[0] != 64 or obs_shape[1] != 64:
                raise ValueError("CnnPolicy expects 64x64xC image observations, got %s" % (obs_shape,))
            if not sp.has_finite_bounds(self.observation_space) or np.any(self.observation_space.low != 0) \
                    or np.any(self.observation_space.high != 255):
                raise NotImplementedError("image observation spaces other than Box(0, 255) are not implemented")
            if self.replay_rgb_u8 and obs_shape[2] - (1 if n_direct > 0 else 0) != 4:
                raise ValueError("replay_rgb_u8 needs RGB-D observations (R, G, B, depth [+ pad channel])")
            cfg = _capi.make_config(extractor, obs_channels=obs_shape[2], n_direct=n_direct,
                                    replay_rgb_u8=self.replay_rgb_u8, **kw)
        self._extractor = extractor
        if self.seed is not None and hasattr(self.action_space, "seed"):     # BaseRLModel.set_random_seed: action_space.seed(seed)
            self.action_space.seed(int(self.seed))
        self._norm_stamp = None         # a new engine starts with zero statistics
        self.engine = self._engine_factory(cfg, self.device)
        params = init_parameters(self.engine.table, seed=0 if self.seed is None else int(self.seed))
        params["model/log_ent_coef:0"] = np.float32(np.log(self._ent_init)).reshape(())
        self.engine.set_parameters(params)
        if rt is not None:
            self._dp = rt.make_exchange(self.engine, prefer=self.dp_exchange, overlap=self.dp_overlap, mode=self.dp_mode)
            self._dp.broadcast_parameters(src=0)

    def _sync_norm_stats(self):
        """VecNormalize statistics are updated on the env side every step and read at sample time
        with their *current* values: push them to the device before updating."""
        vn = self._vec_normalize_env
        if vn is None or not self.engine.cfg.normalize:
            return
        if vn.hands_out_raw_observations:        # observation statistics live on the device: only the return variance moves
            stamp = ("dev", float(vn.ret_rms.count))
            if stamp != getattr(self, "_norm_stamp", None):
                self._norm_stamp = stamp
                self.engine.set_ret_var(float(vn.ret_rms.var))
            return
        # (identity + counts + a content checksum: in-place edits of the arrays and re-loaded pickles are seen too)
        stamp = (id(vn.obs_rms), float(vn.obs_rms.count), id(vn.ret_rms), float(vn.ret_rms.count),
                 float(np.sum(vn.obs_rms.mean)), float(np.sum(vn.obs_rms.var)), float(np.sum(vn.ret_rms.var)))
        if stamp == getattr(self, "_norm_stamp", None):
            return                      # frozen wrapper (training=False) or no env step since the last push
        self._norm_stamp = stamp
        self.engine.set_obs_stats(vn.obs_rms.mean, vn.obs_rms.var, float(vn.ret_rms.var))   # asynchronous, stream-ordered

    # ------------------------------------------------------------------ acting
    def _act(self, obs, deterministic, raw=False, observed=False):
        n = len(obs)
        if observed:      # the env wrapper uploaded these observations already (engine.observe): only noise and actions move
            eps = None if deterministic else self._rng.standard_normal((n, int(np.prod(self.action_space.shape)))).astype(np.float32)
            return self.engine.act(n, deterministic, eps, raw=raw, observed=True)
        obs = np.asarray(obs, np.float32)
        out = np.empty((n, int(np.prod(self.action_space.shape))), np.float32)
        cap = self.engine.cfg.act_batch
        for k0 in range(0, n, cap):
            chunk = obs[k0:k0 + cap]
            eps = None if deterministic else self._rng.standard_normal((chunk.shape[0], out.shape[1])).astype(np.float32)
            out[k0:k0 + cap] = self.engine.act(chunk, deterministic, eps, raw=raw) if raw else self.engine.act(chunk, deterministic, eps)
        return out

    def _unscale(self, a):
        low, high = np.asarray(self.action_space.low, np.float32), np.asarray(self.action_space.high, np.float32)
        return low + 0.5 * (a + 1.0) * (high - low)

    def _scale(self, a):
        low, high = np.asarray(self.action_space.low, np.float32), np.asarray(self.action_space.high, np.float32)
        return 2.0 * (a - low) / (high - low) - 1.0

    def predict(self, observation, state=None, mask=None, deterministic=True):
        observation = np.asarray(observation)
        single = observation.shape == tuple(self.observation_space.shape)
        obs = observation.reshape((-1,) + tuple(self.observation_space.shape))
        act = self._unscale(self._act(obs, deterministic)).reshape((-1,) + tuple(self.action_space.shape))
        act = np.clip(act, self.action_space.low, self.action_space.high)
        return (act[0] if single else act), None

    # ------------------------------------------------------------------ learn
    def learn(self, total_timesteps, callback=None, log_interval=4, tb_log_name="SAC", reset_num_timesteps=True,
              replay_wrapper=None):
        if self.env is None:
            raise ValueError("learn() needs an environment: pass env to the constructor / load() or call set_env")
        total_timesteps = int(total_timesteps)
        if reset_num_timesteps:
            self.num_timesteps = 0
        rt, dp = self._dp_rt, self._dp
        W = 1 if rt is None else rt.world
        lead = rt is None or rt.rank == 0
        device_norm = resolve_device_norm(self.device_norm, callback, rt, dp)
        callback = as_callback(callback if lead else None)      # data parallel: evaluation / checkpoints / logging on rank 0 only
        callback.init_callback(self)
        eng, vn, N = self.engine, self._vec_normalize_env, self.n_envs
        if rt is not None and vn is not None:       # running statistics merged over the ranks (ret_rms always on the host)
            from ..parallel import share_running_stats
            share_running_stats(vn, rt.ctrl)
        episode_rewards = [0.0]
        episode_successes = []
        ep_info_buf = deque(maxlen=100)
        self.episode_reward = np.zeros((N,))
        n_episodes = 0
        infos_values = {}
        start = time.time()
        # stable-baselines hands callbacks a FileWriter when tensorboard_log is set, None otherwise
        writer = logger.SummaryWriter(self.tensorboard_log, tb_log_name) if self.tensorboard_log else None
        if device_norm and vn is not None and vn.norm_obs and eng.cfg.normalize in (1, 2):
            vn.attach_device(eng)
            self._norm_stamp = None
        finished = False
        try:
            self._learn_loop(total_timesteps, callback, log_interval, writer, rt, dp, W, lead, eng, vn, N, episode_rewards,
                             episode_successes, ep_info_buf, start)
            finished = True
        finally:
            if vn is not None and vn._dev is not None:     # whatever ended the loop: the wrapper carries the statistics again
                vn.detach_device()
                self._norm_stamp = None
            if rt is not None and vn is not None:          # the collective hook must not outlive the collective loop: a later
                for rms in (vn.obs_rms, vn.ret_rms):       # step of this env on ONE rank (the script's own evaluation run on
                    rms.__dict__.pop("gather", None)       # rank 0) would wait in all_gather for peers that have left
            if writer is not None:
                writer.close()
        if dp is not None and hasattr(dp, "check"):
            dp.check()                  # raises if an exchange timed out (replicas out of step)
        if dp is not None and finished and hasattr(dp, "close"):
            # collective (every rank left the loop in the same iteration): drain + barrier, so that no rank goes on to
            # release its engine -- and with it the exchange memory its peers' last kernels pull from -- early
            dp.close()
        return self

    def _learn_loop(self, total_timesteps, callback, log_interval, writer, rt, dp, W, lead, eng, vn, N, episode_rewards,
                    episode_successes, ep_info_buf, start):
        n_episodes = 0
        infos_values = {}
        obs = self.env.reset()
        obs_ = vn.get_original_obs() if vn is not None else obs
        serial = vn.observed_serial if vn is not None else None     # engine.observe ticket of `obs` (device-side statistics)
        callback.on_training_start(locals(), globals())
        callback.on_rollout_start()
        step = 0
        # stable-baselines: `for step in range(total_timesteps)` -- a continued run (reset_num_timesteps=False) takes
        # total_timesteps MORE environment steps from where the counter stands
        end = self.num_timesteps + total_timesteps
        while self.num_timesteps < end:
            raw_obs = vn is not None and vn.hands_out_raw_observations     # (a callback may toggle vn.training)
            if self.num_timesteps < self.learning_starts or self._rng.random() < self.random_exploration:
                unscaled_action = np.stack([np.asarray(self.action_space.sample(), np.float32) for _ in range(N)])
                action = self._scale(unscaled_action)
            else:
                action = self._act(obs, deterministic=False, raw=raw_obs, observed=raw_obs and serial is not None)
                if self.action_noise is not None:
                    action = np.clip(action + self.action_noise(), -1, 1)
                unscaled_action = self._unscale(action)
            def run_updates(step_index=step):
                # the updates that follow vectorised env step `step_index` (0-based): N * step_index environment steps
                # precede it -- the same count in the strict and in the overlapped order
                callback.on_rollout_end()
                # None: one update per ENVIRONMENT step of the JOB -- N * W of them happen per vectorised step, and under data
                # parallelism every update consumes the global minibatch on all ranks at once
                k = N * W if self.gradient_steps is None else int(self.gradient_steps)
                done_steps = N * W * step_index
                if k > 0 and eng.replay_size() >= self._local_batch and done_steps + N * W >= self.learning_starts:
                    self.n_updates += k
                    self._sync_norm_stats()
                    if callable(self.learning_rate):    # SB: frac = 1 - step / total (step = index of this env step)
                        eng.set_learning_rate(self.learning_rate(1.0 - done_steps / max(1, total_timesteps)))
                    if dp is not None:
                        dp.train(k)      # k updates on the GLOBAL minibatch: gradients exchanged inside each update's graph
                    else:
                        eng.train(k)
                callback.on_rollout_start()

            self.env.step_async(unscaled_action.reshape((N,) + tuple(self.action_space.shape)))
            if self.overlap_env_step and (step + 1) % self.train_freq == 0:
                run_updates()          # GPU works while the simulator workers step
            new_obs, reward, done, info = self.env.step_wait()
            self.num_timesteps += N * W
            callback.update_locals(locals())
            stop = callback.on_step() is False
            if rt is not None:
                stop = rt.any(stop)      # rank 0's callbacks decide for every replica (and no rank runs ahead of the others)
            if stop:
                break
            new_serial = vn.observed_serial if vn is not None else None
            # an auto-reset env returns the first observation of the next episode: the replay row keeps the terminal one
            ended = ([i for i in range(N) if done[i] and isinstance(info[i], dict) and "terminal_observation" in info[i]]
                     if np.any(done) else [])
            if serial is not None and new_serial == serial + 1:
                # both sides of the transitions are on the device already (uploaded once each by the wrapper)
                new_obs_, reward_ = vn.old_obs, vn.old_rews
                eng.replay_add_observed(action.reshape(N, -1), reward_, done, ended,
                                        np.stack([info[i]["terminal_observation"] for i in ended]) if ended else None)
            else:
                if vn is not None:
                    new_obs_, reward_ = vn.get_original_obs(), vn.get_original_reward()
                else:
                    new_obs_, reward_ = new_obs, reward
                next_store = np.array(new_obs_, np.float32, copy=True)
                for i in ended:
                    next_store[i] = info[i]["terminal_observation"]
                eng.replay_add(np.asarray(obs_, np.float32), action.reshape(N, -1), np.asarray(reward_, np.float32),
                               next_store, np.asarray(done, np.float32))
            obs, obs_, serial = new_obs, new_obs_, new_serial
            step += 1
            if step % self.train_freq == 0 and not self.overlap_env_step:
                run_updates()          # strict order: the GPU idles from the action to this point -- bookkeeping comes after it
            for i in range(N):
                maybe = info[i].get("episode") if isinstance(info[i], dict) else None
                if maybe is not None:
                    ep_info_buf.append(maybe)
                if isinstance(info[i], dict) and info[i].get("is_success") is not None and done[i]:
                    episode_successes.append(float(info[i]["is_success"]))
            self.episode_reward += np.asarray(reward_, np.float64).reshape(N)
            if step % self.train_freq == 0 and self.n_updates > 0 and (step // self.train_freq) % 50 == 0:
                infos_values = eng.metrics()
            episode_rewards[-1] += float(np.asarray(reward_).reshape(N)[0])
            if done[0]:
                if self.action_noise is not None:
                    self.action_noise.reset()
                episode_rewards.append(0.0)
                n_episodes += 1
                if lead and self.verbose >= 1 and log_interval is not None and n_episodes % log_interval == 0:
                    fps = int(self.num_timesteps / (time.time() - start + 1e-9))
                    logger.logkv("episodes", n_episodes)
                    logger.logkv("mean 100 episode reward", round(float(np.mean(episode_rewards[-101:-1])), 1))
                    if ep_info_buf:
                        logger.logkv("ep_rewmean", float(np.mean([e["r"] for e in ep_info_buf])))
                        logger.logkv("eplenmean", float(np.mean([e["l"] for e in ep_info_buf])))
                    logger.logkv("n_updates", self.n_updates)
                    logger.logkv("current_lr", self._learning_rate_value())
                    logger.logkv("fps", fps)
                    logger.logkv("time_elapsed", int(time.time() - start))
                    if episode_successes:
                        logger.logkv("success rate", float(np.mean(episode_successes[-100:])))
                    for k, v in (infos_values or eng.metrics()).items():
                        logger.logkv(k, v)
                    logger.logkv("total timesteps", self.num_timesteps)
                    logger.dumpkvs()
        callback.on_training_end()

    # ------------------------------------------------------------------ parameters / persistence
    def get_parameter_list(self):
        return self.engine.param_names()

    def get_parameters(self):
        return self.engine.get_parameters()

    def load_parameters(self, load_path_or_dict, exact_match=True):
        params = load_path_or_dict
        if isinstance(params, str):
            _, params = save_util.load_from_zip(params)
        elif isinstance(params, (list, tuple)):
            params = dict(zip(self.get_parameter_list(), params))
        known = set(self.get_parameter_list())
        unknown = [k for k in params if k not in known]
        if unknown:
            raise RuntimeError("load_parameters: unknown parameter(s) %s" % unknown[:5])
        self.engine.set_parameters(params, exact_match=exact_match)

    def _data(self):
        return {
            "learning_rate": self._learning_rate_value(), "buffer_size": self.buffer_size,
            "learning_starts": self.learning_starts, "train_freq": self.train_freq, "batch_size": self.batch_size,
            "tau": self.tau, "ent_coef": self.ent_coef if isinstance(self.ent_coef, str) else float(self.ent_coef),
            "target_entropy": float(self.target_entropy), "gamma": self.gamma, "verbose": self.verbose,
            "observation_space": self.observation_space, "action_space": self.action_space, "policy": self.policy,
            "n_envs": self.n_envs, "n_cpu_tf_sess": self.n_cpu_tf_sess, "seed": self.seed,
            "action_noise": self.action_noise, "random_exploration": self.random_exploration,
            "_vectorize_action": self._vectorize_action, "policy_kwargs": self.policy_kwargs,
        }

    def save(self, save_path, cloudpickle=False):
        return save_util.save_to_zip(save_path, self._data(), self.get_parameters())

    @classmethod
    def load(cls, load_path, env=None, custom_objects=None, **kwargs):
        data, params = save_util.load_from_zip(load_path)
        if "policy_kwargs" in kwargs and kwargs["policy_kwargs"] != data.get("policy_kwargs"):
            raise ValueError("the specified policy kwargs do not equal the stored policy kwargs")
        policy = data.get("policy")
        inferred_policy, inferred_kwargs = pol.infer_sac_policy_kwargs(params)
        if policy is None or not isinstance(policy, type):
            policy = inferred_policy
        if not isinstance(data.get("policy_kwargs"), dict):   # e.g. a cloudpickled extractor closure that was not unpickled
            data["policy_kwargs"] = inferred_kwargs
        model = cls(policy=policy, env=None, _init_setup_model=False)
        for k in ("gamma", "buffer_size", "learning_starts", "train_freq", "batch_size", "tau", "ent_coef", "verbose",
                  "n_envs", "seed", "random_exploration", "policy_kwargs", "target_entropy"):
            if k in data and data[k] is not None:
                setattr(model, k, data[k])
        if isinstance(data.get("learning_rate"), (int, float)):
            model.learning_rate = float(data["learning_rate"])
        model.policy_kwargs = dict(model.policy_kwargs or {})
        model.target_entropy = float(np.asarray(model.target_entropy)) if model.target_entropy != "auto" else "auto"
        for k, v in kwargs.items():
            setattr(model, k, v)
        model.observation_space, model.action_space = data.get("observation_space"), data.get("action_space")
        if env is not None:
            model.set_env(env)
        if model.observation_space is None or model.action_space is None:
            raise ValueError("the zip holds no readable spaces; pass env=")
        model.setup_model()
        model.load_parameters(params)
        return model
