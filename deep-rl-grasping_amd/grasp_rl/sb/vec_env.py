"""Vectorised-environment wrappers with the stable-baselines call surface the reference uses:
``DummyVecEnv([fn])`` (train_stable_baselines.py:52-54,65-67,86), ``VecNormalize(venv, training=,
norm_obs=, norm_reward=, clip_obs=)`` + ``.save/.load`` (sb_helper.py:76-78,100-103,117-119,247;
train_stable_baselines.py:88-91), ``sync_envs_normalization`` (base_callbacks.py:82), ``get_attr``
(sb_helper.py:42-45), ``envs[0]`` (sb_helper.py:86), ``buf_infos`` (utils.py:76).

VecNormalize arithmetic follows SURVEY.md A.1 (float64 statistics, clip, epsilon 1e-8, discounted
return for the reward scale); its statistics are what the engine reads at replay-sample time.
"""
import copy
import os
import pickle

import numpy as np

from . import spaces as sp
from .running_mean_std import RunningMeanStd


class VecEnv:
    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self.venv.unwrapped if isinstance(self, VecEnvWrapper) else self


class _RankedEnvFn:
    """``fn`` called with the number of the environment it builds announced to ``grasp_rl.sb.monitor`` first, so that the
    ``Monitor(env, filename)`` inside an opaque factory (train_stable_baselines.py:54: one lambda, one file name) writes
    ``<filename>.env<k>.monitor.csv`` for k > 0 instead of N workers truncating one file (``load_results`` globs
    ``*monitor.csv``; environment 0 keeps the name a single-env run uses)."""

    def __init__(self, fn, rank):
        self.fn, self.rank = fn, int(rank)

    def __call__(self):
        from . import monitor
        from ..autoencoder import ENV_WORKER_VAR
        monitor.ENV_RANK = self.rank
        # ... and with the marker that turns the sensor's `encoders.SimpleAutoEncoder(config)` (sensor.py:190-192) into the
        # deferred form while THIS factory runs (grasp_rl.autoencoder): the parent encodes the images of all environments at once
        before = os.environ.get(ENV_WORKER_VAR)
        os.environ[ENV_WORKER_VAR] = "1"
        try:
            return self.fn()
        finally:
            monitor.ENV_RANK = 0
            if before is None:
                os.environ.pop(ENV_WORKER_VAR, None)
            else:
                os.environ[ENV_WORKER_VAR] = before


def _drop_empty_monitor_file(env):
    """Remove the monitor CSV of a (closed) template environment when it holds no episode row."""
    seen = 0
    while env is not None and seen < 16:
        fh = getattr(env, "__dict__", {}).get("file_handler")
        name = getattr(fh, "name", None)
        if isinstance(name, str) and name.endswith("monitor.csv") and os.path.isfile(name):
            try:
                with open(name) as f:
                    if len(f.read().strip().splitlines()) <= 2:
                        os.remove(name)
            except OSError:
                pass
            return
        env = getattr(env, "__dict__", {}).get("env")
        seen += 1


class DummyVecEnv(VecEnv):
    """Steps its environments sequentially in this process; resets an env automatically when its
    episode ends (the terminal observation is kept in info['terminal_observation']).

    ``fan_out(n)`` (row J3: "vectorised PyBullet envs fan out across host cores", reached from the reference's unmodified
    ``DummyVecEnv([lambda: Monitor(gym.make(...), ...)])``, train_stable_baselines.py:52-54): the ONE factory this object
    was built from is called in n worker processes (``SubprocVecEnv``) and every VecEnv call is forwarded there; the
    instance built in this process stays in ``.envs`` as a template for the static questions the reference asks before it
    constructs the model (``env.envs[0].is_simplified()`` / ``.depth_obs`` / ``.full_obs``, sb_helper.py:86-91) and is
    closed -- the simulators that are stepped live in the workers (``get_attr('history')`` etc., sb_helper.py:42-45, go
    there).  ``grasp_rl.sb.SAC`` calls it on its TRAINING env when GRL_NUM_ENVS > 1 (`expand_training_env`); evaluation
    envs are never handed to a model constructor and stay single."""

    def __init__(self, env_fns):
        self._env_fns = list(env_fns)
        self._fan = None
        self.envs = [fn() for fn in self._env_fns]
        env = self.envs[0]
        super().__init__(len(self.envs), env.observation_space, env.action_space)
        self._alloc()
        self.actions = None

    def _alloc(self):
        shape = tuple(self.observation_space.shape)
        dtype = getattr(self.observation_space, "dtype", np.float32)
        self.buf_obs = np.zeros((self.num_envs,) + shape, dtype=dtype)
        self.buf_dones = np.zeros((self.num_envs,), dtype=bool)
        self.buf_rews = np.zeros((self.num_envs,), dtype=np.float32)
        self.buf_infos = [{} for _ in range(self.num_envs)]

    # -- fan-out ----------------------------------------------------------------------------------------------------
    @property
    def fanned_out(self):
        return self._fan is not None

    def fan_out(self, n, envs_per_worker=1, start_method=None, first_rank=0, seed=None):
        """Replace the single in-process environment by n worker processes built from the same factory.  `first_rank`:
        number of this object's first environment within the job (data parallel: rank * n), used for Monitor file names
        and seeds.  Idempotent for the same n."""
        n = int(n)
        if self._fan is not None:
            if n != self.num_envs:
                raise ValueError("this DummyVecEnv already fans out to %d environments" % self.num_envs)
            return self
        if len(self._env_fns) != 1:
            raise ValueError("fan_out multiplies ONE environment factory; this DummyVecEnv was built from %d" % len(self._env_fns))
        if n < 1:
            raise ValueError("fan_out needs n >= 1")
        for env in self.envs:                    # the template's Monitor file / simulator: environment 0 of the workers owns them now
            if hasattr(env, "close"):
                try:
                    env.close()
                except Exception:                # noqa: BLE001  (a simulator that cannot close twice must not stop training)
                    pass
            if first_rank > 0:                   # ... and where this process owns no environment 0 (data parallel, ranks > 0)
                _drop_empty_monitor_file(env)    # the template's header-only log_file.monitor.csv is nobody's log
        fns = [_RankedEnvFn(self._env_fns[0], first_rank + k) for k in range(n)]
        self._fan = SubprocVecEnv(fns, start_method=start_method, envs_per_worker=envs_per_worker)
        self._fan = self._batched_encoder(self._fan, n)
        self.num_envs = n
        self._alloc()
        if seed is not None:
            self._fan.seed(int(seed) + first_rank)
        return self

    def _batched_encoder(self, sub, n):
        """Auto-encoder observations (config/simplified_object_picking.yaml as shipped: `depth_observation: False`,
        sensor.py:176-222): every worker built its environment with the deferred form of the encoder (``_RankedEnvFn``,
        ``grasp_rl.autoencoder.SimpleAutoEncoder.__new__``) and hands out [4096 pixels | other sensors]; ONE
        ``SimpleAutoEncoder`` in this process -- the template environment's own when it is still around, else one built from
        what the workers recorded -- encodes the images of all n environments in one call per step (``VecBatchedEncoder``).
        Returns `sub` itself when no worker deferred an encoder."""
        recs = sub.deferred_encoders()
        if not any(recs):
            return sub
        try:
            from .. import autoencoder as ae
            flat = [r for per_env in recs for r in per_env]
            dirs = {r.get("model_dir") for r in flat}
            if any(len(per_env) != 1 for per_env in recs) or len(dirs) != 1 or None in dirs:
                raise ValueError("the worker environments built %s deferred encoders over %s weight directories: the batched encoder "
                                 "needs exactly one per environment, all of one model_dir (set GRL_BATCHED_ENCODER=0 to keep one "
                                 "encoder per worker)" % ([len(x) for x in recs], sorted(map(str, dirs))))
            model_dir = flat[0]["model_dir"]
            enc = ae.find_live_encoder(model_dir)
            if enc is None:
                enc = ae.SimpleAutoEncoder(flat[0]["config"])
                enc.load_weights(model_dir)
            enc.ensure_act_batch(2 * n)          # every environment's observation + every terminal observation in one call
            wrapped = VecBatchedEncoder(sub, enc)
            if tuple(wrapped.observation_space.shape) != tuple(self.observation_space.shape):
                raise ValueError("the workers' observations %s encode to %s, the template environment announced %s"
                                 % (tuple(sub.observation_space.shape), tuple(wrapped.observation_space.shape),
                                    tuple(self.observation_space.shape)))
            return wrapped
        except Exception:
            sub.close()
            raise

    def step_async(self, actions):
        if self._fan is not None:
            return self._fan.step_async(actions)
        self.actions = actions

    def step_wait(self):
        if self._fan is not None:
            out = self._fan.step_wait()
            self.buf_infos = self._fan.buf_infos
            return out
        for i, env in enumerate(self.envs):
            obs, self.buf_rews[i], self.buf_dones[i], self.buf_infos[i] = env.step(self.actions[i])
            if self.buf_dones[i]:
                self.buf_infos[i] = dict(self.buf_infos[i])
                self.buf_infos[i]["terminal_observation"] = obs
                obs = env.reset()
            self.buf_obs[i] = obs
        return self.buf_obs.copy(), self.buf_rews.copy(), self.buf_dones.copy(), list(self.buf_infos)

    def reset(self):
        if self._fan is not None:
            return self._fan.reset()
        for i, env in enumerate(self.envs):
            self.buf_obs[i] = env.reset()
        return self.buf_obs.copy()

    def close(self):
        if self._fan is not None:
            return self._fan.close()             # (the template was closed when the workers took over)
        for env in self.envs:
            if hasattr(env, "close"):
                env.close()

    def render(self, *a, **k):
        if self._fan is not None:
            return self._fan.render(*a, **k)
        return self.envs[0].render(*a, **k)

    def seed(self, seed=None):
        if self._fan is not None:
            return self._fan.seed(seed)
        return [env.seed(None if seed is None else seed + i) if hasattr(env, "seed") else None
                for i, env in enumerate(self.envs)]

    def _targets(self, indices):
        if indices is None:
            return self.envs
        if isinstance(indices, int):
            return [self.envs[indices]]
        return [self.envs[i] for i in indices]

    def get_attr(self, name, indices=None):
        if self._fan is not None:
            return self._fan.get_attr(name, indices)
        return [getattr(e, name) for e in self._targets(indices)]

    def set_attr(self, name, value, indices=None):
        if self._fan is not None:
            return self._fan.set_attr(name, value, indices)
        for e in self._targets(indices):
            setattr(e, name, value)

    def env_method(self, name, *args, indices=None, **kwargs):
        if self._fan is not None:
            return self._fan.env_method(name, *args, indices=indices, **kwargs)
        return [getattr(e, name)(*args, **kwargs) for e in self._targets(indices)]


def _main_module_imports():
    """Names of the modules the running script imported at its top level.  The reference's environment factory --
    ``lambda: gym.make('gripper-env-v0', ...)`` -- relies on a side effect of the script's imports (``import manipulation_main``
    registers the env, train_stable_baselines.py:10).  Worker processes of the forkserver / spawn start methods normally get
    that by re-importing the parent's ``__main__`` from its file (also under ``runpy.run_path``, i.e. ``python -m
    grasp_rl.dp_run script.py``: tests/test_subproc_vec_env.py); where ``__main__`` has no file to re-import (``python -c``, an
    interactive session, an embedding application) they import the same top-level modules themselves, best effort, before
    they unpickle the factories."""
    import sys
    import types
    main = sys.modules.get("__main__")
    names = {v.__name__ for v in vars(main).values() if isinstance(v, types.ModuleType)} if main is not None else set()
    return sorted(n for n in names if n not in ("__main__", "__mp_main__"))


def _subproc_worker(remote, parent_remote, env_fns_pickled, preimport=()):
    """Worker loop of SubprocVecEnv: owns ONE OR MORE environments (`envs_per_worker`), answers commands over a pipe;
    every command carries / returns one entry per owned environment.  After an "shm" command the observations are
    written into this worker's slots of a shared-memory block and the pipe only carries `None` in their place (a
    64x64x5 float32 observation is 80 KB: pickling it through a pipe costs more than the rest of the exchange)."""
    parent_remote.close()
    import importlib
    for name in preimport:
        try:
            importlib.import_module(name)
        except Exception:       # noqa: BLE001  (best effort: a module that cannot be imported here is the factory's problem)
            pass
    envs = [fn() for fn in pickle.loads(env_fns_pickled)]
    shm, slots, fast = None, None, None

    def out(k, obs):
        if slots is None:
            return obs
        slots[k][...] = obs
        return None
    try:
        while True:
            msg = remote.recv_bytes()
            if msg == b"s":       # the per-step exchange: actions, rewards and done flags through the shared block as well
                acts, rews, dones = fast
                infos, plain = [], True
                for k, env in enumerate(envs):
                    obs, rew, done, info = env.step(acts[k].copy() if acts[k].ndim else acts[k][()])
                    if done:
                        info = dict(info)
                        info["terminal_observation"] = np.array(obs, copy=True)
                        obs = env.reset()
                    slots[k][...] = obs
                    rews[k], dones[k] = rew, done
                    plain = plain and not info
                    infos.append(info)
                if plain:
                    remote.send_bytes(b"\0")            # nothing to report: one byte instead of a pickled list
                else:
                    remote.send(infos)
                continue
            cmd, data = pickle.loads(msg)
            if cmd == "step":
                res = []
                for k, (env, action) in enumerate(zip(envs, data)):
                    obs, rew, done, info = env.step(action)
                    if done:
                        info = dict(info)
                        info["terminal_observation"] = np.array(obs, copy=True)
                        obs = env.reset()
                    res.append((out(k, obs), rew, done, info))
                remote.send(res)
            elif cmd == "reset":
                remote.send([out(k, env.reset()) for k, env in enumerate(envs)])
            elif cmd == "shm":
                from multiprocessing import shared_memory
                name, first, shape, dtype, extra = data
                shm = shared_memory.SharedMemory(name=name)
                item = int(np.prod(shape)) * np.dtype(dtype).itemsize
                slots = [np.ndarray(shape, dtype=dtype, buffer=shm.buf, offset=(first + k) * item) for k in range(len(envs))]
                if extra is not None:
                    n_all, a_off, a_shape, a_dtype, r_off, d_off = extra
                    a_all = np.ndarray((n_all,) + tuple(a_shape), dtype=a_dtype, buffer=shm.buf, offset=a_off)
                    r_all = np.ndarray((n_all,), dtype=np.float64, buffer=shm.buf, offset=r_off)
                    d_all = np.ndarray((n_all,), dtype=np.uint8, buffer=shm.buf, offset=d_off)
                    fast = (a_all[first:first + len(envs)], r_all[first:first + len(envs)], d_all[first:first + len(envs)])
                remote.send(True)
            elif cmd == "seed":
                remote.send([env.seed(sd) if hasattr(env, "seed") else None for env, sd in zip(envs, data)])
            elif cmd == "spaces":
                remote.send((envs[0].observation_space, envs[0].action_space))
            elif cmd == "get_attr":
                remote.send([getattr(envs[k], data[0]) for k in data[1]])
            elif cmd == "set_attr":
                for k in data[2]:
                    setattr(envs[k], data[0], data[1])
                remote.send(None)
            elif cmd == "env_method":
                remote.send([getattr(envs[k], data[0])(*data[1], **data[2]) for k in data[3]])
            elif cmd == "deferred_encoders":     # what the DeferredEncoders built with these environments recorded
                from ..autoencoder import deferred_records
                remote.send(deferred_records())
            elif cmd == "render":
                remote.send(envs[0].render(*data[0], **data[1]) if hasattr(envs[0], "render") else None)
            elif cmd == "close":
                for env in envs:
                    if hasattr(env, "close"):
                        env.close()
                slots = fast = None
                if shm is not None:
                    shm.close()
                remote.close()
                break
            else:
                raise NotImplementedError(cmd)
    except (EOFError, KeyboardInterrupt):
        pass


class SubprocVecEnv(VecEnv):
    """Worker PROCESSES for the environments (PyBullet steps are CPU-bound and hold the GIL): the fan-out
    of BASELINE configs 2 and 5 -- N simulators on the host cores feeding one engine.  Same protocol as
    stable-baselines' SubprocVecEnv: ``step_async`` posts the actions, ``step_wait`` collects
    ``(obs, rew, done, info)``; an env resets itself when its episode ends and reports the last
    observation in ``info['terminal_observation']``.  ``start_method``: 'forkserver' / 'spawn' keep the
    workers free of this process's HIP context ('fork' after HIP initialisation is unsafe).
    ``envs_per_worker`` (default 1 = stable-baselines): consecutive environments share a worker, which steps them one
    after the other -- fewer processes to wake per step when the simulator is cheap (16 free envs on 4 workers: the
    parent waits for 4 pipes instead of 16; with PyBullet, milliseconds per step, keep 1)."""

    def __init__(self, env_fns, start_method=None, shared_memory=True, envs_per_worker=1):
        import multiprocessing as mp
        if start_method is None:
            start_method = "forkserver" if "forkserver" in mp.get_all_start_methods() else "spawn"
        ctx = mp.get_context(start_method)
        self.waiting = False
        self.closed = False
        k = max(1, int(envs_per_worker))
        groups = [list(range(i, min(i + k, len(env_fns)))) for i in range(0, len(env_fns), k)]
        self._groups = groups
        self._owner = {e: (w, j) for w, g in enumerate(groups) for j, e in enumerate(g)}
        self.remotes, self.work_remotes = zip(*[ctx.Pipe() for _ in groups])
        self.processes = []
        try:
            import cloudpickle as _pk
        except ImportError:      # plain pickle: env_fns must be importable callables
            _pk = pickle
        pre = _main_module_imports()
        for work_remote, remote, g in zip(self.work_remotes, self.remotes, groups):
            proc = ctx.Process(target=_subproc_worker, args=(work_remote, remote, _pk.dumps([env_fns[e] for e in g]), pre), daemon=True)
            proc.start()
            self.processes.append(proc)
            work_remote.close()
        self.remotes[0].send(("spaces", None))
        obs_space, act_space = self.remotes[0].recv()
        super().__init__(len(env_fns), obs_space, act_space)
        self.buf_infos = [{} for _ in range(self.num_envs)]
        # observations through shared memory (array observation spaces): one block, one slot per environment
        self._shm, self._shm_arr = None, None
        self._act_arr = self._rew_arr = self._done_arr = None
        self._fds = [r.fileno() for r in self.remotes]
        shape = tuple(getattr(obs_space, "shape", ()) or ())
        if shared_memory and shape:
            from multiprocessing import shared_memory as _sm
            dtype = np.dtype(getattr(obs_space, "dtype", np.float32))
            n = self.num_envs
            obs_bytes = n * int(np.prod(shape)) * dtype.itemsize
            # actions, rewards and done flags travel through the same block when the action space is an array type
            # (Box / Discrete): the pipes then carry one byte each way per step instead of pickled tuples
            a_shape, a_dtype = getattr(act_space, "shape", None), getattr(act_space, "dtype", None)
            extra = None
            size = obs_bytes
            if a_shape is not None and a_dtype is not None:
                a_dtype = np.dtype(a_dtype)
                a_off = (obs_bytes + 63) // 64 * 64
                r_off = (a_off + n * int(np.prod(a_shape, dtype=np.int64)) * a_dtype.itemsize + 63) // 64 * 64
                d_off = r_off + n * 8
                size = d_off + n
                extra = (n, a_off, tuple(a_shape), a_dtype.str, r_off, d_off)
            self._shm = _sm.SharedMemory(create=True, size=max(1, size))
            self._shm_arr = np.ndarray((n,) + shape, dtype=dtype, buffer=self._shm.buf)
            if extra is not None:
                self._act_arr = np.ndarray((n,) + tuple(a_shape), dtype=a_dtype, buffer=self._shm.buf, offset=a_off)
                self._rew_arr = np.ndarray((n,), dtype=np.float64, buffer=self._shm.buf, offset=r_off)
                self._done_arr = np.ndarray((n,), dtype=np.uint8, buffer=self._shm.buf, offset=d_off)
            for remote, g in zip(self.remotes, groups):
                remote.send(("shm", (self._shm.name, g[0], shape, dtype.str, extra)))
            for remote in self.remotes:
                remote.recv()

    def _stack_obs(self, obs):
        dtype = getattr(self.observation_space, "dtype", np.float32)
        if self._shm_arr is not None:
            return np.array(self._shm_arr, dtype=dtype, copy=True)     # the next step overwrites the slots
        return np.stack(obs).astype(dtype)

    _STEP_MSG = b"\x00\x00\x00\x01s"      # multiprocessing.Connection framing of send_bytes(b"s"): length, payload

    def _recv_framed(self, fd):
        """One Connection message read straight from the pipe: the workers' replies are a few hundred bytes, written with
        one write(), so one read() normally returns header and payload together."""
        buf = os.read(fd, 65536)
        if not buf:
            raise EOFError("environment worker closed its pipe")
        while len(buf) < 4:
            chunk = os.read(fd, 65536)
            if not chunk:
                raise EOFError("environment worker closed its pipe")
            buf += chunk
        n = int.from_bytes(buf[:4], "big", signed=True)
        if n < 0:                    # (the 8-byte length form of messages over 2 GB: not something a step reply is)
            raise RuntimeError("unexpected message framing from an environment worker")
        while len(buf) < 4 + n:
            chunk = os.read(fd, 4 + n - len(buf))
            if not chunk:
                raise EOFError("environment worker closed its pipe")
            buf += chunk
        return buf[4:4 + n]

    def step_async(self, actions):
        if self._act_arr is not None:
            self._act_arr[...] = np.asarray(actions).reshape(self._act_arr.shape)
            for fd in self._fds:
                os.write(fd, self._STEP_MSG)
        else:
            for remote, g in zip(self.remotes, self._groups):
                remote.send(("step", [actions[e] for e in g]))
        self.waiting = True

    def step_wait(self):
        if self._act_arr is not None:
            infos = []
            for fd, g in zip(self._fds, self._groups):
                m = self._recv_framed(fd)
                infos.extend([{} for _ in g] if m == b"\0" else pickle.loads(m))
            self.waiting = False
            self.buf_infos = infos
            return (self._stack_obs(None), self._rew_arr.astype(np.float32), self._done_arr.astype(bool), infos)
        results = [r for remote in self.remotes for r in remote.recv()]
        self.waiting = False
        obs, rews, dones, infos = zip(*results)
        self.buf_infos = list(infos)
        return (self._stack_obs(obs), np.asarray(rews, dtype=np.float32), np.asarray(dones, dtype=bool), list(infos))

    def reset(self):
        for remote in self.remotes:
            remote.send(("reset", None))
        obs = [o for remote in self.remotes for o in remote.recv()]
        return self._stack_obs(obs)

    def seed(self, seed=None):
        for remote, g in zip(self.remotes, self._groups):
            remote.send(("seed", [None if seed is None else seed + e for e in g]))
        return [x for remote in self.remotes for x in remote.recv()]

    def close(self):
        if self.closed:
            return
        gone = (EOFError, BrokenPipeError, ConnectionResetError, OSError)      # a worker that died must not keep close() from
        if self.waiting:                                                        # releasing the others and the shared block
            for remote in self.remotes:
                try:
                    remote.recv_bytes()
                except gone:
                    pass
            self.waiting = False
        for remote in self.remotes:
            try:
                remote.send(("close", None))
            except gone:
                pass
        for proc in self.processes:
            proc.join()
        if self._shm is not None:
            self._shm_arr = self._act_arr = self._rew_arr = self._done_arr = None
            self._shm.close()
            try:
                self._shm.unlink()
            except FileNotFoundError:       # (a resource tracker shared with forked workers got there first)
                pass
            self._shm = None
        self.closed = True

    def render(self, *a, **k):
        self.remotes[0].send(("render", (a, k)))
        return self.remotes[0].recv()

    def _by_worker(self, indices):
        """environment indices -> [(worker, [local indices], [positions in the answer])], in worker order"""
        if indices is None:
            indices = range(self.num_envs)
        elif isinstance(indices, int):
            indices = [indices]
        per = {}
        for pos, e in enumerate(indices):
            w, j = self._owner[e]
            per.setdefault(w, ([], []))
            per[w][0].append(j)
            per[w][1].append(pos)
        return [(w, loc, pos) for w, (loc, pos) in sorted(per.items())], len(list(indices)) if not isinstance(indices, range) else len(indices)

    def _gathered(self, cmd, payload, indices):
        plan, n = self._by_worker(indices)
        for w, loc, _ in plan:
            self.remotes[w].send((cmd, payload + (loc,)))
        out = [None] * n
        for w, _, pos in plan:
            ans = self.remotes[w].recv()
            if ans is not None:
                for p, a in zip(pos, ans):
                    out[p] = a
        return out

    def deferred_encoders(self):
        """Per environment: the records of the ``grasp_rl.autoencoder.DeferredEncoder``s its worker process built (a worker that
        owns several environments reports its list split evenly over them)."""
        out = []
        for remote, g in zip(self.remotes, self._groups):
            remote.send(("deferred_encoders", None))
            recs = remote.recv()
            k = len(recs) // len(g) if len(recs) % len(g) == 0 else None
            for j in range(len(g)):
                out.append(recs[j * k:(j + 1) * k] if k is not None else list(recs))
        return out

    def get_attr(self, name, indices=None):
        return self._gathered("get_attr", (name,), indices)

    def set_attr(self, name, value, indices=None):
        self._gathered("set_attr", (name, value), indices)

    def env_method(self, name, *args, indices=None, **kwargs):
        return self._gathered("env_method", (name, args, kwargs), indices)


class VecEnvWrapper(VecEnv):
    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        super().__init__(venv.num_envs, observation_space or venv.observation_space,
                         action_space or venv.action_space)

    def step_async(self, actions):
        self.venv.step_async(actions)

    def close(self):
        return self.venv.close()

    def render(self, *a, **k):
        return self.venv.render(*a, **k)

    def seed(self, seed=None):
        return self.venv.seed(seed)

    def get_attr(self, name, indices=None):
        return self.venv.get_attr(name, indices)

    def set_attr(self, name, value, indices=None):
        return self.venv.set_attr(name, value, indices)

    def env_method(self, name, *args, indices=None, **kwargs):
        return self.venv.env_method(name, *args, indices=indices, **kwargs)

    def __getattr__(self, name):            # e.g. .envs / .buf_infos of the wrapped DummyVecEnv
        if name.startswith("__") or name == "venv":
            raise AttributeError(name)
        return getattr(self.venv, name)


class VecBatchedEncoder(VecEnvWrapper):
    """Auto-encoder features for vectorised environments, batched.  The wrapped environments run with
    ``grasp_rl.autoencoder.DeferredEncoder`` (their observation = the 4096 floats of the filtered depth image followed
    by the other sensor readings, robot.py:185-190); this wrapper replaces the image part by
    ``encoder.encode(images of all N environments)`` -- one device call per step instead of one per environment -- and
    presents the observation space the reference's env would have had ([100 + k], sensor.py:196).  `encoder`: a
    ``SimpleAutoEncoder`` (weights loaded); terminal observations in ``info`` are encoded in the same call."""

    def __init__(self, venv, encoder, n_pixels=64 * 64):
        self.encoder, self.n_pixels = encoder, int(n_pixels)
        raw = int(np.prod(venv.observation_space.shape))
        if raw <= self.n_pixels:
            raise ValueError("the wrapped observation must be [image pixels | other sensor readings]")
        self.n_extra = raw - self.n_pixels
        dim = int(np.prod(encoder.encoding_shape)) + self.n_extra
        space = sp.Box(-np.inf, np.inf, shape=(dim,), dtype=np.float32)
        super().__init__(venv, observation_space=space)

    def _encode(self, raw):
        raw = np.asarray(raw, np.float32).reshape(-1, self.n_pixels + self.n_extra)
        z = self.encoder.encode(raw[:, :self.n_pixels].reshape(-1, 64, 64, 1))
        return np.concatenate([np.asarray(z, np.float32).reshape(raw.shape[0], -1), raw[:, self.n_pixels:]], axis=1)

    def reset(self):
        return self._encode(self.venv.reset())

    def step_wait(self):
        obs, rews, dones, infos = self.venv.step_wait()
        term = [i for i, inf in enumerate(infos) if isinstance(inf, dict) and "terminal_observation" in inf]
        rows = [obs] + [np.asarray(infos[i]["terminal_observation"], np.float32).reshape(1, -1) for i in term]
        z = self._encode(np.concatenate([np.asarray(r, np.float32).reshape(-1, self.n_pixels + self.n_extra) for r in rows]))
        n = np.asarray(obs).shape[0]
        infos = list(infos)
        for k, i in enumerate(term):
            infos[i] = dict(infos[i], terminal_observation=z[n + k])
        self.buf_infos = infos          # (what `task.buf_infos[0]` of utils.py:76 reads: the ENCODED terminal observations)
        return z[:n], rews, dones, infos


class VecNormalize(VecEnvWrapper):
    def __init__(self, venv, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0, clip_reward=10.0,
                 gamma=0.99, epsilon=1e-8):
        super().__init__(venv)
        self.obs_rms = RunningMeanStd(shape=tuple(self.observation_space.shape))
        self.ret_rms = RunningMeanStd(shape=())
        self.clip_obs, self.clip_reward = clip_obs, clip_reward
        self.ret = np.zeros(self.num_envs)
        self.gamma, self.epsilon = gamma, epsilon
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.old_obs = np.array([])
        self.old_rews = np.array([])
        self._dev = None
        self.observed_serial = None      # engine.observe serial of the observations last handed out (None: not on the device)

    # -- observation statistics on the device (opt-in, grasp_rl.sb.SAC(device_norm=True)) -------------------------
    def attach_device(self, engine):
        """From now on `obs_rms` is maintained by the engine (grl_norm_update: the same float32 batch moments and
        float64 Chan merge, on the GPU) and, while training, `step_wait` / `reset` hand out RAW observations for the
        engine to normalise where it consumes them (grl_act flag 2, sample-time gather).  The host copy is refreshed by
        `pull_device_stats` -- automatically before pickling / `save` and in `sync_envs_normalization`."""
        self._dev = engine
        engine.set_obs_stats(self.obs_rms.mean, self.obs_rms.var, float(self.ret_rms.var))
        engine.set_running_stats(self.obs_rms.mean, self.obs_rms.var, self.obs_rms.count)

    def _observe(self, obs):
        """One upload per env step: the statistics are updated from the device copy, and `observed_serial` tells the learn
        loop that the action and the replay rows can come from it too (engine.observe)."""
        if hasattr(self._dev, "observe") and obs.shape[0] <= getattr(self._dev, "observe_rows", 0):
            self.observed_serial = self._dev.observe(obs, update_stats=True)
        else:
            self._dev.norm_update(obs)

    def detach_device(self):
        if self._dev is not None:
            self.pull_device_stats()
        self._dev = None

    def pull_device_stats(self):
        if self._dev is not None:
            self.obs_rms.mean, self.obs_rms.var, self.obs_rms.count = self._dev.get_obs_stats(tuple(self.observation_space.shape))

    @property
    def hands_out_raw_observations(self):
        return self._dev is not None and self.training and self.norm_obs

    # -- pickling: everything but the wrapped env (stable-baselines convention) ------------------
    def __getstate__(self):
        self.pull_device_stats()
        state = self.__dict__.copy()
        for k in ("venv", "class_attributes", "ret", "_dev", "observed_serial"):
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.venv = None
        self._dev = None
        self.observed_serial = None

    def set_venv(self, venv):
        if self.venv is not None:
            raise ValueError("this VecNormalize already wraps an environment")
        VecEnvWrapper.__init__(self, venv)
        if tuple(self.obs_rms.mean.shape) != tuple(self.observation_space.shape):
            raise ValueError("venv is incompatible with the saved statistics")
        self.ret = np.zeros(self.num_envs)

    # -- stepping ---------------------------------------------------------------------------------
    def step_wait(self):
        obs, rews, dones, infos = self.venv.step_wait()
        self.old_obs, self.old_rews = obs, rews
        self.observed_serial = None
        if self.hands_out_raw_observations:
            self._observe(obs)                         # statistics on the device; the consumer normalises there too
        else:
            if self._dev is not None:                  # attached but frozen (a callback cleared `training`): the host
                self.pull_device_stats()               # arithmetic below needs the statistics the device holds
            if self.training:
                self.obs_rms.update(obs)               # batch moments in the observations' own dtype (float32), as NumPy forms them
            obs = self.normalize_obs(obs)
        if self.training:
            self.ret = self.ret * self.gamma + rews
            self.ret_rms.update(self.ret)
        rews = self.normalize_reward(rews)
        self.ret[dones] = 0
        return obs, rews, dones, infos

    def reset(self):
        obs = self.venv.reset()
        self.old_obs = obs
        self.ret = np.zeros(self.num_envs)
        self.observed_serial = None
        if self.hands_out_raw_observations:
            self._observe(obs)
            return obs
        if self._dev is not None:
            self.pull_device_stats()
        if self.training:
            self.obs_rms.update(obs)
        return self.normalize_obs(obs)

    def normalize_obs(self, obs):
        if self.norm_obs:
            x = obs - self.obs_rms.mean                 # float64 (the statistics are): same arithmetic as
            x /= np.sqrt(self.obs_rms.var + self.epsilon)   # clip((obs - mean) / sqrt(var + eps), -c, c), in place
            obs = np.clip(x, -self.clip_obs, self.clip_obs, out=x)
        return obs

    def normalize_reward(self, reward):
        if self.norm_reward:
            reward = np.clip(reward / np.sqrt(self.ret_rms.var + self.epsilon), -self.clip_reward, self.clip_reward)
        return reward

    def get_original_obs(self):
        return self.old_obs.copy()

    def get_original_reward(self):
        return self.old_rews.copy()

    # -- persistence --------------------------------------------------------------------------------
    def save(self, path):
        with open(path, "wb") as f:
            pickle.dump(self, f)

    @staticmethod
    def load(load_path, venv):
        """Reads both our own pickles and the ones the reference ships under trained_models/ (those name
        ``stable_baselines...VecNormalize`` / ``RunningMeanStd`` / ``gym.spaces`` classes)."""
        with open(load_path, "rb") as f:
            obj = _CompatUnpickler(f).load()
        if not isinstance(obj, VecNormalize):
            raise TypeError("%s does not contain a VecNormalize object" % load_path)
        for rms in (obj.obs_rms, obj.ret_rms):
            rms.mean = np.asarray(rms.mean, np.float64)
            rms.var = np.asarray(rms.var, np.float64)
        obj.venv = None
        obj.set_venv(venv)
        return obj


class _CompatUnpickler(pickle.Unpickler):
    """Unpickler for files that may come from elsewhere (``vecnormalize.pkl``, the ``:serialized:`` members of
    a model zip).  Only what those files legitimately hold can be constructed: NumPy arrays / dtypes / random
    states, plain containers, ``VecNormalize`` / ``RunningMeanStd`` / ``Box`` / ``Discrete`` (whatever package
    path they were pickled under) and the policy *selector* classes.  Any other global -- in particular
    arbitrary callables such as ``os.system`` or a cloudpickled function body -- raises ``UnpicklingError``;
    the model loaders then rebuild the value from the readable copies stored next to it
    (save_util.json_to_data) or from the parameter shapes.  ``GRL_TRUST_PICKLES=1`` lifts the restriction for
    files you produced yourself."""
    _MAP = {
        ("VecNormalize",): lambda: VecNormalize,
        ("RunningMeanStd",): lambda: RunningMeanStd,
        ("Box",): lambda: sp.Box,
        ("Discrete",): lambda: sp.Discrete,
    }
    _BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray",
                 "complex", "slice", "range", "object"}
    _OTHER = {("collections", "OrderedDict"), ("collections", "deque"), ("copyreg", "_reconstructor"),
              ("copy_reg", "_reconstructor"), ("_codecs", "encode")}

    # NumPy globals a pickled array / dtype / scalar / random state needs -- named one by one: a wildcard over the
    # numpy package would also hand out callables such as numpy.testing._private.utils.runstring (an exec wrapper)
    _NUMPY = {("numpy", "ndarray"), ("numpy", "dtype"), ("numpy", "float64"), ("numpy", "float32"), ("numpy", "int64"),
              ("numpy", "int32"), ("numpy", "bool_"), ("numpy", "uint8"),
              ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
              ("numpy.core.numeric", "_frombuffer"),
              ("numpy.random", "__RandomState_ctor"), ("numpy.random._pickle", "__randomstate_ctor"),
              ("numpy.random._pickle", "__generator_ctor"), ("numpy.random._pickle", "__bit_generator_ctor"),
              ("numpy.random.mtrand", "RandomState"), ("numpy.random._mt19937", "MT19937"),
              ("numpy.random._pcg64", "PCG64"), ("numpy.random._generator", "Generator")}

    def find_class(self, module, name):
        if module.startswith("numpy"):
            canon = module.replace("numpy._core", "numpy.core")
            if (canon, name) in self._NUMPY:
                for m in (module, module.replace("numpy.core", "numpy._core"), canon):
                    try:
                        return super().find_class(m, name)
                    except (ImportError, AttributeError):
                        continue
        top = module.split(".")[0]
        if top in ("stable_baselines", "gym", "gymnasium", "grasp_rl"):
            if (name,) in self._MAP:
                return self._MAP[(name,)]()
            if module.endswith("policies"):
                from . import policies as pol
                try:
                    obj = super().find_class(module, name)
                except (ImportError, AttributeError):
                    obj = None
                if isinstance(obj, type) and issubclass(obj, (pol.BasePolicy, pol.AugmentedNatureCnn)):
                    return obj
        if (module in ("builtins", "__builtin__") and name in self._BUILTINS) or (module, name) in self._OTHER:
            return super().find_class("builtins" if module == "__builtin__" else module, name)
        if os.environ.get("GRL_TRUST_PICKLES", "0") == "1":
            return super().find_class(module, name)
        raise pickle.UnpicklingError("global %s.%s is not on the allow-list of files this package reads "
                                     "(set GRL_TRUST_PICKLES=1 for files you created yourself)" % (module, name))


def unwrap_vec_normalize(env):
    while isinstance(env, VecEnvWrapper):
        if isinstance(env, VecNormalize):
            return env
        env = env.venv
    return None


def sync_envs_normalization(env, eval_env):
    """Copy the running statistics of the training env's VecNormalize into the evaluation env's."""
    a, b = env, eval_env
    while isinstance(a, VecEnvWrapper) and isinstance(b, VecEnvWrapper):
        if isinstance(a, VecNormalize) and isinstance(b, VecNormalize):
            a.pull_device_stats()
            b.obs_rms = copy.deepcopy(a.obs_rms)
            b.ret_rms = copy.deepcopy(a.ret_rms)
        a, b = a.venv, b.venv


def requested_num_envs(world=1):
    """GRL_NUM_ENVS: how many environments the JOB trains on (the knob the reference's CLI does not have: its script
    wraps one env, train_stable_baselines.py:52-54; BASELINE configs[1] = 16, configs[4] = 64 over 8 ranks).  Returns this
    process's share, or None when the variable is not set."""
    raw = os.environ.get("GRL_NUM_ENVS", "").strip()
    if not raw:
        return None
    n = int(raw)
    if n < 1:
        raise ValueError("GRL_NUM_ENVS must be a positive integer, got %r" % raw)
    if n % max(1, world):
        raise ValueError("GRL_NUM_ENVS=%d environments do not divide over %d ranks" % (n, world))
    return n // max(1, world)


def expand_training_env(env, n=None, rank=0, world=1):
    """The training env a model was handed, fanned out to this process's share of GRL_NUM_ENVS worker environments when
    it is (a wrapper chain around) a single-factory ``DummyVecEnv``; anything else -- no request, a request of 1, an env
    that already is vectorised by its maker -- is returned as it is.  Wrappers built around the single env before the
    model existed (sb_helper.py:117-119 wraps VecNormalize first) get their per-env state re-sized.
    GRL_ENVS_PER_WORKER (default 1), GRL_ENV_START_METHOD (forkserver | spawn | fork; fork only before HIP is
    initialised) and GRL_ENV_SEED (env k of the job is seeded GRL_ENV_SEED + k) tune the workers."""
    if n is None:
        n = requested_num_envs(world)
    if n is None or env is None:
        return env
    chain, inner = [], env
    while isinstance(inner, VecEnvWrapper):
        chain.append(inner)
        inner = inner.venv
    if not isinstance(inner, DummyVecEnv):
        return env
    if inner.fanned_out or (n > 1 and len(getattr(inner, "_env_fns", ())) == 1):
        if inner.fanned_out and inner.num_envs != n:
            raise ValueError("this training env already fans out to %d worker environments; GRL_NUM_ENVS now asks for %d per "
                             "process (another world size?): build a fresh env for the new model" % (inner.num_envs, n))
        # wrappers built around the single env keep per-env state in its 1-env shape: VecNormalize is re-sized here, any other
        # wrapper must say how (a `resize_envs(n)` method) -- a silent fan-out would break it on the first step
        stuck = [type(w).__name__ for w in chain
                 if not isinstance(w, (VecNormalize, VecBatchedEncoder)) and not hasattr(w, "resize_envs")]
        if stuck and inner.num_envs != n:
            raise ValueError("GRL_NUM_ENVS=%d cannot fan out beneath %s: the wrapper holds per-env state sized for one environment "
                             "and has no resize_envs(n) hook (wrap after the fan-out, or unset GRL_NUM_ENVS)" % (n, ", ".join(stuck)))
        seed = os.environ.get("GRL_ENV_SEED")
        inner.fan_out(n, envs_per_worker=int(os.environ.get("GRL_ENVS_PER_WORKER", "1")),
                      start_method=os.environ.get("GRL_ENV_START_METHOD") or None, first_rank=rank * n,
                      seed=None if seed in (None, "") else int(seed))
    for w in chain:
        if w.num_envs != inner.num_envs:
            w.num_envs = inner.num_envs
            if isinstance(w, VecNormalize):
                w.ret = np.zeros(w.num_envs)
                w.old_obs, w.old_rews = np.array([]), np.array([])      # (the 1-env observation of a reset before the fan-out)
            elif hasattr(w, "resize_envs"):
                w.resize_envs(w.num_envs)
    return env
