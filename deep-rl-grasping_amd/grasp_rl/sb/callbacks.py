"""Callback protocol of stable-baselines 2.10 as the reference's callbacks rely on it
(sb_helper.py:25-54 TensorboardCallback, base_callbacks.py:16-245 EvalCallback /
SaveVecNormalizeCallback / TrainingTimeCallback subclass these): ``init_callback(model)``,
``on_training_start(locals, globals)``, ``on_rollout_start``, ``on_step() -> bool`` (False aborts
training), ``on_rollout_end``, ``on_training_end``; attributes ``model``, ``training_env``,
``n_calls``, ``num_timesteps``, ``locals``, ``globals``."""
import os

import numpy as np

from . import logger as sb_logger
from .evaluation import evaluate_policy
from .vec_env import DummyVecEnv, VecEnv, sync_envs_normalization


class BaseCallback:
    def __init__(self, verbose=0):
        self.model = None
        self.training_env = None
        self.n_calls = 0
        self.num_timesteps = 0
        self.verbose = verbose
        self.locals, self.globals = None, None
        self.logger = sb_logger
        self.parent = None

    def init_callback(self, model):
        self.model = model
        self.training_env = model.get_env()
        self._init_callback()

    def _init_callback(self):
        pass

    def on_training_start(self, locals_, globals_):
        self.locals, self.globals = locals_, globals_
        self._on_training_start()

    def _on_training_start(self):
        pass

    def on_rollout_start(self):
        self._on_rollout_start()

    def _on_rollout_start(self):
        pass

    def _on_step(self):
        return True

    def on_step(self):
        self.n_calls += 1
        self.num_timesteps = self.model.num_timesteps
        return self._on_step()

    def on_training_end(self):
        self._on_training_end()

    def _on_training_end(self):
        pass

    def on_rollout_end(self):
        self._on_rollout_end()

    def _on_rollout_end(self):
        pass

    def update_locals(self, locals_):
        if self.locals is None:
            self.locals = {}
        self.locals.update(locals_)


class EventCallback(BaseCallback):
    def __init__(self, callback=None, verbose=0):
        super().__init__(verbose)
        self.callback = callback
        if callback is not None:
            callback.parent = self

    def init_callback(self, model):
        super().init_callback(model)
        if self.callback is not None:
            self.callback.init_callback(model)

    def _on_training_start(self):
        if self.callback is not None:
            self.callback.on_training_start(self.locals, self.globals)

    def _on_event(self):
        return self.callback.on_step() if self.callback is not None else True

    def _on_step(self):
        return True


class CallbackList(BaseCallback):
    def __init__(self, callbacks):
        super().__init__()
        self.callbacks = list(callbacks)

    def _init_callback(self):
        for c in self.callbacks:
            c.init_callback(self.model)

    def _on_training_start(self):
        for c in self.callbacks:
            c.on_training_start(self.locals, self.globals)

    def _on_rollout_start(self):
        for c in self.callbacks:
            c.on_rollout_start()

    def _on_step(self):
        ok = True
        for c in self.callbacks:
            ok = c.on_step() and ok      # every callback runs; any False aborts training
        return ok

    def _on_rollout_end(self):
        for c in self.callbacks:
            c.on_rollout_end()

    def _on_training_end(self):
        for c in self.callbacks:
            c.on_training_end()

    def update_locals(self, locals_):
        super().update_locals(locals_)
        for c in self.callbacks:
            c.update_locals(locals_)


class ConvertCallback(BaseCallback):
    """Wraps a legacy functional callback ``f(locals, globals) -> bool``."""

    def __init__(self, fn, verbose=0):
        super().__init__(verbose)
        self.fn = fn

    def _on_step(self):
        r = self.fn(self.locals, self.globals) if self.fn is not None else True
        return True if r is None else bool(r)


# Callbacks of the reference that are known not to look at the observations in `locals` (sb_helper.py:25-54,
# base_callbacks.py:16-245: they read attributes of the envs, evaluate on their own env, save files, keep time)
_REFERENCE_CALLBACKS = {"EvalCallback", "SaveVecNormalizeCallback", "TrainingTimeCallback", "TensorboardCallback", "CheckpointCallback"}
_REFERENCE_MODULES = {"base_callbacks", "sb_helper", "manipulation_main.training.base_callbacks", "manipulation_main.training.sb_helper"}


def may_read_observations(callback):
    """False when every callback in the tree is known not to read `locals['new_obs']` / `locals['obs']`: this module's
    classes, the reference's own (by class and module name), or a class that declares ``reads_observations = False``.
    ``SAC(device_norm='auto')`` keeps the VecNormalize statistics on the device only then -- with them there the learn loop
    sees RAW observations (the engine normalises where it consumes them), which a callback that looks at them would notice."""
    if callback is None:
        return False
    kids = []
    if isinstance(callback, CallbackList):
        kids = list(callback.callbacks)
    elif isinstance(callback, EventCallback) and callback.callback is not None:
        kids = [callback.callback]
    cls = type(callback)
    declared = getattr(cls, "reads_observations", None)
    own = cls.__module__ == __name__ and cls is not ConvertCallback
    ref = cls.__name__ in _REFERENCE_CALLBACKS and cls.__module__ in _REFERENCE_MODULES
    if declared is True or (declared is None and not own and not ref):
        return True
    return any(may_read_observations(k) for k in kids)


def as_callback(callback):
    if callback is None:
        return BaseCallback()
    if isinstance(callback, (list, tuple)):
        return CallbackList(callback)
    if isinstance(callback, BaseCallback):
        return callback
    return ConvertCallback(callback)


class CheckpointCallback(BaseCallback):
    def __init__(self, save_freq, save_path, name_prefix="rl_model", verbose=0):
        super().__init__(verbose)
        self.save_freq, self.save_path, self.name_prefix = save_freq, save_path, name_prefix

    def _init_callback(self):
        if self.save_path is not None:
            os.makedirs(self.save_path, exist_ok=True)

    def _on_step(self):
        if self.n_calls % self.save_freq == 0:
            self.model.save(os.path.join(self.save_path, "%s_%d_steps" % (self.name_prefix, self.num_timesteps)))
        return True


class EvalCallback(EventCallback):
    def __init__(self, eval_env, callback_on_new_best=None, n_eval_episodes=5, eval_freq=10000, log_path=None,
                 best_model_save_path=None, deterministic=True, render=False, verbose=1):
        super().__init__(callback_on_new_best, verbose=verbose)
        self.n_eval_episodes, self.eval_freq = n_eval_episodes, eval_freq
        self.best_mean_reward, self.last_mean_reward = -np.inf, -np.inf
        self.deterministic, self.render = deterministic, render
        if not isinstance(eval_env, VecEnv):
            eval_env = DummyVecEnv([lambda: eval_env])
        self.eval_env = eval_env
        self.best_model_save_path = best_model_save_path
        self.log_path = os.path.join(log_path, "evaluations") if log_path is not None else None
        self.evaluations_results, self.evaluations_timesteps, self.evaluations_length = [], [], []

    def _init_callback(self):
        if self.best_model_save_path is not None:
            os.makedirs(self.best_model_save_path, exist_ok=True)
        if self.log_path is not None:
            os.makedirs(os.path.dirname(self.log_path), exist_ok=True)

    def _on_step(self):
        if self.eval_freq > 0 and self.n_calls % self.eval_freq == 0:
            sync_envs_normalization(self.training_env, self.eval_env)
            rewards, lengths = evaluate_policy(self.model, self.eval_env, n_eval_episodes=self.n_eval_episodes,
                                               render=self.render, deterministic=self.deterministic,
                                               return_episode_rewards=True)
            if self.log_path is not None:
                self.evaluations_timesteps.append(self.num_timesteps)
                self.evaluations_results.append(rewards)
                self.evaluations_length.append(lengths)
                np.savez(self.log_path, timesteps=self.evaluations_timesteps, results=self.evaluations_results,
                         ep_lengths=self.evaluations_length)
            mean_reward = float(np.mean(rewards))
            self.last_mean_reward = mean_reward
            if self.verbose > 0:
                print("Eval num_timesteps=%d, episode_reward=%.2f +/- %.2f" % (self.num_timesteps, mean_reward, np.std(rewards)))
            if mean_reward > self.best_mean_reward:
                if self.best_model_save_path is not None:
                    self.model.save(os.path.join(self.best_model_save_path, "best_model"))
                self.best_mean_reward = mean_reward
                if self.callback is not None:
                    return self._on_event()
        return True
