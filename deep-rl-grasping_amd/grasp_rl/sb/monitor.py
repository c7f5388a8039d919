"""``Monitor(env, filename)`` episode logger (reference call site: train_stable_baselines.py:54).
Writes the stable-baselines monitor CSV: a ``#{json}`` header then ``r,l,t`` rows (+ extra
``info_keywords`` columns, which is how the reference's bdq_sb fork adds success/curriculum
columns)."""
import csv
import json
import os
import time

# number of the environment being constructed in this process (set by grasp_rl.sb.vec_env._RankedEnvFn around the factory
# call inside a fan-out worker; 0 everywhere else): environment k > 0 logs to <filename>.env<k>.monitor.csv
ENV_RANK = 0


class Monitor:
    EXT = "monitor.csv"

    def __init__(self, env, filename=None, allow_early_resets=True, reset_keywords=(), info_keywords=()):
        self.env = env
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self.t_start = time.time()
        self.file_handler, self.logger = None, None
        if filename is not None:
            if ENV_RANK > 0 and not os.path.isdir(filename):
                stem = filename[:-len(Monitor.EXT) - 1] if filename.endswith("." + Monitor.EXT) else filename
                filename = "%s.env%d" % (stem, ENV_RANK)
            elif ENV_RANK > 0:
                filename = os.path.join(filename, "env%d" % ENV_RANK)
            if not filename.endswith(Monitor.EXT):
                filename = os.path.join(filename, Monitor.EXT) if os.path.isdir(filename) else filename + "." + Monitor.EXT
            self.file_handler = open(filename, "wt")
            self.file_handler.write("#%s\n" % json.dumps({"t_start": self.t_start,
                                                          "env_id": getattr(getattr(env, "spec", None), "id", None)}))
            self.logger = csv.DictWriter(self.file_handler, fieldnames=("r", "l", "t") + tuple(reset_keywords) + tuple(info_keywords))
            self.logger.writeheader()
            self.file_handler.flush()
        self.reset_keywords, self.info_keywords = reset_keywords, info_keywords
        self.allow_early_resets = allow_early_resets
        self.rewards = None
        self.needs_reset = True
        self.episode_rewards, self.episode_lengths, self.episode_times = [], [], []
        self.total_steps = 0
        self.current_reset_info = {}

    def reset(self, **kwargs):
        if not self.allow_early_resets and not self.needs_reset:
            raise RuntimeError("tried to reset an environment before done")
        self.rewards = []
        self.needs_reset = False
        for k in self.reset_keywords:
            if k not in kwargs:
                raise ValueError("expected reset keyword %s" % k)
            self.current_reset_info[k] = kwargs[k]
        return self.env.reset(**kwargs)

    def step(self, action):
        if self.needs_reset:
            raise RuntimeError("tried to step an environment that needs reset")
        obs, rew, done, info = self.env.step(action)
        self.rewards.append(rew)
        if done:
            self.needs_reset = True
            ep_rew, ep_len = sum(self.rewards), len(self.rewards)
            ep_info = {"r": round(float(ep_rew), 6), "l": ep_len, "t": round(time.time() - self.t_start, 6)}
            for k in self.info_keywords:
                ep_info[k] = info[k]
            self.episode_rewards.append(ep_rew)
            self.episode_lengths.append(ep_len)
            self.episode_times.append(time.time() - self.t_start)
            ep_info.update(self.current_reset_info)
            if self.logger:
                self.logger.writerow(ep_info)
                self.file_handler.flush()
            info = dict(info)
            info["episode"] = ep_info
        self.total_steps += 1
        return obs, rew, done, info

    def close(self):
        if self.file_handler is not None:
            self.file_handler.close()
        if hasattr(self.env, "close"):
            self.env.close()

    def get_total_steps(self):
        return self.total_steps

    def get_episode_rewards(self):
        return self.episode_rewards

    def get_episode_lengths(self):
        return self.episode_lengths

    def __getattr__(self, name):            # gym.Wrapper behaviour: forward everything else
        if name.startswith("__") or name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)
