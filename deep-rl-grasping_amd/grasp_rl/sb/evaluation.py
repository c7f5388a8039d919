"""``evaluate_policy`` with the keyword set the reference passes (base_callbacks.py:84-88)."""
import numpy as np


def evaluate_policy(model, env, n_eval_episodes=10, deterministic=True, render=False, callback=None,
                    reward_threshold=None, return_episode_rewards=False):
    if getattr(env, "num_envs", 1) != 1:
        raise ValueError("evaluate_policy expects a single environment")
    episode_rewards, episode_lengths = [], []
    for _ in range(n_eval_episodes):
        obs = env.reset()
        done, state = False, None
        ep_rew, ep_len = 0.0, 0
        while not done:
            action, state = model.predict(obs, state=state, deterministic=deterministic)
            obs, reward, done_, _info = env.step(action)
            done = bool(np.asarray(done_).reshape(-1)[0])
            ep_rew += float(np.asarray(reward).reshape(-1)[0])
            ep_len += 1
            if callback is not None:
                callback(locals(), globals())
            if render:
                env.render()
        episode_rewards.append(ep_rew)
        episode_lengths.append(ep_len)
    mean_reward, std_reward = float(np.mean(episode_rewards)), float(np.std(episode_rewards))
    if reward_threshold is not None and mean_reward <= reward_threshold:
        raise AssertionError("mean reward %.2f below threshold %.2f" % (mean_reward, reward_threshold))
    if return_episode_rewards:
        return episode_rewards, episode_lengths
    return mean_reward, std_reward
