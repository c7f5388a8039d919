import numpy as np


class NormalActionNoise:
    def __init__(self, mean, sigma):
        self._mu, self._sigma = mean, sigma

    def __call__(self):
        return np.random.normal(self._mu, self._sigma)

    def reset(self):
        pass


class OrnsteinUhlenbeckActionNoise:
    def __init__(self, mean, sigma, theta=0.15, dt=1e-2, initial_noise=None):
        self._theta, self._mu, self._sigma, self._dt = theta, mean, sigma, dt
        self.initial_noise = initial_noise
        self.reset()

    def __call__(self):
        n = self.noise_prev + self._theta * (self._mu - self.noise_prev) * self._dt \
            + self._sigma * np.sqrt(self._dt) * np.random.normal(size=self._mu.shape)
        self.noise_prev = n
        return n

    def reset(self):
        self.noise_prev = self.initial_noise if self.initial_noise is not None else np.zeros_like(self._mu)


class AdaptiveParamNoiseSpec:
    def __init__(self, initial_stddev=0.1, desired_action_stddev=0.1, adoption_coefficient=1.01):
        self.current_stddev = initial_stddev
