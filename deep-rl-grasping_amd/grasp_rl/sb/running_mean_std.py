"""Running mean / variance of the VecNormalize wrapper: float64, parallel
(Chan et al.) update from batch moments, count starting at epsilon = 1e-4.

Data-parallel training keeps the statistics of all replicas identical: when ``gather`` is
set (grasp_rl.parallel.share_running_stats) every update exchanges the batch moments of all ranks and
merges them in rank order -- the same Chan merge, applied world_size times."""
import numpy as np


class RunningMeanStd:
    gather = None      # optional callable (mean, var, count) -> [(mean, var, count) of rank 0, rank 1, ...]

    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon

    def __getstate__(self):                       # the exchange hook is process-local, never pickled
        state = self.__dict__.copy()
        state.pop("gather", None)
        return state

    def update(self, arr):
        arr = np.asarray(arr, np.float64)
        # batch moments exactly as ndarray.mean / ndarray.var form them (sum / n; mean of squared deviations), with
        # the mean computed once and the deviations squared in place: same bits, one pass and two temporaries less
        mean = arr.mean(axis=0)
        dev = arr - mean
        np.multiply(dev, dev, out=dev)
        moments = [(mean, dev.mean(axis=0), arr.shape[0])]
        if self.gather is not None:
            moments = self.gather(*moments[0])
        for m in moments:
            self.update_from_moments(*m)

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot
        m2 = self.var * self.count + batch_var * batch_count + np.square(delta) * self.count * batch_count / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot
