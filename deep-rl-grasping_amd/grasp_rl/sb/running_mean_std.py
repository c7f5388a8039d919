"""Running mean / variance of the VecNormalize wrapper, restating stable-baselines 2.10.1
``common/running_mean_std.py``: ``update(arr)`` forms ``np.mean(arr, axis=0)`` / ``np.var(arr, axis=0)`` IN THE DTYPE OF
``arr`` -- float32 for the observations DummyVecEnv hands over (its buffers carry the observation space's dtype; the
reference's env builds float64 images with np.dstack, robot.py:183-205, and the Box is float32, robot.py:207-228),
float64 for the discounted returns -- and merges them into the float64 running moments with the parallel (Chan et
al.) update, count starting at epsilon = 1e-4.  The device keeps the same statistics with the same arithmetic
(csrc/elem_kernels.h: norm_update_kernel).

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
        arr = np.asarray(arr)
        if arr.dtype.kind != "f":
            arr = arr.astype(np.float64)
        moments = [(np.mean(arr, axis=0), np.var(arr, axis=0), arr.shape[0])]
        if self.gather is not None:
            moments = self.gather(*moments[0])
        for m in moments:
            self.update_from_moments(*m)

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot_count = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot_count
        m_a = self.var * self.count
        m_b = batch_var * batch_count
        m_2 = m_a + m_b + np.square(delta) * self.count * batch_count / (self.count + batch_count)
        new_var = m_2 / (self.count + batch_count)
        self.mean, self.var, self.count = np.asarray(new_mean, np.float64), np.asarray(new_var, np.float64), tot_count
