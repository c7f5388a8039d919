"""TEST INFRASTRUCTURE: CPU restatement of stable-baselines ``PrioritizedReplayBuffer`` + ``SegmentTree``
(proportional prioritisation; the reference enables it with ``prioritized_replay: True``,
/root/reference/config/gripper_grasp.yaml:102, simplified_object_picking.yaml:101,110, and passes
``prioritized_replay`` to ``sb.DQN`` / ``sb.BDQ`` at manipulation_main/training/sb_helper.py:159-165,210-224).
The implementation lives in the un-vendored dependency ``stable-baselines==2.10.1`` (setup.py:7-12; files
stable_baselines/common/buffers.py + segment_tree.py), absent from /root/reference and not installable here:
**parity unpinned** by reference tests.  This module restates the published code operation by operation --
the classes below keep its names, argument meaning and the order of every floating-point operation:

  SegmentTree       ``_value`` = float64 array of 2 * capacity nodes (capacity = power of two >= buffer size),
                    leaf i at ``capacity + i``, node = operation(left, right); ``reduce(start, end)`` via the recursive
                    ``_reduce_helper`` (so a prefix range is summed as ``left + (next_left + (...))``, right-nested along
                    the path to its last leaf); ``__setitem__`` on an index ARRAY (2.10.0, "Parallelized updating and
                    sampling from the replay buffer in DQN"): leaves written, then level by level
                    ``_value[idxs] = operation(_value[2 * idxs], _value[2 * idxs + 1])`` over ``np.unique(idxs // 2)``
  SumSegmentTree    ``sum`` = reduce; ``find_prefixsum_idx(prefixsum)``: from the root, ``left > prefixsum`` -> left,
                    otherwise subtract ``left`` and go right
  MinSegmentTree    ``min`` = reduce with ``min``
  PrioritizedReplayBuffer
    add:            it_sum[idx] = it_min[idx] = max_priority ** alpha
    _sample_proportional (2.10.x):   total = it_sum.sum(0, len(storage) - 1)        # (*)
                                     mass  = np.random.random(size=batch_size) * total
                                     idx   = it_sum.find_prefixsum_idx(mass)
      (baselines / stable-baselines < 2.10, ``stratified=True`` here):
                                     every_range_len = total / batch_size
                                     mass_k = random.random() * every_range_len + k * every_range_len
    sample:         p_min = it_min.min() / it_sum.sum();  max_weight = (p_min * N) ** -beta
                    weights = (it_sum[idx] / it_sum.sum() * N) ** -beta / max_weight
    update_priorities(idxes, priorities):  it_sum[idxes] = it_min[idxes] = priorities ** alpha
                                           max_priority = max(max_priority, np.max(priorities))

(*) ``sum(0, end)`` is ``reduce`` with ``end -= 1`` applied: the range is [0, len(storage) - 2] inclusive, i.e. the
published code leaves the LAST stored transition out of the sampled mass (it can never be drawn) while the
importance weights divide by the full ``it_sum.sum()``.  Restated as published.

Dtypes as they arise at the reference's call sites: DQN.learn passes ``np.abs(td_errors) + prioritized_replay_eps``
-- ``td_errors`` is a float32 TF output, so ``priorities`` is float32 and ``priorities ** alpha`` is a FLOAT32 power
(alpha cast to float32 by NumPy) stored into the float64 tree; ``max_priority`` becomes a float32 scalar whose
``** alpha`` at the next ``add`` is a float64 power (scalar with a Python float).  ``PerOracle`` below does exactly
that with NumPy, on its own: the tests never copy device values into it.

The uniforms are explicit float64 inputs (``np.random.random`` is float64) so that the device sampler can be
checked draw for draw.  Only ``tests/`` may import this module.
"""
import math

import numpy as np


class SegmentTree:
    def __init__(self, capacity, operation, neutral_element):
        assert capacity > 0 and capacity & (capacity - 1) == 0, "capacity must be positive and a power of 2."
        self._capacity = capacity
        self._value = np.full(2 * capacity, neutral_element, np.float64)
        self._operation = operation
        self.neutral_element = neutral_element

    def _reduce_helper(self, start, end, node, node_start, node_end):
        if start == node_start and end == node_end:
            return self._value[node]
        mid = (node_start + node_end) // 2
        if end <= mid:
            return self._reduce_helper(start, end, 2 * node, node_start, mid)
        if mid + 1 <= start:
            return self._reduce_helper(start, end, 2 * node + 1, mid + 1, node_end)
        return self._operation(self._reduce_helper(start, mid, 2 * node, node_start, mid),
                               self._reduce_helper(mid + 1, end, 2 * node + 1, mid + 1, node_end))

    def reduce(self, start=0, end=None):
        if end is None:
            end = self._capacity
        if end < 0:
            end += self._capacity
        end -= 1
        return self._reduce_helper(start, end, 1, 0, self._capacity - 1)

    def __setitem__(self, idx, val):
        idxs = idx + self._capacity
        self._value[idxs] = val
        if isinstance(idxs, (int, np.integer)):
            idxs = np.array([idxs])
        idxs = np.unique(idxs // 2)
        while len(idxs) > 1 or idxs[0] > 0:
            self._value[idxs] = self._operation(self._value[2 * idxs], self._value[2 * idxs + 1])
            idxs = np.unique(idxs // 2)

    def __getitem__(self, idx):
        assert np.max(idx) < self._capacity
        assert 0 <= np.min(idx)
        return self._value[self._capacity + idx]


class SumSegmentTree(SegmentTree):
    def __init__(self, capacity):
        super().__init__(capacity, np.add, 0.0)

    def sum(self, start=0, end=None):
        return super().reduce(start, end)

    def find_prefixsum_idx(self, prefixsum):
        prefixsum = np.atleast_1d(np.asarray(prefixsum, np.float64)).copy()
        assert 0 <= np.min(prefixsum)
        assert np.max(prefixsum) <= self.sum() + 1e-5
        idx = np.ones(len(prefixsum), dtype=int)
        cont = np.ones(len(prefixsum), dtype=bool)
        while np.any(cont):                       # while not all nodes are leafs
            idx[cont] = 2 * idx[cont]
            prefixsum_new = np.where(self._value[idx] <= prefixsum, prefixsum - self._value[idx], prefixsum)
            idx = np.where(np.logical_or(self._value[idx] > prefixsum, np.logical_not(cont)), idx, idx + 1)
            prefixsum = prefixsum_new
            cont = idx < self._capacity
        return idx - self._capacity


class MinSegmentTree(SegmentTree):
    def __init__(self, capacity):
        super().__init__(capacity, np.minimum, float("inf"))

    def min(self, start=0, end=None):
        return super().reduce(start, end)


class PerOracle:
    """``PrioritizedReplayBuffer`` without the transition storage (only ``len(storage)`` matters here)."""

    def __init__(self, capacity, alpha=0.6, eps=1e-6, stratified=False):
        self.cap, self._alpha, self.eps, self.stratified = int(capacity), float(alpha), float(eps), bool(stratified)
        it_capacity = 1
        while it_capacity < self.cap:
            it_capacity *= 2
        self._it_sum = SumSegmentTree(it_capacity)
        self._it_min = MinSegmentTree(it_capacity)
        self._max_priority = 1.0
        self.size, self._next_idx = 0, 0

    def add(self, n=1):
        """n consecutive ``add`` calls (no update in between: they all see the same max_priority, so the vectorised
        ``__setitem__`` of 2.10 leaves the same tree as n single calls)."""
        idx = (self._next_idx + np.arange(n)) % self.cap
        self._next_idx = int((self._next_idx + n) % self.cap)
        self.size = min(self.cap, self.size + n)
        # Python float, or (after an update) np.float32 scalar ** Python float: a FLOAT64 power under the NumPy the
        # reference runs on (1.16-1.19, scalar-scalar promotion); written out because NumPy >= 2 would keep float32
        val = np.float64(self._max_priority) ** self._alpha
        self._it_sum[idx] = val
        self._it_min[idx] = val

    @property
    def leaves(self):
        """float64 leaves of the sum tree over the ring [capacity]."""
        return self._it_sum._value[self._it_sum._capacity:self._it_sum._capacity + self.cap]

    def masses(self, u):
        u = np.asarray(u, np.float64)
        assert self.size >= 2, "sum(0, len(storage) - 1) of the published code needs two stored transitions"
        total = self._it_sum.sum(0, self.size - 1)
        if self.stratified:
            every_range_len = total / len(u)
            return u * every_range_len + np.arange(len(u)) * every_range_len
        return u * total

    def sample(self, u, beta):
        """u: [B] float64 uniforms in [0,1).  Returns (idx [B] int64, weights [B] float64)."""
        assert beta > 0
        idxes = self._it_sum.find_prefixsum_idx(self.masses(u))
        p_min = self._it_min.min() / self._it_sum.sum()
        max_weight = (p_min * self.size) ** (-beta)
        p_sample = self._it_sum[idxes] / self._it_sum.sum()
        weights = (p_sample * self.size) ** (-beta) / max_weight
        return idxes.astype(np.int64), weights

    def update(self, idxes, td_abs):
        """DQN.learn: ``new_priorities = np.abs(td_errors) + prioritized_replay_eps`` (float32), then
        ``update_priorities``.  A transition named twice keeps the value of its last occurrence (NumPy fancy
        assignment)."""
        priorities = np.abs(np.asarray(td_abs, np.float32)) + self.eps          # float32 array + Python float
        assert priorities.dtype == np.float32
        idxes = np.asarray(idxes, np.int64)
        assert len(idxes) == len(priorities) and np.min(priorities) > 0
        assert np.min(idxes) >= 0 and np.max(idxes) < self.size
        # `priorities ** self._alpha` with float32 priorities: NumPy casts alpha to float32 and calls libm powf (the
        # reference-era NumPy has no SIMD math library), which glibc >= 2.28 evaluates in float64 and rounds once --
        # correctly rounded in all but ~1e-8 of the cases.  Restated as exactly that, so that this oracle does not
        # depend on which vector math library the NumPy of THIS container dispatches float32 powers to.
        a32 = float(np.float32(self._alpha))
        powed = np.array([np.float32(math.pow(float(x), a32)) for x in priorities], np.float32)
        self._it_sum[idxes] = powed
        self._it_min[idxes] = powed
        self._max_priority = max(self._max_priority, np.max(priorities))
