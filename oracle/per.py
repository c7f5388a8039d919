"""TEST INFRASTRUCTURE: CPU restatement of stable-baselines 2.10.1 ``PrioritizedReplayBuffer`` (proportional
prioritisation; the reference enables it with ``prioritized_replay: True``,
/root/reference/config/gripper_grasp.yaml:102, simplified_object_picking.yaml:101,110, and passes
``prioritized_replay`` to ``sb.DQN`` / ``sb.BDQ`` at manipulation_main/training/sb_helper.py:159-165,210-224).
The implementation lives in the un-vendored dependency (stable_baselines/common/buffers.py +
segment_tree.py, v2.10.1, setup.py:7): **parity unpinned** by reference tests; restated from its
published algorithm:

    add:     it_sum[i] = it_min[i] = max_priority ** alpha
    sample:  every_range_len = it_sum.sum() / batch_size
             mass_k = random() * every_range_len + k * every_range_len
             idx_k  = it_sum.find_prefixsum_idx(mass_k)
             p_min  = it_min.min() / it_sum.sum();  max_weight = (p_min * N) ** -beta
             w_k    = (it_sum[idx_k] / it_sum.sum() * N) ** -beta / max_weight
    update:  it_sum[i] = it_min[i] = priority ** alpha;  max_priority = max(max_priority, priority)

``it_sum`` is a binary segment tree over ``it_capacity`` = the next power of two >= the buffer size: every
internal node is ``left + right`` in float64, and ``find_prefixsum_idx`` walks down from the root (left child
if its sum exceeds the remaining mass, otherwise subtract it and go right).  Both the association order of the
sums and the subtractions of the walk are restated here exactly (vectorised by tree level), because they
decide which index owns a mass that falls within rounding of an interval boundary: the device sampler
(csrc/per_kernels.h) performs the same float64 operations in the same order, so on identical stored
priorities and identical uniforms its indices are *bit-identical* to this module's.

Only ``tests/`` may import this module.  The uniforms are explicit inputs so that the device sampler can be
checked draw for draw.
"""
import numpy as np


class PerOracle:
    def __init__(self, capacity, alpha=0.6, eps=1e-6):
        self.cap, self.alpha, self.eps = int(capacity), float(alpha), float(eps)
        self.p = np.zeros(self.cap, np.float32)       # priority ** alpha, float32 like the device array
        self.size, self.pos = 0, 0
        self.max_priority = np.float32(1.0)

    def add(self, n=1):
        for _ in range(n):
            self.p[self.pos] = np.float32(self.max_priority) ** np.float32(self.alpha)
            self.pos = (self.pos + 1) % self.cap
            self.size = min(self.cap, self.size + 1)

    def tree_levels(self):
        """Node sums of the segment tree, one float64 array per level (leaves first, root last)."""
        cap2 = 1
        while cap2 < self.cap:
            cap2 *= 2
        leaves = np.zeros(cap2, np.float64)
        leaves[:self.size] = self.p[:self.size].astype(np.float64)
        levels = [leaves]
        while len(levels[-1]) > 1:
            a = levels[-1]
            levels.append(a[0::2] + a[1::2])          # node = left + right
        return levels

    def sample(self, u, beta):
        """u: [B] uniforms in [0,1).  Returns (idx [B] int64, weights [B] float32, mass [B], prefix [size+1]).
        ``prefix`` (sequential float64 cumulative sums) is informational: the indices come from the tree walk."""
        B = len(u)
        levels = self.tree_levels()
        total = float(levels[-1][0])
        mass = (np.asarray(u, np.float64) + np.arange(B)) * total / B
        node = np.zeros(B, np.int64)
        rem = mass.copy()
        for lvl in range(len(levels) - 2, -1, -1):     # children of `node` live in levels[lvl]
            left = levels[lvl][2 * node]
            go_left = left > rem
            rem = np.where(go_left, rem, rem - left)
            node = 2 * node + np.where(go_left, 0, 1)
        idx = np.minimum(node, self.size - 1).astype(np.int64)
        p = self.p[:self.size].astype(np.float64)
        prefix = np.concatenate([[0.0], np.cumsum(p)])
        p_min = p.min() / total
        w = (p[idx] / total * self.size) ** (-beta) / (p_min * self.size) ** (-beta)
        return idx, w.astype(np.float32), mass, prefix

    def update(self, idx, priorities):
        for i, pr in zip(idx, priorities):
            pr = np.float32(pr) + np.float32(self.eps)
            self.p[i] = pr ** np.float32(self.alpha)
            self.max_priority = max(self.max_priority, pr)
