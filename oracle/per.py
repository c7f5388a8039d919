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
             idx_k  = it_sum.find_prefixsum_idx(mass_k)   (largest i with sum(p[:i]) <= mass)
             p_min  = it_min.min() / it_sum.sum();  max_weight = (p_min * N) ** -beta
             w_k    = (it_sum[idx_k] / it_sum.sum() * N) ** -beta / max_weight
    update:  it_sum[i] = it_min[i] = priority ** alpha;  max_priority = max(max_priority, priority)

Only ``tests/`` may import this module.  The uniforms are explicit inputs so that the device sampler
(csrc/per_kernels.h) can be checked draw for draw.
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

    def sample(self, u, beta):
        """u: [B] uniforms in [0,1).  Returns (idx [B] int64, weights [B] float32, mass [B], prefix [size+1])."""
        B = len(u)
        p = self.p[:self.size].astype(np.float64)
        prefix = np.concatenate([[0.0], np.cumsum(p)])
        total = prefix[-1]
        mass = (np.asarray(u, np.float64) + np.arange(B)) * total / B
        idx = np.minimum(np.searchsorted(prefix, mass, side="right") - 1, self.size - 1).astype(np.int64)
        p_min = p.min() / total
        w = (p[idx] / total * self.size) ** (-beta) / (p_min * self.size) ** (-beta)
        return idx, w.astype(np.float32), mass, prefix

    def update(self, idx, priorities):
        for i, pr in zip(idx, priorities):
            pr = np.float32(pr) + np.float32(self.eps)
            self.p[i] = pr ** np.float32(self.alpha)
            self.max_priority = max(self.max_priority, pr)
