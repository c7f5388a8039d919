"""TEST-ONLY memory backend: host (NumPy) arenas for the g++ emulation build of the engine."""
import numpy as np


class NumpyHostBackend:
    def alloc_f32(self, nbytes):
        return np.zeros((nbytes + 3) // 4, np.float32)

    def ptr(self, a):
        return a.ctypes.data

    def stream_ptr(self):
        return 0

    def to_device(self, arr):
        return np.ascontiguousarray(arr).copy()

    def to_host(self, a):
        return np.array(a, copy=True)

    def write(self, view, arr):
        view[...] = np.asarray(arr, np.float32).reshape(view.shape)

    def synchronize(self):
        pass

    def stream_context(self):
        import contextlib
        return contextlib.nullcontext()

    def comm_fork(self):
        return None

    def comm_context(self, fork=None):
        import contextlib
        return contextlib.nullcontext()

    def comm_join(self):
        pass

    def as_torch(self, a):
        import torch
        return torch.from_numpy(a)      # shares memory with the arena: collectives act in place
