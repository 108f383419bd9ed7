"""In-graph timeline marks (diagnostic): `mark(name)` enqueues a one-lane kernel that stores the GPU wall
clock on the CURRENT stream -- inside hipGraph capture it becomes a node of whichever branch is being
captured, so after a replay the slots hold the true concurrent timeline of both branches (a profiler
serialises them).  Disabled (a dictionary lookup per call site) unless `enable(device)` was called;
tools/branch_timeline.py is the user.
"""
import ctypes

import torch

from . import _lib

_active = None


class Timeline:
    def __init__(self, device, capacity=512):
        self.device = torch.device(device)
        self.slots = torch.zeros(capacity, dtype=torch.int64, device=self.device)
        self.names = {}

    def mark(self, name):
        i = self.names.setdefault(name, len(self.names))
        if i >= self.slots.numel():
            raise RuntimeError("timeline is full")
        with torch.cuda.device(self.device):
            _lib.call("sig3d_timestamp", ctypes.c_void_p(self.slots.data_ptr() + 8 * i), _lib.stream_ptr(self.device))

    def read(self):
        """[(name, microseconds since the earliest mark)] in time order (after a synchronize)."""
        hz = ctypes.c_longlong(0)
        _lib.call("sig3d_timestamp_rate", self.device.index or 0, ctypes.byref(hz))
        t = self.slots[:len(self.names)].cpu().tolist()
        t0 = min(t)
        out = [(n, (t[i] - t0) * 1e6 / hz.value) for n, i in self.names.items()]
        return sorted(out, key=lambda kv: kv[1])


def enable(device):
    global _active
    _active = Timeline(device)
    return _active


def disable():
    global _active
    _active = None


def mark(name):
    if _active is not None:
        _active.mark(name)
