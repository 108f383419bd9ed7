"""HIP streams confined to a subset of the compute units (sig3d_stream_create_with_cu_mask, csrc/capi.hip).

Mask bit k addresses XCD k % 8, CU slot k / 8 of that XCD (MI355X: 8 XCDs x 32 CUs; tools/probes/cu_mask_probe.py
prints the map measured on the box).  `cu_mask(per_xcd_lo, per_xcd_hi)` selects CU slots [lo, hi) of EVERY XCD."""
import ctypes
import time

import torch

from . import _lib

XCDS, CUS_PER_XCD = 8, 32


def cu_mask(lo, hi):
    """32-bit words of the mask with CU slots lo..hi-1 of every XCD set."""
    assert 0 <= lo < hi <= CUS_PER_XCD
    bits = 0
    for slot in range(lo, hi):
        for x in range(XCDS):
            bits |= 1 << (slot * XCDS + x)
    n = XCDS * CUS_PER_XCD // 32
    return [(bits >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


class MaskedStream:
    """Owns a CU-masked HIP stream; `.stream` is the torch view of it (torch.cuda.ExternalStream)."""

    def __init__(self, device, words):
        self.device = torch.device(device)
        arr = (ctypes.c_uint32 * len(words))(*words)
        out = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.call("sig3d_stream_create_with_cu_mask", len(words), ctypes.cast(arr, ctypes.c_void_p), ctypes.byref(out))
        self.handle = out.value
        self.stream = torch.cuda.ExternalStream(self.handle, device=self.device)

    def close(self):
        if self.handle:
            with torch.cuda.device(self.device):
                _lib.call("sig3d_stream_destroy", ctypes.c_void_p(self.handle))
            self.handle = None


def run_concurrently(a, b, device, hold_us=200):
    """Do kernels on streams `a` and `b` overlap in time?  HIP multiplexes its streams onto a few hardware queues
    (GPU_MAX_HW_QUEUES, 4 per priority): two streams that landed on the SAME queue are served in order, however
    independent their work is.  Measured: one wave that sleeps for `hold_us` on each stream (sig3d_queue_hold);
    ~hold_us in total = concurrent, ~2 x hold_us = one queue.  Synchronises the device (construction-time check)."""
    dev = torch.device(device)
    best = None
    for _ in range(3):                      # first pass warms the kernel and the queues up; the fastest counts
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for st in (a, b):
            with torch.cuda.stream(st):
                _lib.call("sig3d_queue_hold", int(hold_us), _lib.stream_ptr(dev))
        torch.cuda.synchronize(dev)
        t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    return best < 1.6e-6 * hold_us


def stream_beside(others, device, priority=0, tries=12):
    """A new torch stream that runs concurrently with every stream in `others` (see run_concurrently); falls back
    to the last candidate when none of `tries` does (the chains then serialise: slower, never wrong)."""
    cand = None
    for _ in range(tries):
        cand = torch.cuda.Stream(device, priority=priority)
        if all(run_concurrently(o, cand, device) for o in others):
            return cand, True
    return cand, False
