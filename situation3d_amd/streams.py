"""HIP streams confined to a subset of the compute units (sig3d_stream_create_with_cu_mask, csrc/capi.hip).

Mask bit k addresses XCD k % 8, CU slot k / 8 of that XCD (MI355X: 8 XCDs x 32 CUs; tools/probes/cu_mask_probe.py
prints the map measured on the box).  `cu_mask(per_xcd_lo, per_xcd_hi)` selects CU slots [lo, hi) of EVERY XCD."""
import ctypes

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
