"""Many small device-to-device copies as ONE launch.

A hipGraph replay of the training step ends by handing the freshly computed geometry plan to the next
step (geometry.GeometryPlan.copy_from: 22 tensors) and starts by refreshing its static input buffers (7
tensors).  As separate `copy_` calls those are 29 copy kernels of ~4 us each on the critical path; here
the (source, destination, length) triples live in a device table that `sig3d_gather_table` (csrc/optim.hip:
one workgroup per record of <= 65536 four-byte words) walks in a single launch.  Tables are cached by the
addresses involved, so a loader that cycles through a few staging buffers builds each table once.
"""
import numpy as np
import torch

from . import _lib

_REC = np.dtype([("p", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("n", "i8"), ("wd", "f4"), ("pad", "f4")])
_CHUNK = 65536


class TableCopy:
    def __init__(self, device):
        self.device = device
        self._tables = {}

    def _build(self, pairs):
        recs = []
        for dst, src in pairs:
            assert dst.is_contiguous() and src.is_contiguous() and dst.numel() * dst.element_size() == \
                src.numel() * src.element_size(), "table copies are flat byte copies"
            nbytes = dst.numel() * dst.element_size()
            assert nbytes % 4 == 0 and dst.data_ptr() % 4 == 0 and src.data_ptr() % 4 == 0
            words = nbytes // 4
            for w0 in range(0, words, _CHUNK):
                recs.append((0, src.data_ptr() + 4 * w0, dst.data_ptr() + 4 * w0, 0, min(_CHUNK, words - w0), 0.0, 0.0))
        host = torch.empty(max(len(recs), 1) * _REC.itemsize, dtype=torch.uint8).pin_memory()
        host.numpy().view(_REC)[:len(recs)] = np.array(recs, dtype=_REC)
        table = torch.empty(host.numel(), dtype=torch.uint8, device=self.device)
        table.copy_(host, non_blocking=torch.cuda.is_current_stream_capturing())
        return table, len(recs), host   # the pinned staging buffer stays alive (a captured upload re-reads it)

    def __call__(self, pairs):
        """pairs: [(dst, src), ...] contiguous device tensors of equal byte size (any dtype)."""
        key = tuple((d.data_ptr(), s.data_ptr(), d.numel() * d.element_size()) for d, s in pairs)
        hit = self._tables.get(key)
        if hit is None:
            if len(self._tables) > 64:
                self._tables.clear()
            hit = self._tables[key] = self._build(pairs)
        table, n, _ = hit
        if n:
            with torch.cuda.device(self.device):
                _lib.call("sig3d_gather_table", n, _lib.ptr(table), _lib.stream_ptr(self.device))
