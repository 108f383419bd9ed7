"""Which CUs does a CU-masked stream run on?  sig3d_whereami stores {HW_ID, XCC_ID} per workgroup."""
import collections, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from situation3d_amd import _lib, streams
dev = torch.device("cuda", 0)


def where(stream, blocks=1024, threads=512, hold_us=200):
    slots = torch.zeros(2 * blocks, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        _lib.call("sig3d_whereami", _lib.ptr(slots), blocks, threads, hold_us, _lib.stream_ptr(dev))
    torch.cuda.synchronize()
    v = slots.cpu().view(blocks, 2).numpy().astype("uint32")
    hw, xcc = v[:, 0], v[:, 1] & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0x7
    return collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))


plain = where(torch.cuda.Stream(dev))
print("unmasked stream: %d distinct (xcc, se, sh, cu); per XCD %s" %
      (len(plain), sorted(collections.Counter(k[0] for k in plain).items())))
for lo, hi in ((0, 1), (0, 4), (4, 32), (28, 32)):
    ms = streams.MaskedStream(dev, streams.cu_mask(lo, hi))
    c = where(ms.stream)
    print("mask slots [%d, %d) of every XCD: %d distinct CUs; per XCD %s" %
          (lo, hi, len(c), sorted(collections.Counter(k[0] for k in c).items())))
    if hi - lo <= 4:
        print("   ", sorted(c))
    overlap = None
    ms.close()
a = streams.MaskedStream(dev, streams.cu_mask(0, 4)); b = streams.MaskedStream(dev, streams.cu_mask(4, 32))
ca, cb = where(a.stream), where(b.stream)
print("slots [0,4) and [4,32): %d + %d CUs, %d shared" % (len(ca), len(cb), len(set(ca) & set(cb))))
