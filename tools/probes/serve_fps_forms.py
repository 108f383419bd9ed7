"""bench.forward_only_variant (BASELINE config 2: B = 4 forward, one batch alone and two geometry chains in flight) with the
chains' FPS as the cooperative kernel / the block-list kernel with 16, 8, 4 waves, alternating in one process."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from situation3d_amd import gemm_tuning, geometry
from situation3d_amd.pointnet2 import _ext
dev = torch.device("cuda", 0)
WAVES = [16]
_init = geometry.GeometryPipeline.__init__


def _init_with_waves(self, *args, **kwargs):     # serve.py asks for 16 waves; the probe chooses
    kwargs["fps_waves"] = WAVES[0]
    _init(self, *args, **kwargs)


geometry.GeometryPipeline.__init__ = _init_with_waves
torch.cuda.set_device(dev)
gemm_tuning.enable(tune_missing=True)
for rep in range(3):
    for name, blocks, waves in (("coop", False, 16), ("blocks-16", True, 16), ("blocks-8", True, 8), ("blocks-4", True, 4)):
        _ext.FPS_BLOCKS, _ext.FPS_WAVES = blocks, waves
        WAVES[0] = waves
        v = bench.forward_only_variant(dev)
        print("%-9s single %.3f ms  pipelined %.3f ms per batch (%.0f samples/s)" % (name, v["single_batch_latency_ms"], v["pipelined_ms_per_batch"], v["value"]), flush=True)
