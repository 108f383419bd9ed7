"""Tune the library GEMMs of BASELINE.json's training step on this MI355X and write the winners to
situation3d_amd/tuning/gemm_gfx950.csv (see situation3d_amd/gemm_tuning.py).

python tools/tune_gemms.py [out.csv]      # ~30 s; run through gpurun, copy the file back
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from situation3d_amd import gemm_tuning  # noqa: E402
from situation3d_amd.model import SIG3DQFormer  # noqa: E402
from situation3d_amd.trainer import build_optimizer, train_step  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else gemm_tuning.RESULTS
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
# SIG3D_TUNE_ROTATE_MB: time the candidates on operands rotated through that much memory (cold caches)
rot = os.environ.get("SIG3D_TUNE_ROTATE_MB")
gemm_tuning.enable(tune_missing=True, results=None, rotating_buffer_mb=int(rot) if rot else None)   # tune from scratch
import torch.cuda.tunable as _t  # noqa: E402
print("rotating buffer: %s MB" % _t.get_rotating_buffer_size())
torch.manual_seed(1234)
with torch.cuda.stream(torch.cuda.Stream(device)):
    model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(device).train()
    optimizer = build_optimizer(model, name="flat_adamw")
    for i in range(3):   # eager steps meet every GEMM shape of forward and backward
        train_step(model, optimizer, bench.synthetic_batch(bench.BATCH, bench.N_POINTS, i, device))
    torch.cuda.synchronize()
import torch.cuda.tunable as tunable  # noqa: E402
print("tuned %d GEMM shapes -> %s" % (len(tunable.get_results()), gemm_tuning.save(out)))
