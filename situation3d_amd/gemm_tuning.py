"""Library-GEMM solution selection for the Q-Former / SharedMLP matrix products.

The dense layers of the path (Qformer.py:164-178,242,311,324 nn.Linear; the 1x1 convolutions of
the small SA levels) are plain row-major f32 GEMMs with M = batch * tokens = 160..416 rows -- far
too few rows for hipBLASLt's default heuristic, which picks 96x32 / 32x64 macro tiles at ~20 us
per call.  PyTorch's TunableOp times every rocBLAS / hipBLASLt solution per (shape, layout) once
and replays the winner: -1.0 ms per training step on MI355X (16.8 -> 15.8 ms).

`tuning/gemm_gfx950.csv` holds the winners for BASELINE.json's configuration (B=8, 40k points,
32 queries + 20 question tokens), produced by tools/tune_gemms.py on an MI355X; the file carries
validators (torch / hipBLASLt / rocBLAS versions, gfx arch) and TunableOp ignores it when they do
not match, falling back to tuning the shapes it meets during the eager warm-up steps (~15 s).
Training / bench processes never write the committed file (ranks would race on it): TunableOp's
own output goes to a per-process scratch file.
"""
import os
import tempfile

import torch

RESULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "gemm_gfx950.csv")
_state = {"enabled": False}


def enable(tune_missing=True, results=RESULTS, rotating_buffer_mb=None):
    """Turn TunableOp on for this process.  Call before the first GEMM, after the device is set.
    `rotating_buffer_mb`: candidates are timed on operands rotated through this much memory (cold caches, like the
    weights of a step that streams 700 MB of them); None keeps TunableOp's default (one L2's worth)."""
    if results == RESULTS and os.environ.get("SIG3D_GEMM_TABLE"):
        results = os.environ["SIG3D_GEMM_TABLE"]
    if not torch.cuda.is_available():
        raise RuntimeError("GEMM tuning needs the GPU (there is no CPU path)")
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.set_max_tuning_duration(30)     # ms per candidate
    tunable.set_max_tuning_iterations(20)
    if rotating_buffer_mb is not None:
        tunable.set_rotating_buffer_size(int(rotating_buffer_mb))
    # TunableOp streams what it tunes to its file: point that at scratch, read the committed winners
    scratch = os.path.join(tempfile.gettempdir(), "sig3d_tunableop_%d.csv" % os.getpid())
    tunable.set_filename(scratch, insert_device_ordinal=False)
    if results and os.path.exists(results):
        tunable.read_file(results)
    tunable.tuning_enable(bool(tune_missing))
    _state["enabled"] = True
    return len(tunable.get_results())


def is_enabled():
    return _state["enabled"]


class no_tuning:
    """Context: replay known winners only (stream capture cannot time candidates)."""

    def __enter__(self):
        self.was = False
        if _state["enabled"]:
            import torch.cuda.tunable as tunable
            self.was = tunable.tuning_is_enabled()
            tunable.tuning_enable(False)
        return self

    def __exit__(self, *exc):
        if _state["enabled"] and self.was:
            import torch.cuda.tunable as tunable
            tunable.tuning_enable(True)
        return False


def save(path=RESULTS):
    """Write validators + the in-memory winners in TunableOp's CSV format (tools/tune_gemms.py)."""
    import torch.cuda.tunable as tunable
    with open(path, "w") as f:
        for key, val in tunable.get_validators():
            f.write("Validator,%s,%s\n" % (key, val))
        for op, params, solution, ms in sorted(tunable.get_results()):
            f.write("%s,%s,%s,%s\n" % (op, params, solution, ms))
    return path
