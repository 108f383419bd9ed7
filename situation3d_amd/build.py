"""Build recipe for libsig3d_hip.so (gfx950 only, plain hipcc, no torch headers).

`python -m situation3d_amd.build` compiles every .hip file under csrc/ into one shared
library next to this file.  hipcc cross-compiles for gfx950 without a GPU, so this runs in
the build container and the .so travels with the tree to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsig3d_hip.so")
SOURCES = ["capi.hip", "sampling.hip", "ball_query.hip", "group_points.hip", "interpolate.hip",
           "situational.hip", "attention.hip", "shared_mlp.hip", "shared_mlp_fwd1.hip", "shared_mlp_fwd2.hip", "shared_mlp_fwd4.hip", "shared_mlp_fwd4r.hip", "pos_embed.hip", "rowops.hip", "optim.hip", "voxelize.hip", "compact.hip", "gemm16.hip", "gemmp.hip", "sqa_loss.hip", "sa_first.hip", "small_mlp.hip", "qformer_embed.hip", "heads.hip"]
HEADERS = [os.path.join(CSRC, "sig3d_common.h"), os.path.join(CSRC, "shared_mlp_fwd.h"), os.path.join(CSRC, "gemm16_core.h"), os.path.join(CSRC, "gemmp_core.h"), os.path.join(CSRC, "situational_pose.h"),
           os.path.join(HERE, "..", "include", "sig3d_hip.h"), os.path.join(HERE, "..", "include", "sig3d_debug.h")]
# -ffp-contract=off: distances are spelled with _rn intrinsics already; this keeps every other
# f32 expression in the point ops unfused too (parity contract, see oracle/pointnet2_oracle.c).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    # longest compile first (measured, seconds on one core of the build container; the rest by size)
    cost = {"shared_mlp_fwd4r.hip": 44, "gemmp.hip": 33, "shared_mlp_fwd2.hip": 30, "gemm16.hip": 26,
            "shared_mlp_fwd4.hip": 21, "shared_mlp_fwd1.hip": 18, "shared_mlp.hip": 13}
    for src in sorted(SOURCES, key=lambda f: -cost.get(f, os.path.getsize(os.path.join(CSRC, f)) / 2e4)):
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            jobs.append([hipcc, "-x", "hip", "-c", s, "-o", o] + FLAGS)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        # the longest translation unit (shared_mlp_fwd4r.hip, ~45 s alone) bounds the wall time once every file has a core
        with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
