"""bench.py end to end on the GPU box: the contract line of the default mode, and the data-parallel code
path driven by a REAL RCCL process group of size 1 (SIG3D_SINGLE_RANK_PG=1: all-reduce kernels, graphs
cut around collectives, bucketed AdamW -- everything the 8-GPU run executes except the wire)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_args, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline"] + extra_args, env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return json.loads(lines[-1])   # the JSON line is the LAST line of stdout


def test_default_line_carries_the_contract():
    out = _run([], {})
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["scaling"] == "weak" and out["value"] > 0
    r = out["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "ball_query" in r["kernel"] and "query_group" in r["kernel"]          # the pair the north star names
    assert out["fps_timeouts"] == 0
    # beside the headline: every level dense, and surface-shaped scenes (both slower or equal, both finite)
    v = out["variants"]
    steps = {k: x for k, x in v.items() if "ms_per_step" in x}
    assert len(steps) == 2 and all(x["ms_per_step"] > 0 for x in steps.values())
    dense = [x for k, x in v.items() if k.startswith("dense")][0]
    assert dense["compact_levels"] == {} and out["compact_levels"]              # headline compact, variant dense
    fwd = v["config 2: forward only, B=4"]                                       # BASELINE config 2, driver-timed
    assert fwd["batch"] == 4 and 0 < fwd["pipelined_ms_per_batch"] <= fwd["single_batch_latency_ms"] * 1.05
    # the pair's own launches, its floor model, PMC traffic and the op-level pair beside it
    assert r["launches_per_step"] <= 4 and r["floor_model"]["floor_ms"] > 0 and r["traffic"] > 0
    assert "sig3d_transpose_cn" not in r["parts"]                                # no transpose launch left in the step
    ops = out["roofline_ops"]
    assert ops["bound"] == "hbm" and 0 < ops["frac"] < 1 and ops["algorithmic_bytes"] > 3e8
    assert out["rccl_ranks"] == 1 and out["dist_backend"] is None


@pytest.mark.parametrize("comm", ["pg", "own"])
def test_data_parallel_path_over_real_rccl_group_of_one(comm):
    """comm = "own": the buckets' all-reduces as synchronous ops on the process's own communication stream behind the
    ticket handshake (SIG3D_DDP_COMM=own, ddp.GradBucketReducer._launch_all_on_own_stream)."""
    env = {"SIG3D_SINGLE_RANK_PG": "1", "MASTER_PORT": "29533" if comm == "pg" else "29535"}
    if comm == "own":
        env["SIG3D_DDP_COMM"] = "own"
    out = _run(["--force-reducer", "--no-variants"], env)
    assert out["value"] > 0 and out["n_gpus"] == 1
    # the exchange explains itself (ddp.CommStats): ~614 MB of buckets + the embedding rows, time spent waiting
    c = out["comm"]
    assert c["bytes_per_step"] > 500e6 and c["buckets"] >= 10
    assert c["exposed_ms"] >= 0 and c["comm_window_ms"] > 0 and c["overlap_frac"] is not None
    ref = _run(["--no-variants"], {})
    # same seeds, same batches, mean over one rank == identity: the loss after 6 steps must agree
    assert abs(out["final_loss"] - ref["final_loss"]) <= 2e-3 * abs(ref["final_loss"])


def test_bare_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent starts two ranks itself
    (reference launch: slurm_3dllm_run.slurm:30).  One-GPU box: both ranks share GPU 0 and the collectives run
    over gloo -- the same graphs-around-collectives step that RCCL drives on the node."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"SIG3D_DIST_BACKEND": "gloo", "SIG3D_SHARE_GPU": "1"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dist_backend"] == "gloo"
    assert out["config"]["global_batch"] == 16 and out["value"] > 0
    assert out["comm"]["bytes_per_step"] > 500e6 and out["comm"]["buckets"] >= 10 and out["comm"]["exposed_ms"] >= 0


def test_probe_switches_are_refused():
    """A SIG3D_PROBE_* variable in the environment (tools/probes/geo_probes.py) must not produce a bench line."""
    env = dict(os.environ, SIG3D_PROBE_SKIP_CHAIN="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "SIG3D_PROBE_SKIP_CHAIN" in (p.stderr + p.stdout)
