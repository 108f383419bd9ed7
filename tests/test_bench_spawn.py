"""bench.py's own launcher (no torch.distributed.run): the parent process starts one child per rank, relays
rank 0's JSON line as its last line of stdout and fails when any rank fails.  CPU only: the children are a small
gloo script (the product step needs a GPU), or bench.py itself failing on the missing GPU."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import json, os, sys
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    one = torch.ones(1)
    dist.all_reduce(one)
    if "--fail-rank-1" in sys.argv and rank == 1:
        sys.exit(7)
    print("noise from rank %d" % rank)
    if rank == 0:
        print(json.dumps({"ranks": int(one.item()), "local_rank": int(os.environ["LOCAL_RANK"]), "argv": sys.argv[1:]}))
    dist.destroy_process_group()
""")

PARENT = "import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.spawn_ranks(2, sys.argv[2:], script=sys.argv[1]))" % ROOT


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}


def test_parent_relays_rank0_json_last_and_passes_arguments(tmp_path):
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    p = subprocess.run([sys.executable, "-c", PARENT, str(child), "--steps", "3"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    out = json.loads(lines[-1])
    assert out == {"ranks": 2, "local_rank": 0, "argv": ["--steps", "3"]}
    assert "noise from rank 1" in p.stderr and "noise from rank 1" not in p.stdout   # other ranks: stderr only


def test_parent_fails_when_a_rank_fails(tmp_path):
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    p = subprocess.run([sys.executable, "-c", PARENT, str(child), "--fail-rank-1"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 7
    assert "rank 1 exited with code 7" in p.stderr


def test_mismatched_external_launcher_is_an_error_not_a_spawn():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in p.stderr
