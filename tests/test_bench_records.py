"""The bench lines committed under profiles/ for this round, read as the judge reads them (no GPU): each is one JSON line
with bench.py's contract fields, every roofline fraction follows from its own achieved / peak and stays below 1, the metric
is BASELINE.json's, and the figures DESIGN.md / README.md quote are the ones in the files."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = ["profiles/r06_bench_full.json", "profiles/r06_bench_reducer_cut6.json", "profiles/r06_bench_reducer_cut0.json"]
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _line(path):
    lines = [ln for ln in open(os.path.join(ROOT, path)).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "%s: expected ONE JSON line, found %d" % (path, len(lines))
    return json.loads(lines[0])


def _fractions(obj, where=""):
    """Every {achieved, peak, frac} object anywhere in the line -> (path, object)."""
    if isinstance(obj, dict):
        if "frac" in obj and "achieved" in obj and "peak" in obj:
            yield where, obj
        for k, v in obj.items():
            yield from _fractions(v, where + "/" + str(k))
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            yield from _fractions(v, where + "/" + str(i))


@pytest.mark.parametrize("path", RECORDS)
def test_committed_line_carries_the_contract(path):
    d = _line(path)
    for key in CONTRACT:
        assert key in d, key
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert base["metric"].startswith(d["metric"])              # BASELINE adds "at 1/2/4/8 MI355X"
    assert d["unit"] == "samples/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and base["published"] == {}   # nothing published to compare against
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    # value = samples of all ranks / time: B = 8 per GPU
    assert abs(d["value"] - 8 * d["n_gpus"] / d["ms_per_step"] * 1e3) < 0.5
    assert d["fps_timeouts"] == 0 and d["launch_mode"] == "hipGraph replay"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert "ball_query" in r["kernel"] and "query_group" in r["kernel"]          # the pair the north star names
    assert r["traffic"] is None or r["traffic"] > 0


@pytest.mark.parametrize("path", RECORDS)
def test_every_roofline_fraction_follows_from_its_fields_and_none_exceeds_one(path):
    d = _line(path)
    seen = list(_fractions(d))
    assert len(seen) >= 4
    for where, r in seen:
        assert 0 < r["frac"] < 1, (where, r["frac"])
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3, (where, r["frac"], r["achieved"], r["peak"])


def test_cpu_baseline_is_the_oracle_on_the_whole_batch():
    c = _line(RECORDS[0])["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "samples/s" and c["value"] > 0 and c["cores"] >= 1
    assert c["sample"].startswith("8 scenes (B=8 of the B=8 step)")


def test_the_cut_is_cheaper_than_one_cut_in_the_committed_pair():
    """The data-parallel step behind a real RCCL group of one: cutting the backward pass a second time (after Q-Former
    layer 6, two kind-major arenas) is the cheaper form -- the numbers README.md and DESIGN.md quote."""
    cut6, cut0 = _line(RECORDS[1]), _line(RECORDS[2])
    assert cut6["dist_backend"] == cut0["dist_backend"] == "nccl" and cut6["rccl_ranks"] == 1
    assert cut6["ms_per_step"] < cut0["ms_per_step"]
    for name in ("README.md", "DESIGN.md"):
        text = open(os.path.join(ROOT, name)).read()
        assert "%.2f" % cut6["ms_per_step"] in text, (name, cut6["ms_per_step"])
    forms = cut6["comm"]["model"]["forms"]          # predicted exposure of both forms at every world size, in the line
    assert {"N=2", "N=4", "N=8"} <= set(forms)
    for at in forms.values():
        assert at["cut"]["exposed_ms"] <= at["uncut"]["exposed_ms"] and at["cut"]["wire_ms"] == at["uncut"]["wire_ms"]


def test_the_pair_counters_were_read_in_a_tree_whose_pair_sources_are_the_head_s():
    """roofline.traffic comes from profiles/*_pmc_group_pair.json; bench.py marks it stale when a source that decides the
    pair's launches changed after the counters were read.  The committed counters must not be stale against this tree, and
    the rule must notice a change in one of the pair's files and ignore one elsewhere."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    pmcs = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_group_pair.json"))
    hashes = json.load(open(os.path.join(ROOT, "profiles", pmcs[-1])))["source_hashes"]
    assert any(f.endswith("ball_query.hip") for f in hashes) and any(f.endswith("group_points.hip") for f in hashes)
    assert bench.pair_traffic_is_stale(hashes) is False
    touched = dict(hashes)
    touched["situation3d_amd/csrc/group_points.hip"] = "0" * 16
    assert bench.pair_traffic_is_stale(touched) is True
    elsewhere = dict(hashes)
    elsewhere["situation3d_amd/csrc/gemmp_core.h"] = "0" * 16
    assert bench.pair_traffic_is_stale(elsewhere) is False
    assert bench.pair_traffic_is_stale({"situation3d_amd/csrc/gemmp_core.h": "0" * 16}) is True    # nothing to compare


def test_algorithmic_bytes_are_survey_8d_s():
    """SURVEY.md 8(d) fixes the numerator of every `achieved`: ball_query B(12N + 12M + 4 M ns), group_points(C)
    B(4CN + 4 M ns + 4C M ns); at B = 8 over the suggested stack that is 8.23 / 1.34 / 0.41 / 0.20 MB of ball query and
    314.8 MB for the pair.  bench.py's functions must give those figures (the fused launch reads the index list once for
    both groups, so it counts 4 M ns less than the two groups apart), and the committed line's op-level pair must be them."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    b = 8
    levels = [(40000, 2048, 64, 3), (2048, 1024, 32, 128), (1024, 512, 16, 256), (512, 256, 16, 256)]

    def group(c, n, m, ns):
        return b * (4 * c * n + 4 * m * ns + 4 * c * m * ns)

    bq = [bench.ball_query_algorithmic_bytes(b, n, m, ns) for n, m, ns, _ in levels]
    assert [round(x / 1e6, 2) for x in bq] == [8.23, 1.34, 0.41, 0.20]
    apart = [(group(3, n, m, ns), group(c, n, m, ns)) for n, m, ns, c in levels]
    assert [(round(x / 1e6, 1), round(f / 1e6, 1)) for x, f in apart] == [(20.6, 20.6), (4.4, 143.7), (1.1, 75.8), (0.6, 37.9)]
    assert round((sum(bq) + sum(x + f for x, f in apart)) / 1e6, 1) == 314.8
    fused = [bench.group_algorithmic_bytes(b, n, m, ns, c) for n, m, ns, c in levels]
    assert all(x + f - g == b * 4 * m * ns for (x, f), g, (n, m, ns, c) in zip(apart, fused, levels))
    ops = _line(RECORDS[0])["roofline_ops"]
    assert ops["algorithmic_bytes"] == sum(bq) + sum(fused)
