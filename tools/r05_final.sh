# the round's evidence in dependency order: kernel stats, the pair's PMC (copied into profiles/ BEFORE the bench lines are
# taken, so that their traffic_stale compares against the counters of the same tree), the full line, the reducer line
bash tools/r05_measurements.sh $1
bash tools/r05_group_pmc.sh $1
cp gpurun_out/r05_pmc_group_pair.json profiles/r05_pmc_group_pair.json
timeout 900 python bench.py > gpurun_out/r05_bench_full.json 2> gpurun_out/r05_bench_full.err
timeout 300 python bench.py --force-reducer --no-variants --no-cpu-baseline --no-ops-roofline > gpurun_out/r05_bench_reducer.json 2> gpurun_out/r05_bench_reducer.err
python - <<'PY'
import json
for f in ("r05_bench_full.json", "r05_bench_reducer.json"):
    for line in open("gpurun_out/" + f):
        if line.startswith("{"):
            d = json.loads(line)
            print(f, d["ms_per_step"], d["value"], "traffic_stale", d["roofline"]["traffic_stale"], d["roofline"]["traffic_commit"])
PY
