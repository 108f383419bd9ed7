# round 6: ONE evidence pass after the last code commit (run on the GPU box: gpurun -- 'bash tools/r06_final.sh <commit>')
#   step + surface-step kernel stats, the pair's PMC traffic, MFMA-pipe busy of the attention / gemmp kernels, the full
#   bench line, the data-parallel forms behind a real RCCL group of one (one cut / two cuts), FPS alone, the RCCL repro
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
C=${1:-unknown}
cd $R
bash tools/prof_step.sh r06_a_step
bash tools/prof_step.sh r06_surface_step --surface
# the pair's HBM traffic: separate --pmc passes, eager launches (a hipGraph replay has no per-kernel counters)
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r06_F -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline --no-ops-roofline > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_r06_W -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline --no-ops-roofline > /dev/null 2>&1
# MFMA-pipe busy: the step's attention kernels (52 x 52, 32 x 256: incl. attention_bwd_small_kernel) and, through the
# config-5 variant, the attention kernels at Nk = 80 000 / 5000 and gemmp_kernel
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_r06_mfma -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-ops-roofline > $O/pmc_r06_mfma.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_r06_F gpurun_out/pmc_r06_W gpurun_out/r06_pmc_group_pair.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline --no-ops-roofline" $C
{ echo "# MFMA-pipe busy per kernel (round 6)"; echo; echo 'command: `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-ops-roofline` (eager launches; the variants bring the config-5 shapes)'; echo;
  echo "busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 x 1024 SIMDs), per launch, averaged over the launches of one grid size"; echo;
  python tools/pmc_mfma_summary.py gpurun_out/pmc_r06_mfma attention; echo; python tools/pmc_mfma_summary.py gpurun_out/pmc_r06_mfma gemmp; } > gpurun_out/r06_pmc_mfma.md
rm -f gpurun_out/pmc_r06_*/r_kernel_trace.csv gpurun_out/pmc_r06_*/*/r_kernel_trace.csv
cp gpurun_out/r06_pmc_group_pair.json profiles/r06_pmc_group_pair.json   # BEFORE the bench lines: their traffic_stale compares against it
timeout 900 python bench.py > gpurun_out/r06_bench_full.json 2> gpurun_out/r06_bench_full.err
export SIG3D_SINGLE_RANK_PG=1
timeout 300 python bench.py --force-reducer --qf-cut 6 --no-variants --no-cpu-baseline --no-ops-roofline > gpurun_out/r06_bench_reducer_cut6.json 2> gpurun_out/r06_bench_reducer_cut6.err
timeout 300 python bench.py --force-reducer --qf-cut 0 --no-variants --no-cpu-baseline --no-ops-roofline > gpurun_out/r06_bench_reducer_cut0.json 2> gpurun_out/r06_bench_reducer_cut0.err
unset SIG3D_SINGLE_RANK_PG
bash tools/cut_ab.sh 6 3 > gpurun_out/r06_cut_ab.txt 2>&1
python tools/fps_bench.py > gpurun_out/r06_fps_bench.txt 2>&1
timeout 600 python tools/rccl_graph_replay_repro.py > gpurun_out/r06_rccl_graph_replay.txt 2>&1
python - <<'PY'
import json
for f in ("r06_bench_full.json", "r06_bench_reducer_cut6.json", "r06_bench_reducer_cut0.json"):
    for line in open("gpurun_out/" + f):
        if line.startswith("{"):
            d = json.loads(line)
            print(f, d["ms_per_step"], d["value"], "pair frac", d["roofline"]["frac"], "traffic_stale", d["roofline"].get("traffic_stale"),
                  "ops", (d.get("roofline_ops") or {}).get("frac"), (d.get("roofline_ops") or {}).get("group_points", {}).get("frac"),
                  "comm", (d.get("comm") or {}).get("exposed_ms"))
PY
