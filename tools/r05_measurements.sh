# round-5 measurement batch (run on the GPU box through gpurun): kernel stats of the step (headline data and
# surface-shaped data), PMC traffic of the pair, the full bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r05 -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $O/prof_r05.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r05_surface -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --surface > $O/prof_r05_surface.log 2>&1
rm -f $O/prof_r05*/r_kernel_trace.csv
cd $R
python tools/summarize_rocprof.py gpurun_out/prof_r05 gpurun_out/r05_a_step.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants"
python tools/summarize_rocprof.py gpurun_out/prof_r05_surface gpurun_out/r05_surface_step.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --surface"
if [ "$2" = "full" ]; then
  bash tools/r05_group_pmc.sh $1
  timeout 900 python bench.py > gpurun_out/r05_bench_full.json 2> gpurun_out/r05_bench_full.err
  timeout 300 python bench.py --force-reducer --no-variants --no-cpu-baseline --no-ops-roofline > gpurun_out/r05_bench_reducer.json 2> gpurun_out/r05_bench_reducer.err
fi
