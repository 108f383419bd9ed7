# round-end measurement batch (run on the GPU box through gpurun): kernel stats, PMC traffic of the pair, timelines
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $R/gpurun_out/prof_final.log 2>&1
rm -f $R/gpurun_out/prof_final/r_kernel_trace.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_final_F -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_final_W -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline > /dev/null 2>&1
rm -f $R/gpurun_out/pmc_final_*/r_kernel_trace.csv
cd $R
python tools/branch_timeline.py > gpurun_out/tl_final_pref.txt 2>&1
python tools/branch_timeline.py --no-prefetch > gpurun_out/tl_final_nopref.txt 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
