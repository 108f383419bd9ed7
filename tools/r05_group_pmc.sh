# round 5: PMC traffic of the north-star pair after the XCD-local grouping grids (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r05_F -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline --no-ops-roofline > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_r05_W -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline --no-ops-roofline > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_r05_F gpurun_out/pmc_r05_W gpurun_out/r05_pmc_group_pair.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline --no-ops-roofline" $1
rm -f gpurun_out/pmc_r05_*/r_kernel_trace.csv
