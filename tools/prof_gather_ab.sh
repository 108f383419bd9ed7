cd /tmp && export TMPDIR=/tmp && export SIG3D_COMPACT=0
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_g1 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants > /dev/null 2>&1
export SIG3D_GATHER_L0=0
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_g0 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants > /dev/null 2>&1
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_g*/r_kernel_trace.csv
