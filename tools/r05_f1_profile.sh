cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SIG3D_MLP_MIN_POSITIONS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f1 -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants > /dev/null 2>&1
rm -f $R/gpurun_out/prof_f1/r_kernel_trace.csv
cd $R
python tools/summarize_rocprof.py gpurun_out/prof_f1 gpurun_out/r05_f1_step.md "SIG3D_MLP_MIN_POSITIONS=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants"
