# kernel stats of the step under rocprofv3 -> gpurun_out/<tag>.md   (usage: bash tools/prof_step.sh <tag> [bench flags])
# environment switches for the profiled run are exported by the caller (the program must follow `--` directly)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-ops-roofline "$@" > $O/prof_$TAG.log 2>&1
rm -f $O/prof_$TAG/r_kernel_trace.csv $O/prof_$TAG/*/r_kernel_trace.csv
cd $R
python tools/summarize_rocprof.py gpurun_out/prof_$TAG gpurun_out/$TAG.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-ops-roofline $*"
tail -1 $O/prof_$TAG.log | cut -c1-200
