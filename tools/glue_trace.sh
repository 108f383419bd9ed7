# ordered kernel list of one replayed step (main branch) + the Python sites of the torch glue launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/glue_tr -o r -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-variants --no-ops-roofline > $O/glue_tr.log 2>&1
cd $R
python tools/step_timeline.py $(ls gpurun_out/glue_tr/*kernel_trace.csv | head -1) 2 --list > gpurun_out/glue_timeline.txt 2>&1
rm -f gpurun_out/glue_tr/*kernel_trace.csv
timeout 300 python tools/glue_sites.py > gpurun_out/glue_sites.txt 2>&1
