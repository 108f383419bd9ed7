# round-3 measurement batch (run on the GPU box through gpurun): kernel stats of the step, PMC traffic of the pair,
# MFMA-pipe busy of the attention kernels at the config-5 and headline shapes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03 -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $O/prof_r03.log 2>&1
rm -f $O/prof_r03/r_kernel_trace.csv
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r03_F -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline --no-ops-roofline > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_r03_W -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline --no-ops-roofline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_r03_att -o r -- python3 $R/tools/attn_bench.py 256 80000 > $O/pmc_r03_att.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r03_attF -o r -- python3 $R/tools/attn_bench.py 80000 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_r03_step -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline --no-ops-roofline > /dev/null 2>&1
cd $R
(python tools/pmc_mfma_summary.py gpurun_out/pmc_r03_step Cijk; python tools/pmc_mfma_summary.py gpurun_out/pmc_r03_step mlp_ | tail -n +3) > gpurun_out/r03_pmc_step_mfma.md
python tools/summarize_rocprof.py gpurun_out/prof_r03 gpurun_out/r03_profile.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants" 16
python tools/pmc_traffic.py gpurun_out/pmc_r03_F gpurun_out/pmc_r03_W gpurun_out/r03_pmc_group_pair.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline" $1
python tools/pmc_mfma_summary.py gpurun_out/pmc_r03_att attention > gpurun_out/r03_pmc_attention_mfma.md
python tools/pmc_mfma_summary.py gpurun_out/pmc_r03_attF attention > /dev/null 2>&1
rm -f gpurun_out/pmc_r03_*/r_kernel_trace.csv
python tools/attn_bench.py 256 5000 80000 > gpurun_out/r03_attn_bench.txt 2>&1
timeout 300 python bench.py > gpurun_out/r03_bench_full.json 2> gpurun_out/r03_bench_full.err
