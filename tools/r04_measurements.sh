# round-4 measurement batch (run on the GPU box through gpurun): kernel stats of the step (headline data and
# surface-shaped data), PMC traffic of the pair, the own-GEMM A/B profile, the GEMM shape table, the full bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04 -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $O/prof_r04.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04_surface -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --surface > $O/prof_r04_surface.log 2>&1
SIG3D_QF_GEMM=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04_own -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-ops-roofline > $O/prof_r04_own.log 2>&1
rm -f $O/prof_r04*/r_kernel_trace.csv
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r04_F -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline --no-ops-roofline > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_r04_W -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline --no-ops-roofline > /dev/null 2>&1
cd $R
python tools/summarize_rocprof.py gpurun_out/prof_r04 gpurun_out/r04_a_step.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants"
python tools/summarize_rocprof.py gpurun_out/prof_r04_surface gpurun_out/r04_surface_step.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --surface"
python tools/summarize_rocprof.py gpurun_out/prof_r04_own gpurun_out/r04_own_gemm_step.md "SIG3D_QF_GEMM=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-ops-roofline"
python tools/pmc_traffic.py gpurun_out/pmc_r04_F gpurun_out/pmc_r04_W gpurun_out/r04_pmc_group_pair.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline --no-ops-roofline" $1
rm -f gpurun_out/pmc_r04_*/r_kernel_trace.csv
(cd tools/micro && ./gemm16_bench -1 quick) > gpurun_out/r04_gemm16_shapes.txt 2>&1
(cd tools/micro && ./mfma16_rate && ./mfma16_fill) > gpurun_out/r04_mfma16_micro.txt 2>&1
timeout 600 python bench.py > gpurun_out/r04_bench_full.json 2> gpurun_out/r04_bench_full.err
timeout 300 python bench.py --force-reducer --no-variants --no-cpu-baseline --no-ops-roofline > gpurun_out/r04_bench_reducer.json 2> gpurun_out/r04_bench_reducer.err
# late round 4: the bf16 x 6 GEMM core beside the f32 core and the library, product by product; the read + MFMA stream
timeout 600 python tools/gemmx6_bench.py > gpurun_out/r04_gemmx6_shapes.txt 2>&1
(cd tools/micro && ./bf16x6_rate) > gpurun_out/r04_bf16x6_micro.txt 2>&1
bash tools/glue_trace.sh
