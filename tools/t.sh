cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | tail -3 > $O/t_dw.txt
bash tools/glue_trace.sh
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --no-ops-roofline 2>/dev/null | python -c "import sys,json; print('bench', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" >> $O/t_dw.txt
