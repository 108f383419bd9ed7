cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | tail -3 > $O/t_dw.txt
for i in 1 2; do
SIG3D_MLP_NT=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-variants --no-ops-roofline 2>/dev/null | python -c "import sys,json; print('new', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" >> $O/t_dw.txt
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-ops-roofline --surface 2>/dev/null | python -c "import sys,json; print('surface', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" >> $O/t_dw.txt
