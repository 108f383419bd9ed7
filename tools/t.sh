R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_heads_gpu.py -x -q 2>&1 | tail -30 > $O/t_heads.txt
python tools/ab_step.py situation3d_amd.heads.ENABLED False True > $O/ab_heads.txt 2>&1
bash tools/glue_trace.sh
