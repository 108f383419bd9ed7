cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py -x -q -k streaming 2>&1 | tail -3 > $O/t_dw.txt
python tools/dw_stream_probe.py 2>&1 | tail -6 >> $O/t_dw.txt
python tools/ab_step.py situation3d_amd.pointnet2.fused_mlp.DW_STREAM False True --rounds 12 > $O/ab_dw.txt 2>&1
