cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD"; do
  name=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_att/$name -o r -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py 80000 > $GRAFT_REPO_ROOT/gpurun_out/pmc_att_$name.log 2>&1 || echo "FAILED $set"
done
