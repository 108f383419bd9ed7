# PMC passes over ONE program (kernel-trace + counters only, one small counter set per pass):
#   bash tools/pmc_kernel.sh <out-name> <program> [args...]      -> gpurun_out/pmc_<out-name>/<set>/
# Summarise with tools/pmc_table.py.
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
prog=$(realpath $1); shift
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  d=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $root/gpurun_out/pmc_$name/$d -o r -- $prog "$@" > $root/gpurun_out/pmc_${name}_$d.log 2>&1 || echo "FAILED $set"
done
