# PMC passes over ONE program (kernel-trace + counters only, one small counter set per pass):
#   bash tools/pmc_kernel.sh <out-name> <program> [args...]      -> gpurun_out/pmc_<out-name>/<set>/
# Summarise with tools/pmc_table.py.
# <program> must be an ELF executable (a compiled bench, or the python3 interpreter itself followed by the script): with
# --pmc the profiler initialises the GPU before the program starts, and any exec hop after that -- a shell script, a
# "#!/usr/bin/env" shebang, a launcher that re-execs -- takes the machine down on this pool.
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
prog=$(command -v "$1" || true); shift
[ -n "$prog" ] && prog=$(readlink -f "$prog")
if [ -z "$prog" ] || [ "$(head -c4 "$prog" | tr -d '\0' | cut -c2-4)" != "ELF" ]; then
  echo "pmc_kernel.sh: '$prog' is not an ELF executable (see the header)"; exit 2
fi
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  d=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $root/gpurun_out/pmc_$name/$d -o r -- $prog "$@" > $root/gpurun_out/pmc_${name}_$d.log 2>&1 || echo "FAILED $set"
done
