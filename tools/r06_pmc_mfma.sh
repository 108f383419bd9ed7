# MFMA-pipe busy of this round's attention and gemmp kernels: one --pmc pass per small program (the pass over the whole
# bench.py run -- variants included -- crashed the profiler), program directly after `--`
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
P="--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv"
timeout 300 rocprofv3 $P -d $O/pmc_r06_mfma_small -o r -- python3 $R/tools/attn_small_bench.py > $O/pmc_r06_mfma_small.log 2>&1
timeout 300 rocprofv3 $P -d $O/pmc_r06_mfma_big -o r -- python3 $R/tools/attn_bench.py 80000 5000 > $O/pmc_r06_mfma_big.log 2>&1
timeout 600 rocprofv3 $P -d $O/pmc_r06_mfma_c5 -o r -- python3 $R/tools/config5_repeat.py 1 > $O/pmc_r06_mfma_c5.log 2>&1
cd $R
{ echo "# MFMA-pipe busy per kernel (round 6)"; echo
  echo 'busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 x 1024 SIMDs), per launch, averaged over the launches of one grid size.'
  echo 'One pass per program: `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 <program>`.'; echo
  echo '## the step'"'"'s shapes (B = 8, 12 heads of 64: 52 x 52 self-attention, 32 x 256 cross-attention): `tools/attn_small_bench.py`'; echo
  python tools/pmc_mfma_summary.py gpurun_out/pmc_r06_mfma_small attention; echo
  echo '## the 3D-LLM shapes (B = 4, 32 queries x 80 000 / 5000 scene tokens): `tools/attn_bench.py 80000 5000`'; echo
  python tools/pmc_mfma_summary.py gpurun_out/pmc_r06_mfma_big attention; echo
  echo '## config 5 (`bench.config5_variant` once, `tools/config5_repeat.py 1`): the key / value projections on `gemmp_kernel` (bf16 MFMAs: six per f32-equivalent product), `planes_split_kernel`, and the attention kernels behind them'; echo
  python tools/pmc_mfma_summary.py gpurun_out/pmc_r06_mfma_c5 gemmp; echo
  python tools/pmc_mfma_summary.py gpurun_out/pmc_r06_mfma_c5 attention; } > gpurun_out/r06_pmc_mfma.md
rm -f gpurun_out/pmc_r06_mfma_*/r_kernel_trace.csv gpurun_out/pmc_r06_mfma_*/*/r_kernel_trace.csv
cat gpurun_out/r06_pmc_mfma.md | cut -c1-200
