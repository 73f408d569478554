mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -io "SQ_[A-Z_]*MFMA[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_WAIT[A-Z_]*\|SQ_LDS[A-Z_]*\|SQ_ACTIVE[A-Z_]*" | sort -u | tr '\n' ' ' > gpurun_out/r6_counters.txt
for V in tiles 32; do
  PMR_FIR_MFMA=$V bash tools/pmc_any.sh cfg2 k_fir "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES" > gpurun_out/r6_pmc_$V.txt 2>&1
done
cat gpurun_out/r6_*.txt
