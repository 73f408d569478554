mkdir -p gpurun_out
bash tools/lib_ab.sh $GRAFT_REPO_ROOT/sdr_pmr446_amd/alt_base.so cfg2 cfg3 cfg5 > gpurun_out/r16_libab.txt 2>&1
bash tools/lib_ab.sh $GRAFT_REPO_ROOT/sdr_pmr446_amd/alt_cw128.so cfg2 > gpurun_out/r16_cw128.txt 2>&1
cat gpurun_out/r16_libab.txt gpurun_out/r16_cw128.txt
