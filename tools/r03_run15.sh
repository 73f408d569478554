mkdir -p gpurun_out
bash tools/lib_ab.sh $GRAFT_REPO_ROOT/sdr_pmr446_amd/alt_base.so cfg2 cfg3 cfg5 > gpurun_out/r15_libab.txt 2>&1
BENCH_ARGS="" bash tools/env_ab.sh cfg5 "PMR_X=0" "PMR_FFT_FPW=3" "PMR_X=0" "PMR_FFT_FPW=3" > gpurun_out/r15_fpw.txt 2>&1
cat gpurun_out/r15_libab.txt gpurun_out/r15_fpw.txt
