mkdir -p gpurun_out
BENCH_ARGS="--no-kernel-events" bash tools/env_ab.sh cfg3 "PMR_X=0" "PMR_STREAM_PRIO=fe PMR_FIR_MFMA=global" "PMR_FIR_MFMA=global" "PMR_X=0" "PMR_STREAM_PRIO=fe PMR_FIR_MFMA=global" > gpurun_out/r21.txt 2>&1
cat gpurun_out/r21.txt
