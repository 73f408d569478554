mkdir -p gpurun_out
( timeout 1200 python3 -m pytest tests/test_gpu_pipelined.py -q 2>&1 | tail -4 ) > gpurun_out/r19.txt
BENCH_ARGS="--no-kernel-events" bash tools/env_ab.sh cfg5 "PMR_X=0" "PMR_STREAM_PRIO=0" "PMR_X=0" "PMR_STREAM_PRIO=0" >> gpurun_out/r19.txt 2>&1
cat gpurun_out/r19.txt
