mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_gpu_fullbank.py tests/test_gpu_mask.py tests/test_gpu_ctcss.py -x -q -s 2>&1 | grep -v "^$" | tail -30 ) > gpurun_out/r8_test.txt
( timeout 600 python3 bench.py --gpus 2 --dist-backend gloo --no-cpu-baseline --regions 3 --parity-blocks 0 --no-kernel-events 2>gpurun_out/r8_2rank.err | grep '^{' > gpurun_out/r8_bench_2rank_1gpu.json; echo rc=$? >> gpurun_out/r8_test.txt; tail -3 gpurun_out/r8_2rank.err >> gpurun_out/r8_test.txt )
cat gpurun_out/r8_test.txt; python3 -c "
import json; d=json.load(open('gpurun_out/r8_bench_2rank_1gpu.json')); print(d['value'], d['n_gpus'], d.get('dist'))"
