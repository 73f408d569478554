for t in 1 2; do echo TPW=$t; PMR_FIR_TPW=$t bash tools/quick_bench.sh cfg2; done
bash tools/quick_bench.sh cfg3; bash tools/quick_bench.sh cfg5
python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_ctcss.py tests/test_io.py -m gpu -q -x 2>&1 | tail -1
