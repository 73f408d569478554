mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_gpu_ctcss.py tests/test_gpu_pipelined.py -x -q 2>&1 | tail -8 ) > gpurun_out/r9_test.txt
for W in cfg2; do
  for A in "" "--ctcss"; do
    python3 bench.py --workload $W --also none --no-cpu-baseline --regions 5 --parity-blocks 0 --no-kernel-events $A 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$W $A all %.1f GS/s  one-open %.1f GS/s' % (d['value']/1e3, d['one_open_channel']['value']/1e3))
" >> gpurun_out/r9_test.txt
  done
done
cat gpurun_out/r9_test.txt
