mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_gpu_ctcss.py tests/test_gpu_pipelined.py tests/test_gpu_mask.py -x -q 2>&1 | tail -8 ) > gpurun_out/r11_test.txt
python3 bench.py --workload cfg2 --also none --no-cpu-baseline --regions 5 --parity-blocks 0 --ctcss 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('ctcss: all %.1f GS/s one-open %.1f GS/s isolated' % (d['value']/1e3, d['one_open_channel']['value']/1e3), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
" >> gpurun_out/r11_test.txt
cat gpurun_out/r11_test.txt
