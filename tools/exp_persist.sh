for w in cfg2 cfg5; do
for k in 0 4; do
  echo "WORKLOAD=$w PERSIST=$k"
  PMR_FE_PERSIST=$k python3 bench.py --workload $w --no-cpu-baseline --steps 30 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value %.1f GS/s  ms/step %.4f  fe(contended) %.4f  isolated:'%(d['value']/1e3,d['ms_per_step'],r['avg_kernel_ms']), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
"
done
done
python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_dsd.py -m gpu -q -x 2>&1 | tail -3
