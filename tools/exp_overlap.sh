for x in 0 17000 44000; do
  echo "FE_LDS_EXTRA=$x"
  PMR_FE_LDS_EXTRA=$x python3 bench.py --no-cpu-baseline --steps 30 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value %.1f GS/s  ms/step %.4f  fe(contended) %.4f  isolated:'%(d['value']/1e3,d['ms_per_step'],r['avg_kernel_ms']), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
"
done
