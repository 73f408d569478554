for g in default 256x8 128x16; do
  echo "GEOM=$g"
  PMR_FE_GEOM=$g python3 bench.py --no-cpu-baseline --steps 30 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value %.1f GS/s  ms/step %.4f  fe(contended) %.4f  isolated:'%(d['value']/1e3,d['ms_per_step'],r['avg_kernel_ms']), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
"
done
PMR_FE_GEOM=256x8 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cfg2 or blocks or split" 2>&1 | tail -2
