for f in 0 1; do
  if [ $f = 1 ]; then export PMR_NOFIX=1; fi
  python3 bench.py --no-cpu-baseline --steps 30 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('NOFIX=$f value %.1f GS/s  ms/step %.4f  isolated' % (d['value']/1e3,d['ms_per_step']), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
"
done
