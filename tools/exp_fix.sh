for f in 1 0; do
  PMR_DCFIX_FUSE=$f python3 bench.py --no-cpu-baseline --steps 30 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('FUSE=$f value %.1f GS/s  ms/step %.4f  fe(contended) %.4f isolated' % (d['value']/1e3,d['ms_per_step'],r['avg_kernel_ms']), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
"
done
python3 bench.py --workload cfg3 --no-cpu-baseline --steps 30 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('cfg3 value %.1f GS/s  ms/step %.4f  isolated' % (d['value']/1e3,d['ms_per_step']), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
"
python3 -m pytest tests -m gpu -q -x 2>&1 | tail -2
