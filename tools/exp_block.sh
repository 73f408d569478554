for lb in 24 26 27 28; do
  python3 bench.py --no-cpu-baseline --steps 20 --log2-block $lb 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('log2_block=$lb value %.1f GS/s  ms/step %.4f  fe(contended) %.4f isolated' % (d['value']/1e3,d['ms_per_step'],r['avg_kernel_ms']), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
"
done
