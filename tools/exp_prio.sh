for pr in 0 1; do
for m in lds global; do
  PMR_STREAM_PRIO=$pr PMR_FIR_MFMA=$m python3 bench.py --no-cpu-baseline --steps 30 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('PRIO=$pr FIR=$m value %.1f GS/s  ms/step %.4f  fe(contended) %.4f' % (d['value']/1e3,d['ms_per_step'],r['avg_kernel_ms']))
"
done
done
