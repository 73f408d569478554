# usage: gpurun -- 'bash tools/quick_bench.sh [workload]'
python3 bench.py --workload ${1:-cfg2} --also none --no-cpu-baseline --no-host-io --regions 5 --parity-blocks 0 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value %.1f GS/s  ms/step %.4f  fe(contended) %.4f isolated' % (d['value']/1e3,d['ms_per_step'],r['avg_kernel_ms']), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
"
