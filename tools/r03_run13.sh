mkdir -p gpurun_out
for E in "PMR_X=0" "PMR_FIR_MFMA=4"; do
env $E python3 bench.py --workload cfg2 --also none --no-cpu-baseline --regions 5 --parity-blocks 0 --ctcss 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$E ctcss: all %.1f GS/s one-open %.1f GS/s isolated' % (d['value']/1e3, d['one_open_channel']['value']/1e3), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
"
done > gpurun_out/r13.txt
cat gpurun_out/r13.txt
