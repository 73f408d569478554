mkdir -p gpurun_out
for rep in 1 2 3; do for E in "PMR_X=0" "PMR_STREAM_PRIO=0"; do
  echo "== $E"
  env $E python3 bench.py --workload cfg5 --also none --no-cpu-baseline --parity-blocks 0 --no-kernel-events --steps 20 --warmup 3 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['timed_regions']
        print('   %.1f GS/s  ms/step med %.4f min %.4f max %.4f first %.4f' % (d['value']/1e3,t['ms_per_step_median'],t['ms_per_step_min'],t['ms_per_step_max'],t['ms_per_step_first']))
"
done; done > gpurun_out/r20.txt 2>&1
cat gpurun_out/r20.txt
