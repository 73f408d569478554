mkdir -p gpurun_out
for E in "PMR_PRIO_DEBUG=1" "PMR_PRIO_BASE=0"; do
  echo "== $E"
  env $E python3 bench.py --workload cfg5 --also cfg2,cfg3 --no-cpu-baseline --parity-blocks 0 --no-kernel-events --regions 5 2>gpurun_out/r23.err | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   cfg5 %.1f' % (d['value']/1e3), {k: round(v['value']/1e3,1) for k,v in d['also'].items()})
"
  grep "priority range" gpurun_out/r23.err | head -1
done > gpurun_out/r23.txt 2>&1
cat gpurun_out/r23.txt
