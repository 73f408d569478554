mkdir -p gpurun_out
for E in "PMR_X=0" "PMR_STREAM_PRIO=0"; do
  echo "== $E"
  env $E python3 bench.py --workload cfg5 --also cfg2,cfg3 --no-cpu-baseline --parity-blocks 0 --no-kernel-events --regions 5 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   cfg5 %.1f' % (d['value']/1e3), {k: round(v['value']/1e3,1) for k,v in d['also'].items()})
"
done > gpurun_out/r22.txt 2>&1
echo "== order cfg2 first" >> gpurun_out/r22.txt
python3 bench.py --workload cfg2 --also cfg3,cfg5 --no-cpu-baseline --parity-blocks 0 --no-kernel-events --regions 5 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   cfg2 %.1f' % (d['value']/1e3), {k: round(v['value']/1e3,1) for k,v in d['also'].items()})
" >> gpurun_out/r22.txt 2>&1
cat gpurun_out/r22.txt
