for r in 1 2 3; do
for W in cfg3 cfg2; do
  for E in "" "--no-kernel-events"; do
    echo -n "$W bench.py $E: "; python3 bench.py --workload $W --also none --steps 20 --warmup 5 --no-cpu-baseline --no-host-io --parity-blocks 0 --no-one-open $E 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['timed_regions']; print('%.1f GS/s med %.4f min %.4f max %.4f first %.4f' % (d['value']/1e3, t['ms_per_step_median'], t['ms_per_step_min'], t['ms_per_step_max'], t['ms_per_step_first']))"
  done
  echo -n "$W lean: "; python3 tools/ab_libs.py --libs x= --reps 1 --workloads $W --steps 20 --regions 25 | tail -1
done; done
