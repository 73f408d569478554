#!/bin/bash
# usage (GPU box): bash tools/steps_ab.sh <workload> : steps-per-region x kernel-events matrix (pipeline fill/drain and event-marker cost)
W=${1:-cfg5}
for K in ${STEPS:-20 100 400}; do for EV in "" "--no-kernel-events"; do
  echo "== steps $K $EV"
  python3 bench.py --workload $W --also none --no-cpu-baseline --no-host-io --regions 9 --steps $K --parity-blocks 0 $EV 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline') or {}; t=d['timed_regions']
        print('   %.1f GS/s  ms/step med %.4f min %.4f max %.4f  fe(contended) %s' % (d['value']/1e3,t['ms_per_step_median'],t['ms_per_step_min'],t['ms_per_step_max'],r.get('avg_kernel_ms')))
"
done; done
