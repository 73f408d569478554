#!/bin/bash
# usage (GPU box): [BENCH_ARGS="--no-kernel-events"] bash tools/env_ab.sh <workload> "ENV=VAL ..." ... : bench the workload under each environment setting
W=$1; shift
for E in "$@"; do
  echo "== $E"
  env $E python3 bench.py --workload $W --also none --no-cpu-baseline --no-host-io --regions 9 --parity-blocks 0 $BENCH_ARGS 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline') or {}; t=d['timed_regions']
        print('   %.1f GS/s  ms/step med %.4f min %.4f max %.4f  fe(contended) %s' % (d['value']/1e3,t['ms_per_step_median'],t['ms_per_step_min'],t['ms_per_step_max'],r.get('avg_kernel_ms')))
"
done
