#!/bin/bash
# usage (GPU box): bash tools/lib_ab.sh <other libpmr446_hip.so> [workloads...] : the in-tree build vs another build of the library,
# interleaved on the same box (box-to-box spread is +-2 %, more than most single changes)
ALT=$1; shift
for W in ${@:-cfg5 cfg2 cfg3}; do
  for rep in 1 2; do for L in "" "$ALT"; do
    echo "== $W ${L:-in-tree}"
    PMR_LIBRARY=$L python3 bench.py --allow-experiment --workload $W --also none --no-cpu-baseline --no-host-io --regions 9 --parity-blocks 0 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline') or {}; t=d['timed_regions']
        print('   %.1f GS/s  ms/step med %.4f min %.4f  fe(contended) %.4f  isolated %s' % (d['value']/1e3,t['ms_per_step_median'],t['ms_per_step_min'],r.get('avg_kernel_ms'), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()}))
"
  done; done
done
