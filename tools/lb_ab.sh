#!/bin/bash
# usage (GPU box): bash tools/lb_ab.sh <workload> : in-kernel dc carry (default) vs PMR_FE_LOOKBACK=0, with the fallback-tile count
W=${1:-cfg2}
for E in PMR_X=0 PMR_FE_LOOKBACK=0 PMR_X=0 PMR_FE_LOOKBACK=0; do
  echo "== $E"
  env $E python3 bench.py --workload $W --also none --no-cpu-baseline --regions 9 --parity-blocks 2 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline') or {}; t=d['timed_regions']
        print('   %.1f GS/s  ms/step med %.4f min %.4f max %.4f  fe(contended) %s' % (d['value']/1e3,t['ms_per_step_median'],t['ms_per_step_min'],t['ms_per_step_max'],r.get('avg_kernel_ms')))
        print('   isolated', {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()}, 'fallback', r.get('carry_fallback_tiles',{}).get('flagged'), '/', r.get('carry_fallback_tiles',{}).get('tiles'), 'parity', (d.get('parity_checked') or {}).get('max_abs_pcm_diff_lsb'))
"
done
