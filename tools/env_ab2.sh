#!/bin/bash
# usage (GPU box): bash tools/env_ab2.sh <workload> "ENV=VAL ..." ... : like env_ab.sh, also the one-open-channel rate
W=$1; shift
for E in "$@"; do
  echo "== $E"
  env $E python3 bench.py --workload $W --also none --no-cpu-baseline --no-host-io --regions 9 --parity-blocks 0 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['timed_regions']
        print('   all channels %.1f GS/s (ms/step %.4f)   one open channel %.1f GS/s' % (d['value']/1e3,t['ms_per_step_median'],d['one_open_channel']['value']/1e3))
"
done
