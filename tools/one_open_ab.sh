#!/bin/bash
# usage (GPU box): [BENCH_ARGS="--ctcss"] bash tools/one_open_ab.sh <workload> "ENV=VAL ..." ... : all-channel and one-open-channel rates under each environment
W=$1; shift
for E in "$@"; do
  echo "== $E"
  env $E python3 bench.py --workload $W --also none --no-cpu-baseline --no-host-io --regions 7 --parity-blocks 0 $BENCH_ARGS 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o=d.get('one_open_channel') or {}
        print('   all channels %.1f GS/s   one open channel %s GS/s' % (d['value']/1e3, o.get('value') and round(o['value']/1e3,1)))
"
done
