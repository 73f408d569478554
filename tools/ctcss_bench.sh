#!/bin/bash
# usage (GPU box): bash tools/ctcss_bench.sh [workload] : chain with the CTCSS detector on -- all channels / one open channel, isolated kernel times
python3 bench.py --workload ${1:-cfg2} --also none --no-cpu-baseline --no-host-io --regions 7 --parity-blocks 0 --ctcss 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; o=d.get('one_open_channel') or {}
        print('ctcss on: all channels %.1f GS/s (ms/step %.4f), one open channel %s GS/s' % (d['value']/1e3, d['ms_per_step'], o.get('value') and round(o['value']/1e3,1)))
        print('  isolated', {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
        print('  one-open isolated', {k:round(v,4) for k,v in (o.get('kernels_ms_per_step_isolated') or {}).items()})
"
