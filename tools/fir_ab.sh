#!/bin/bash
# usage (GPU box): bash tools/fir_ab.sh > gpurun_out/fir_ab.txt : audio FIR A/B at cfg2 -- MFMA vs packed-VALU, and the CTCSS branch in the same pass vs its own
run() { echo "== $*"; env "$@" python3 bench.py --workload cfg2 --also none --no-cpu-baseline --regions 5 --parity-blocks 0 $EXTRA 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('   chain %.1f GS/s  ms/step %.4f  kernels(isolated ms):' % (d['value']/1e3,d['ms_per_step']), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
"; }
EXTRA=""; run PMR_FIR=mfma; run PMR_FIR=pair
EXTRA="--ctcss"; run PMR_FIR_DUAL=1; run PMR_FIR_DUAL=0
