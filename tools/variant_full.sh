#!/bin/bash
# usage (GPU box): bash tools/variant_full.sh <workload> "<extra hipcc flags>" ... : like variant_bench.sh, but also the one-open-channel
# leg and the front end's un-pipelined duration
W=$1; shift
for F in "$@"; do
  echo "== flags: $F"
  # the -D flags reach BOTH compilers (hipcc for the kernels, gcc for the host C: a macro that lives in pmr_chain.c was silently
  # ignored in round 4 -- ADVICE r04); other flags are hipcc's only
  CCF=$(for t in $F; do case $t in -D*) echo -n "$t ";; esac; done)
  if ! PMR_HIPCC_FLAGS="-fno-slp-vectorize $F" PMR_CC_FLAGS="$CCF" python3 sdr_pmr446_amd/build.py --force > /tmp/variant_build.log 2>&1; then
    echo "BUILD FAILED for flags: $F"; grep -m3 -E "error" /tmp/variant_build.log; continue
  fi
  python3 bench.py --workload $W --also none --no-cpu-baseline --no-host-io --regions 5 --parity-blocks 0 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value %.1f GS/s  one open channel %.1f  front end in the loop %.4f ms, alone %.4f ms' % (d['value']/1e3, d['one_open_channel']['value']/1e3, r['avg_kernel_ms'], r['isolated']['avg_kernel_ms']))
"
done
