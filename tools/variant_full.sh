#!/bin/bash
# usage (GPU box): bash tools/variant_full.sh <workload> "<extra hipcc flags>" ... : like variant_bench.sh, but also the one-open-channel
# leg and the front end's un-pipelined duration
W=$1; shift
for F in "$@"; do
  echo "== flags: $F"
  # the build goes to build_ab/_variant/ (never the in-tree product library) with -DPMR_EXPERIMENT added by build.py; the -D flags
  # reach BOTH compilers (hipcc for the kernels, gcc for the host C)
  if ! python3 sdr_pmr446_amd/build.py --variant _variant "$F" > /tmp/variant_build.log 2>&1; then
    echo "BUILD FAILED for flags: $F"; grep -m3 -E "error" /tmp/variant_build.log; continue
  fi
  export PMR_LIBRARY=$PWD/build_ab/_variant/libpmr446_hip.so
  python3 bench.py --allow-experiment --workload $W --also none --no-cpu-baseline --no-host-io --regions 5 --parity-blocks 0 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value %.1f GS/s  one open channel %.1f  front end in the loop %.4f ms, alone %.4f ms' % (d['value']/1e3, d['one_open_channel']['value']/1e3, r['avg_kernel_ms'], r['isolated']['avg_kernel_ms']))
"
done
