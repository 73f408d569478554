#!/bin/bash
# usage (GPU box): bash tools/variant_kstats.sh <workload> "<extra hipcc flags>" ... : rebuild with each flag set, kernel durations (us) of
# one bench run with blocks NOT pipelined (rocprofv3 kernel trace)
W=$1; shift
for F in "$@"; do
  PMR_HIPCC_FLAGS="-fno-slp-vectorize $F" python3 sdr_pmr446_amd/build.py --force > /dev/null 2>&1
  echo "== flags: $F"
  PMR_OVERLAP=0 bash tools/kstats.sh vk_tmp.txt --workload $W --also none --no-cpu-baseline --regions 2 --parity-blocks 0 --no-kernel-events
  head -6 gpurun_out/vk_tmp.txt
done
