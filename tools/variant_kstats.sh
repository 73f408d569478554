#!/bin/bash
# usage (GPU box): bash tools/variant_kstats.sh <workload> "<extra hipcc flags>" ... : build a variant (build_ab/_variant/) with each flag set, kernel durations (us) of
# one bench run with blocks NOT pipelined (rocprofv3 kernel trace)
W=$1; shift
for F in "$@"; do
  echo "== flags: $F"
  # the build goes to build_ab/_variant/ (never the in-tree product library) with -DPMR_EXPERIMENT added by build.py; the -D flags
  # reach BOTH compilers (hipcc for the kernels, gcc for the host C)
  if ! python3 sdr_pmr446_amd/build.py --variant _variant "$F" > /tmp/variant_build.log 2>&1; then
    echo "BUILD FAILED for flags: $F"; grep -m3 -E "error" /tmp/variant_build.log; continue
  fi
  export PMR_LIBRARY=$PWD/build_ab/_variant/libpmr446_hip.so
  PMR_OVERLAP=0 bash tools/kstats.sh vk_tmp.txt --workload $W --also none --no-cpu-baseline --no-host-io --regions 2 --parity-blocks 0 --no-kernel-events --allow-experiment
  head -6 gpurun_out/vk_tmp.txt
done
