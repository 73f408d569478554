#!/bin/bash
# usage (GPU box): bash tools/variant_kstats.sh <workload> "<extra hipcc flags>" ... : rebuild with each flag set, kernel durations (us) of
# one bench run with blocks NOT pipelined (rocprofv3 kernel trace)
W=$1; shift
for F in "$@"; do
  echo "== flags: $F"
  # (a build that fails must not fall through to the previous flag set's library: round 4 lost four "baselines" to a -DX=1 that
  #  collided with a variable named X.  Use -DPMR_BASELINE for "no change".)
  # the -D flags reach BOTH compilers (hipcc for the kernels, gcc for the host C: a macro that lives in pmr_chain.c was silently
  # ignored in round 4 -- ADVICE r04); other flags are hipcc's only
  CCF=$(for t in $F; do case $t in -D*) echo -n "$t ";; esac; done)
  if ! PMR_HIPCC_FLAGS="-fno-slp-vectorize $F" PMR_CC_FLAGS="$CCF" python3 sdr_pmr446_amd/build.py --force > /tmp/variant_build.log 2>&1; then
    echo "BUILD FAILED for flags: $F"; grep -m3 -E "error" /tmp/variant_build.log; continue
  fi
  PMR_OVERLAP=0 bash tools/kstats.sh vk_tmp.txt --workload $W --also none --no-cpu-baseline --no-host-io --regions 2 --parity-blocks 0 --no-kernel-events
  head -6 gpurun_out/vk_tmp.txt
done
