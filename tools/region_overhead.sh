#!/bin/bash
# usage (GPU box): bash tools/region_overhead.sh [workloads...] : time of a K-step region T(K) = a + b K -- what a timed region pays once (pipeline fill /
# drain, whatever else) beside its steady-state step.  bench.py's driver command uses 20-step regions.
for W in ${@:-cfg2 cfg3 cfg5}; do
  for K in 1 2 3 5 10 20 40 100; do
    python3 tools/ab_libs.py --leg $W --steps $K --regions 25 | awk -v k=$K -v w=$W '{ t = k * 67108864 / ($1 * 1e9) * 1e6; printf "%s K=%3d  median region %8.1f us  (%.1f us per step; fastest region %.1f)\n", w, k, t, t / k, k * 67108864 / ($3 * 1e9) * 1e6 }'
  done
done
