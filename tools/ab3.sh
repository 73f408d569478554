#!/bin/bash
# usage (GPU box): bash tools/ab3.sh <tag> [ENV=VAL ...] : quick bench of cfg2 / cfg3 / cfg5 with the given environment, one line each
tag=$1; shift
for w in cfg2 cfg3 cfg5; do
  echo -n "$tag $* $w: "
  env "$@" bash tools/quick_bench.sh $w
done
