for S in 0 2 4; do echo "stagger $S"; PMR_FE_STAGGER=$S bash tools/quick_bench.sh cfg2 2>&1 | tail -1; done
