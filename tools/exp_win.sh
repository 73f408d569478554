for m in pair win; do echo SMALL=$m; PMR_CHANNELIZER_SMALL=$m bash tools/quick_bench.sh cfg2; done
python3 -m pytest tests -m gpu -q -x 2>&1 | tail -1
