#!/bin/bash
# Run on the GPU box (gpurun): the bench lines and rocprofv3 passes whose summaries are committed under profiles/.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'   then   python3 tools/summarize_profiles.py rNN
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 30 --host-io 2>/dev/null | grep '^{' > $O/bench_cfg2.json
python3 $R/bench.py --workload cfg3 --steps 20 2>/dev/null | grep '^{' > $O/bench_cfg3.json
python3 $R/bench.py --workload cfg5 --steps 20 2>/dev/null | grep '^{' > $O/bench_cfg5.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
# keep the merged-back payload small: only this library's kernels
for f in $(find $O -name '*_kernel_trace.csv' -o -name '*_counter_collection.csv'); do
  (head -1 $f; grep -E 'k_frontend|k_channelize|k_fir|k_fe_|k_rssi|k_ct_|k_dsd' $f) > $f.trim; mv $f.trim $f
done
find $O -name '*agent_info.csv' -delete
ls -la $O $O/*/* | head -30
