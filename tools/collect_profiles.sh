#!/bin/bash
# Run on the GPU box (gpurun): the bench line and the rocprofv3 passes whose summaries are committed under profiles/.
#   gpurun --timeout 1800 -- 'bash tools/collect_profiles.sh'   then (here)   python3 tools/summarize_profiles.py rNN
# Every rocprofv3 command has the program (python3 bench.py) directly after `--`; the PMC passes are separate runs with
# --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots: FETCH_SIZE and WRITE_SIZE do not fit one pass).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
# the kernels these counters belong to (profiles/traffic.json carries the hash; bench.py marks roofline.traffic stale when the
# tree it runs from has other kernel sources)
python3 $R/sdr_pmr446_amd/build.py --kernel-hash > $O/kernel_sources.sha256
cd /tmp && export TMPDIR=/tmp
[ -n "$SKIP_BENCH" ] || python3 $R/bench.py 2>/dev/null | grep '^{' > $O/bench.json
B="--also none --regions 2 --steps 50 --warmup 2 --no-cpu-baseline --no-host-io --no-kernel-events --parity-blocks 0 --no-one-open"
for W in ${WORKLOADS:-cfg5 cfg3 cfg2}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -- python3 $R/bench.py --workload $W $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$W -- python3 $R/bench.py --workload $W $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$W -- python3 $R/bench.py --workload $W $B > /dev/null 2>&1
done
# instruction mix of the front-end kernels (blocks not pipelined: clean attribution)
export PMR_OVERLAP=0
for W in ${INST_WORKLOADS:-cfg5 cfg2}; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $O/insts_$W -- python3 $R/bench.py --workload $W $B > /dev/null 2>&1
done
unset PMR_OVERLAP
# two ranks on this one GPU (gloo for the barrier / max): the N > 1 code path of bench.py for the record -- NOT a scaling figure
[ -n "$SKIP_BENCH" ] || python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 2 \
    --dist-backend gloo --no-cpu-baseline --no-host-io --regions 5 2>/dev/null | grep '^{' > $O/bench_2rank_1gpu.json
# keep the merged-back payload small: only this library's kernels
for f in $(find $O -name '*_kernel_trace.csv' -o -name '*_counter_collection.csv'); do
  (head -1 $f; grep -E 'k_frontend|k_fe_|k_channelize|k_pfb|k_fft|k_fir|k_rssi|k_ct_|k_dsd|k_iq|k_poison' $f) > $f.trim; mv $f.trim $f
done
find $O -name '*agent_info.csv' -delete; find $O -name '*_stats.csv' -delete
du -sh $O
