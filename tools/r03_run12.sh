mkdir -p gpurun_out
python3 bench.py --workload cfg2 --also none --no-cpu-baseline --regions 5 --parity-blocks 0 --ctcss 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('ctcss: all %.1f GS/s one-open %.1f GS/s isolated' % (d['value']/1e3, d['one_open_channel']['value']/1e3), {k:round(v,4) for k,v in r['kernels_ms_per_step_isolated'].items()})
" > gpurun_out/r12.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r12prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg2 --also none --regions 1 --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-events --parity-blocks 0 --ctcss > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r12prof -name '*kernel_stats.csv' | head -1); head -12 $f | cut -c1-60,200-400 >> gpurun_out/r12.txt; rm -rf gpurun_out/r12prof
cat gpurun_out/r12.txt
