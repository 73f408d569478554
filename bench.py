#!/usr/bin/env python3
"""bench.py -- complex-IQ Msamples/s through the full chain (BASELINE.json metric) on N MI355X.

One "step" = one pmr_chain_process_block_device() call: one block of synthetic cf32 IQ, already resident in
HBM, through dc-block -> resample -> NCO -> M-channel polyphase channelizer -> NBFM discriminator for all M
channels -> CTCSS high-pass -> gain -> de-emphasis -> int16 PCM (left in HBM).  The path shards by independent
IQ stream: rank r owns stream r on GPU r, no data-path collective (torch.distributed is used only for the
start/stop barrier and the max-over-ranks of the elapsed time).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (fs_in, M, default log2 block)  -- BASELINE.json configs
    "cfg2": (2.4e6, 16, 26),      # configs[1]: 16-ch PMR446 chain @ 2.4 MS/s on one MI355X (the metric's config)
    "cfg3": (61.44e6, 256, 26),   # configs[2]
    "cfg5": (1.0e9, 1024, 26),    # configs[4]
}
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def _oracle_loop(fs, M, block_host, seconds_target, only_channel=-1):
    import oracle
    n_probe = min(len(block_host), 1 << 20)
    n_avail = max(1, len(block_host) // n_probe)
    ch = oracle.OracleChain(fs_in=fs, num_channels=M, max_block=n_probe, only_channel=only_channel)
    ch.process_block(block_host[:n_probe], want=("pcm",))               # warm-up (page-in, caches)
    n_blocks = 0
    t0 = time.perf_counter()
    while True:                                                         # same stream continued; sample re-used cyclically
        b = n_blocks % n_avail
        ch.process_block(block_host[b * n_probe:(b + 1) * n_probe], want=("pcm",))
        n_blocks += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target:
            break
    ch.close()
    return n_blocks, n_probe, dt


_WORKER = """
import sys, time, numpy as np
sys.path.insert(0, %r)
import bench
fs, M, n, secs = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
from sdr_pmr446_amd import synth
x = synth.synth_iq(n, fs, M, stream_id=int(sys.argv[5]), channels=list(range(0, M, max(1, M // 16))))
nb, npr, dt = bench._oracle_loop(fs, M, x, secs)
print(nb * npr, dt)
"""


def cpu_baseline(fs, M, block_host, seconds_target=8.0):
    """Time the CPU oracle (kind 'port': the reference itself needs liquid-dsp and cannot be built here) on a
    bounded sample of the same workload: (i) single thread like the reference's DSP thread (src/sdr_pmr446.c:788) --
    the headline `value`; (ii) N independent streams on N cores, the CPU analogue of one stream per GPU;
    (iii) the reference's own semantics, one squelch-selected channel demodulated (:876-877)."""
    import subprocess
    import sys
    n_blocks, n_probe, dt = _oracle_loop(fs, M, block_host, seconds_target)
    total = n_blocks * n_probe
    out = {"value": total / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "%d blocks x %d samples of the same synthetic IQ, all %d channels demodulated, %.1f s" %
                     (n_blocks, n_probe, M, dt)}
    nb1, np1, dt1 = _oracle_loop(fs, M, block_host, seconds_target / 2, only_channel=0)
    out["one_channel"] = {"value": nb1 * np1 / dt1 / 1e6, "cores": 1,
                          "sample": "reference semantics: only the selected channel demodulated, %.1f s" % dt1}
    try:
        ncores = min(len(os.sched_getaffinity(0)), 32)
        if ncores > 1:
            procs = [subprocess.Popen([sys.executable, "-c", _WORKER % ROOT, str(fs), str(M), str(n_probe),
                                       str(seconds_target / 2), str(i)], stdout=subprocess.PIPE, cwd=ROOT)
                     for i in range(ncores)]
            rates = []
            for p in procs:
                o = p.communicate(timeout=120)[0].split()
                rates.append(float(o[0]) / float(o[1]))
            out["multi"] = {"value": sum(rates) / 1e6, "cores": ncores,
                            "sample": "%d independent streams, one single-threaded oracle per core" % ncores}
    except Exception as e:                                              # the single-thread figure stands on its own
        out["multi"] = {"value": None, "error": str(e)[:120]}
    return out


def load_measured_traffic(kernel, workload, block):
    """HBM bytes per launch of the roofline kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE doubled per MI355X_MICROARCH.md, + WRITE_SIZE), if they were taken for this workload/block."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        e = t.get("%s/%s/%d" % (workload, kernel, block))
        return e["hbm_bytes_per_launch"] if e else None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--log2-block", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real runs; gloo lets two ranks share one "
                                                          "GPU to exercise the N > 1 code path on a 1-GPU box")
    ap.add_argument("--host-io", action="store_true",
                    help="also time the host-buffer entry point (H2D of the IQ + D2H of the PCM inside the call)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-kernel HIP events in the timed region")
    args = ap.parse_args()

    import torch
    from sdr_pmr446_amd import chain as pmr
    from sdr_pmr446_amd import multigpu
    from sdr_pmr446_amd.synth_torch import synth_iq_torch

    rank, local_rank, world = multigpu.env_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    if args.dist_backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()          # self-test only: ranks may share a device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = multigpu.init_dist(args.dist_backend, dev)

    fs, M, lb = WORKLOADS[args.workload]
    lb = args.log2_block if args.log2_block is not None else lb
    block = 1 << lb

    ch = pmr.PmrChain(fs_in=fs, num_channels=M, max_block=block, device=local_rank)
    S = ch.max_frames
    iq = synth_iq_torch(block, fs, M, dev, stream_id=multigpu.stream_id_for_rank(rank))   # resident in HBM before timing
    pcm = torch.zeros((M, S), dtype=torch.int16, device=dev)        # PCM stays in HBM
    torch.cuda.synchronize()

    def step():
        return ch.process_block_device(iq.data_ptr(), block, d_pcm=pcm.data_ptr(), stride=S)

    for _ in range(args.warmup):
        step()
    ch.synchronize()
    ch.profile_reset()
    # HIP events in the timed region only around the roofline kernel (k_frontend): event records around all six
    # kernels of a step cost 6-16 % of the step time (measured); the full per-kernel breakdown is taken right after
    ch.profile_enable(0 if args.no_kernel_events else 2)

    def run():
        n = 0
        for _ in range(args.steps):
            n += step()
        ch.synchronize()
        return n

    dt, frames = multigpu.timed_region(run, dist, torch.cuda.synchronize, dev if args.dist_backend == "nccl" else None)
    ch.profile_enable(0)
    prof_roof = ch.profile()
    breakdown_steps = 0
    if not args.no_kernel_events:
        # separate pass, outside the timed region: every kernel, blocks NOT pipelined (uncontended kernel times)
        ch.set_overlap(False)
        ch.profile_reset()
        ch.profile_enable(1)
        breakdown_steps = min(args.steps, 5)
        for _ in range(breakdown_steps):
            step()
        ch.synchronize()
        ch.profile_enable(0)
        ch.set_overlap(True)

    prof = ch.profile()
    if rank == 0:
        r = M * 12500.0 / fs
        b_alg = 8.0 + 2.0 * r                                        # bytes per input sample
        value = multigpu.aggregate_throughput(world, args.steps, block, dt) / 1e6
        roof = None
        if prof_roof:
            name, (ms, n) = max(prof_roof.items(), key=lambda kv: kv[1][0])
            avg_s = ms / n * 1e-3
            launches_per_step = n / args.steps
            achieved = b_alg * block / launches_per_step / avg_s / 1e9
            roof = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "avg_kernel_ms": ms / n,
                    "launches_per_step": launches_per_step,
                    "algorithmic_bytes_per_sample": b_alg,
                    "kernel_symbols": "slot k_frontend = k_frontend_fast<MODE,N3,TAIL> (specialised cascades) or "
                                      "k_frontend<NT,SPT,MODE> in a rocprofv3 trace; k_fir_tm<hp> = k_fir_mfma16<...>",
                    "kernels_ms_per_step_isolated": {k: v[0] / max(1, breakdown_steps)
                                                     for k, v in sorted(prof.items())}}
            iso = prof.get(name)
            if iso and iso[1]:
                iso_s = iso[0] / iso[1] * 1e-3
                roof["isolated"] = {"avg_kernel_ms": iso[0] / iso[1], "achieved": b_alg * block / iso_s / 1e9,
                                    "frac": b_alg * block / iso_s / 1e9 / HBM_PEAK_GBPS,
                                    "note": "same kernel with block pipelining off (no other kernel sharing the GPU)"}
            roof["traffic"] = load_measured_traffic(name, args.workload, block)
        out = {
            "metric": "complex-IQ Msamples/s through full channelize+demod chain", "value": value,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d-ch PMR446 chain @ %.4g MS/s, one independent IQ stream per GPU" %
                                   (args.workload, M, fs / 1e6),
                       "block_samples": block, "frames_per_step": frames // max(1, args.steps),
                       "channels_demodulated": M, "hbm_frac_of_peak_whole_chain": value * 1e6 * b_alg / 1e9 / world / HBM_PEAK_GBPS},
            "roofline": roof,
        }
        if args.host_io and world == 1:
            # PCIe-inclusive rate of pmr_chain_process_block (host cf32 in, host int16 out), never the headline value
            import numpy as np
            nb = min(block, 1 << 22)                                     # 2^22 samples per call
            import ctypes
            res = {}
            pcm_h = torch.zeros((M, S), dtype=torch.int16)
            ns_c = ctypes.c_uint(0)

            def host_call(xn, pcm_t):
                rc = ch._L.pmr_chain_process_block(ch.h, xn.ctypes.data, nb, pcm_t.data_ptr(), S, ctypes.byref(ns_c),
                                                   None, None)
                assert rc == 0, rc

            for kind in ("pageable", "pinned"):
                xh = iq[:nb].cpu()
                if kind == "pinned":
                    xh = xh.pin_memory()
                xn = xh.numpy().view(np.complex64).reshape(-1)
                pcm_t = pcm_h.pin_memory() if kind == "pinned" else pcm_h
                host_call(xn, pcm_t)
                t0 = time.perf_counter()
                for _ in range(8):
                    host_call(xn, pcm_t)
                res[kind] = 8 * nb / (time.perf_counter() - t0) / 1e6
            out["host_io"] = {"unit": "Msamples/s", "block_samples": nb, **res,
                              "note": "synchronous pmr_chain_process_block: H2D + chain + D2H per call, no overlap"}
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = min(block, 1 << 24)
            out["cpu_baseline"] = cpu_baseline(fs, M, iq[:n_cpu].cpu().numpy())
        print(json.dumps(out), flush=True)
    ch.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
