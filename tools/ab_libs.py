#!/usr/bin/env python3
"""Same-box A/B of several BUILDS of the library (box-to-box spread is +-3 %, more than most single changes): every (build, workload)
pair is timed in a fresh process, the builds interleaved inside every repetition; the table gives the median rate per pair and its
ratio to the first build.

  GPU box:  python3 tools/ab_libs.py --libs base=,exp1=build_ab/exp1/libpmr446_hip.so --workloads cfg2,cfg3,cfg5 --reps 3
            ("" = the in-tree build; builds are made HERE with `python3 sdr_pmr446_amd/build.py --variant NAME "-DFLAGS"`)
  one leg:  python3 tools/ab_libs.py --leg cfg2        (PMR_LIBRARY selects the build; prints one number)

A leg is bench.py's timed loop without its extras: 4 distinct device-resident 2^26-sample blocks rotated through un-synchronised
pmr_chain_process_block_device calls, `--regions` regions of `--steps` steps, median region.  `--env K=V,...` exports variables to
every leg (e.g. PMR_OVERLAP=0 for un-pipelined kernels); `--ctcss` / `--one-open` select those modes."""
import argparse, os, statistics, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORK = {"cfg2": (2.4e6, 16), "cfg3": (61.44e6, 256), "cfg5": (1.0e9, 1024)}


def leg(args):
    os.environ["PMR_NO_TORCH"] = "1"
    from sdr_pmr446_amd import chain, synth
    fs, M = WORK[args.leg]
    lb, rot = args.log2_block, 4
    block = 1 << lb
    ch = chain.PmrChain(fs_in=fs, num_channels=M, max_block=block, resamp_As=args.resamp_as)
    if args.show_plan:
        sys.stderr.write("%s resamp_As %.1f: front-end plan %d (0 staged, 1 generic tile kernel, 2 specialised one level, 3 two levels specialised, "
                         "4 two levels generic), stages m = %s\n" % (args.leg, args.resamp_as, ch.info(8), [ch.info(1, i) for i in range(ch.info(0))]))
    if args.ctcss:
        ch.ctcss_enable()
    if args.one_open:
        ch.set_channel_mask([synth.signal_channels(M, fs)[0]])
    S = ch.max_frames
    iq = chain.synth_iq_device(rot * block, fs, M, period_log2=lb + 2)
    pcm = chain.DeviceBuffer(M * S * 2)
    chain.device_synchronize()
    pos = 0
    for _ in range(6):
        ch.process_block_device(iq.ptr + (pos % rot) * block * 8, block, d_pcm=pcm.ptr, stride=S); pos += 1
    ch.synchronize()
    dts = []
    for _ in range(args.regions):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ch.process_block_device(iq.ptr + (pos % rot) * block * 8, block, d_pcm=pcm.ptr, stride=S); pos += 1
        ch.synchronize()
        dts.append(time.perf_counter() - t0)
    print("%.2f %.2f %.2f" % tuple(args.steps * block / d / 1e9 for d in (statistics.median(dts), max(dts), min(dts))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="base=")
    ap.add_argument("--workloads", default="cfg2,cfg3,cfg5")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--regions", type=int, default=7)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--log2-block", type=int, default=26)
    ap.add_argument("--env", default="")
    ap.add_argument("--ctcss", action="store_true")
    ap.add_argument("--one-open", action="store_true")
    ap.add_argument("--resamp-as", type=float, default=60.0, help="msresamp stop-band (60 = the reference, :426); another value designs "
                    "another cascade: what the plans that miss the specialised kernels cost (SURVEY A.3 risk)")
    ap.add_argument("--show-plan", action="store_true")
    ap.add_argument("--leg", default=None)
    args = ap.parse_args()
    if args.leg:
        return leg(args)
    libs = [kv.split("=", 1) for kv in args.libs.split(",") if kv]
    works = [w for w in args.workloads.split(",") if w]
    res = {}
    base_env = dict(os.environ)
    for kv in args.env.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1); base_env[k] = v
    for rep in range(args.reps):
        for w in works:
            for name, path in list(libs):
                env = dict(base_env)
                if path:
                    env["PMR_LIBRARY"] = os.path.join(ROOT, path) if not os.path.isabs(path) else path
                else:
                    env.pop("PMR_LIBRARY", None)
                cmd = [sys.executable, os.path.abspath(__file__), "--leg", w, "--regions", str(args.regions), "--steps", str(args.steps),
                       "--log2-block", str(args.log2_block), "--resamp-as", str(args.resamp_as)] + (["--ctcss"] if args.ctcss else []) + \
                      (["--one-open"] if args.one_open else []) + (["--show-plan"] if args.show_plan else [])
                # a leg takes ~12 s; one that does not come back (an experiment build with an impossible parameter can hang a kernel: round 6
                # lost ten GPU-minutes to -DPW_FPW1024=1) is given up after 150 s instead of taking the whole comparison down
                try:
                    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
                    v = float(r.stdout.split()[0])
                except subprocess.TimeoutExpired:
                    v = float("nan")
                    sys.stderr.write("leg %s/%s TIMED OUT (150 s): build dropped from the remaining repetitions\n" % (name, w))
                    libs = [kv for kv in libs if kv[0] != name or kv is libs[0]]
                    r = None
                except Exception:
                    v = float("nan")
                    sys.stderr.write("leg %s/%s failed: %s\n" % (name, w, (r.stderr or r.stdout)[-300:]))
                if args.show_plan and rep == 0 and r is not None:
                    sys.stderr.write(r.stderr)
                res.setdefault((name, w), []).append(v)
    print("%-14s" % "build" + "".join("%26s" % w for w in works))
    for name, _ in libs:
        line = "%-14s" % name
        for w in works:
            v = res[(name, w)]
            med = statistics.median(v)
            b = statistics.median(res[(libs[0][0], w)])
            line += "  %6.1f (%+5.1f %%) [%s]" % (med, 100.0 * (med / b - 1.0), " ".join("%.0f" % x for x in v))
        print(line)


if __name__ == "__main__":
    main()
