#!/usr/bin/env python3
"""Turn gpurun_out/prof (tools/collect_profiles.sh) into the committed summaries under profiles/ for round `tag`:
   <tag>_bench.json                        the bench line (headline cfg5 + cfg2 / cfg3 sub-records)
   <tag>_kernel_stats_<cfg>.csv            rocprofv3 --kernel-trace: calls / avg / min / max per kernel, pipelined bench loop
   <tag>_pmc_fetch_write_<cfg>.json        FETCH_SIZE / WRITE_SIZE (KB) per launch and kernel, + HBM bytes (FETCH doubled, see below)
   <tag>_pmc_insts_<cfg>.json              SQ_INSTS_* per launch of the front-end kernels (blocks not pipelined)
   traffic.json                            what bench.py reports as roofline.traffic"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, O = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
BLOCK = 1 << 26
BALG = {"cfg2": 8.0 + 2.0 * 16 * 12500 / 2.4e6, "cfg3": 8.0 + 2.0 * 256 * 12500 / 61.44e6, "cfg5": 8.0 + 2.0 * 1024 * 12500 / 1.0e9}


def short(n):
    return n.split("(")[0].replace("void ", "").strip()


def newest(sub, pat):
    f = sorted(glob.glob(os.path.join(P, sub, "**", pat), recursive=True), key=os.path.getmtime, reverse=True)
    return f[0] if f else None


src = os.path.join(P, "bench.json")
if os.path.exists(src) and os.path.getsize(src):
    d = json.loads(open(src).read())
    json.dump(d, open(os.path.join(O, "%s_bench.json" % tag), "w"), indent=1)
    print("bench: %s %.1f GS/s frac %.3f" % (d["config"]["workload"][:4], d["value"] / 1e3, d["roofline"]["frac"]))

hp = os.path.join(P, "kernel_sources.sha256")
KHASH = open(hp).read().strip() if os.path.exists(hp) else None
tj = os.path.join(O, "traffic.json")
traffic = json.load(open(tj)) if os.path.exists(tj) else {}
for w in ("cfg5", "cfg3", "cfg2"):
    tr = newest("stats_" + w, "*kernel_trace.csv")
    if tr:
        agg = {}
        for r in csv.DictReader(open(tr)):
            agg.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        tot = sum(sum(v) for v in agg.values())
        with open(os.path.join(O, "%s_kernel_stats_%s.csv" % (tag, w)), "w") as f:
            f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
            for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                f.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)))
        print(w, {n: round(sum(v) / len(v) / 1e3, 1) for n, v in agg.items()})
    pm = {}
    for cname, sub in (("FETCH_SIZE", "fetch_" + w), ("WRITE_SIZE", "write_" + w)):
        cc = newest(sub, "*counter_collection.csv")
        if not cc:
            continue
        acc = {}
        for r in csv.DictReader(open(cc)):
            if r["Counter_Name"] == cname:
                acc.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
        for n, v in acc.items():
            pm.setdefault(n, {})[cname + "_KB"] = sum(v) / len(v)
    if pm:
        chain_bytes = 0.0
        # the bench command also runs its one-open-channel leg (pmr_chain_set_channel_mask): the GATHER form of the audio FIR appears in
        # the trace but is not part of the all-channel chain whose bytes are compared with the algorithmic figure
        mask_only = ("k_fir_mfma4<true", "k_fir_mfma16<false, 1, true, true")
        for n, e in pm.items():
            if "FETCH_SIZE_KB" in e and "WRITE_SIZE_KB" in e:
                # gfx950: FETCH_SIZE counts a 128-byte request of a wide streaming read as 64 B (MI355X_MICROARCH.md, HBM) -> doubled
                e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE_KB"] + e["WRITE_SIZE_KB"]) * 1024.0
                if n.startswith(mask_only):
                    e["note"] = "masked (one open channel) leg of the bench command: not counted in _chain"
                else:
                    chain_bytes += e["hbm_bytes_per_launch"]
        alg = BALG[w] * BLOCK
        pm["_chain"] = {"hbm_bytes_per_block": chain_bytes, "algorithmic_bytes_per_block": alg, "ratio": chain_bytes / alg}
        json.dump(pm, open(os.path.join(O, "%s_pmc_fetch_write_%s.json" % (tag, w)), "w"), indent=1)
        fe = [n for n in pm if n.startswith("k_fe_fast") or n.startswith("k_frontend")]
        if fe and "hbm_bytes_per_launch" in pm[fe[0]]:
            e = pm[fe[0]]
            traffic["%s/k_frontend/%d" % (w, BLOCK)] = {
                "FETCH_SIZE_KB": e["FETCH_SIZE_KB"], "WRITE_SIZE_KB": e["WRITE_SIZE_KB"],
                "hbm_bytes_per_launch": e["hbm_bytes_per_launch"], "kernel": fe[0], "round": tag,
                "kernel_sources_sha256": KHASH,
                "chain_hbm_bytes_per_block": chain_bytes, "chain_over_algorithmic": chain_bytes / alg,
                "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); FETCH_SIZE doubled per "
                        "MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B for 16-B/lane streams); algorithmic bytes per "
                        "launch = %.4f * 2^26 = %.1fe6" % (BALG[w], alg / 1e6)}
        print(w, "front end %.1f MB/launch, chain %.1f MB = %.2fx algorithmic" % (pm[fe[0]].get("hbm_bytes_per_launch", 0) / 1e6 if fe else 0, chain_bytes / 1e6, chain_bytes / alg))
    cc = newest("insts_" + w, "*counter_collection.csv")
    if cc:
        acc = {}
        for r in csv.DictReader(open(cc)):
            acc.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        out = {}
        for n, dct in acc.items():
            e = {c: sum(v) / len(v) for c, v in dct.items()}
            if e.get("SQ_WAVES"):
                e["per_wave"] = {c: e[c] / e["SQ_WAVES"] for c in e if c.startswith("SQ_INSTS")}
            out[n] = e
        json.dump(out, open(os.path.join(O, "%s_pmc_insts_%s.json" % (tag, w)), "w"), indent=1)
        for n, e in out.items():
            if n.startswith("k_fe_"):
                print(w, n, {c: round(v) for c, v in e.get("per_wave", {}).items()})
json.dump(traffic, open(tj, "w"), indent=1)
