#!/usr/bin/env python3
"""Turn gpurun_out/prof (tools/collect_profiles.sh) into the committed summaries under profiles/ for round `tag`."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, O = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
OURS = ("k_frontend", "k_channelize", "k_fir", "k_fe_", "k_rssi", "k_ct_", "k_dsd")


def short(n):
    return n.split("(")[0].replace("void ", "").strip()


for w in ("cfg2", "cfg3", "cfg5"):
    src = os.path.join(P, "bench_%s.json" % w)
    if os.path.exists(src) and os.path.getsize(src):
        d = json.loads(open(src).read())
        json.dump(d, open(os.path.join(O, "%s_bench%s.json" % (tag, "" if w == "cfg2" else "_" + w)), "w"), indent=1)
        print(w, "%.1f GS/s" % (d["value"] / 1e3), "frac", round(d["roofline"]["frac"], 3))

# kernel stats from the trace (same numbers rocprofv3 --stats prints, restricted to this library's kernels)
tr = sorted(glob.glob(os.path.join(P, "stats", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime, reverse=True)
if tr:
    agg = {}
    for r in csv.DictReader(open(tr[0])):
        n = short(r["Kernel_Name"])
        if any(k in n for k in OURS):
            agg.setdefault(n, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values())
    with open(os.path.join(O, "%s_kernel_stats.csv" % tag), "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)))
    print(open(os.path.join(O, "%s_kernel_stats.csv" % tag)).read())

# PMC: average counter value per launch and kernel (KB)
pm = {}
for cname, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    cc = sorted(glob.glob(os.path.join(P, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    if not cc:
        continue
    acc = {}
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] != cname:
            continue
        acc.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    for n, v in acc.items():
        pm.setdefault(n, {})[cname] = sum(v) / len(v)
if pm:
    json.dump(pm, open(os.path.join(O, "%s_pmc_fetch_write_per_kernel.json" % tag), "w"), indent=1)
    fe = [n for n in pm if n.startswith("k_frontend")]
    if fe and "FETCH_SIZE" in pm[fe[0]] and "WRITE_SIZE" in pm[fe[0]]:
        e = pm[fe[0]]
        tj = os.path.join(O, "traffic.json")
        t = json.load(open(tj)) if os.path.exists(tj) else {}
        t["cfg2/k_frontend/67108864"] = {
            "FETCH_SIZE_KB": e["FETCH_SIZE"], "WRITE_SIZE_KB": e["WRITE_SIZE"],
            "hbm_bytes_per_launch": (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0, "kernel": fe[0], "round": tag,
            "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); FETCH_SIZE doubled "
                    "per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B for 16-B/lane streams); algorithmic "
                    "bytes/launch = 8.167 * 2^26 = 548.1e6"}
        json.dump(t, open(tj, "w"), indent=1)
        print("traffic", t["cfg2/k_frontend/67108864"]["hbm_bytes_per_launch"] / 1e6, "MB per launch")
