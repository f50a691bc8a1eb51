#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of scripts/profile.sh (gpurun_out/prof_<tag>_<workload>/) into the committed summary
profiles/<name>/{kernel_stats[_wl].csv, pmc_summary[_wl].json, bench_line[_wl].json}:
   python scripts/summarize_profile.py <tag> <name> [workload]"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
wl = sys.argv[3] if len(sys.argv) > 3 else "nsq24"
sfx = "" if wl == "nsq24" else "_" + wl
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag + "_" + wl)
dst = os.path.join(ROOT, "profiles", name)
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"kernel_stats{sfx}.csv"))
summary = {}
for p in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
    f = glob.glob(os.path.join(src, p, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k, v in per.items():
        summary.setdefault(k, {})[p] = {"dispatches": len(disp[k]), "sums": dict(v)}
line = None
for lg in ("trace_bench.log", "pmc_sq.log"):
    try:
        for ln in open(os.path.join(src, lg)):
            if ln.startswith("{") and '"metric"' in ln:
                line = json.loads(ln)
                if lg == "trace_bench.log":
                    json.dump(line, open(os.path.join(dst, f"bench_line{sfx}.json"), "w"), indent=1)
                break
    except OSError:
        pass
    if lg == "pmc_sq.log" and line:
        n_pmc = line["indices"]["n"] + 0
        # the PMC command is `bench.py --steps 1 --warmup 0 --no-time-to-cov`: exactly one 1e6-scenario launch
        ttc = line.get("time_to_cov_1pct", {}).get("samples", 0)
        scen = n_pmc + ttc
        summary["units_per_profiled_launch"] = scen
        ev = [k for k in summary if "eval_kernel" in k and "pmc_fetch" in summary[k] and "pmc_write" in summary[k] and "pmc_sq" in summary[k]]
        ev.sort(key=lambda k: -summary[k]["pmc_sq"]["sums"].get("SQ_INSTS_VALU", 0.0))      # the production kernel, not the calibration probe
        if ev:
            fk = summary[ev[0]]["pmc_fetch"]["sums"]["FETCH_SIZE"]; wk = summary[ev[0]]["pmc_write"]["sums"]["WRITE_SIZE"]
            b = (2.0 * fk + wk) * 1024.0
            summary["hbm_traffic"] = {
                "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0 --no-time-to-cov` (one 1e6-scenario launch); units KiB; FETCH_SIZE "
                        "doubled per MI355X_MICROARCH.md (gfx950 reports half of wide reads), WRITE_SIZE uncalibrated",
                "fetch_kib": fk, "write_kib": wk, "scenarios": scen, "bytes_per_scenario": b / scen, "bytes_per_1e6_scenario_launch": b / scen * 1e6}
# what ties the summary to a binary and to a launch duration (bench.py sets roofline.counters_stale from these): the code-object hash the
# PROFILED bench line printed about the library it ran, and the shortest launch of the production kernel in the kernel trace (the average
# of a handful of launches carries the cold first one)
hashes = set()
for lg in ("trace_bench.log", "pmc_sq.log", "pmc_lds.log", "pmc_fetch.log", "pmc_write.log"):
    try:
        for ln in open(os.path.join(src, lg)):
            if ln.startswith("{") and '"metric"' in ln:
                j = json.loads(ln)
                if "binary" in j:
                    hashes.add(j["binary"]["code_object_sha256"])
                if lg == "trace_bench.log":
                    summary["units_per_traced_launch"] = j["roofline"]["units_per_launch"]
    except OSError:
        pass
if len(hashes) == 1:
    summary["code_object_sha256"] = hashes.pop()
elif hashes:
    summary["code_object_sha256_conflict"] = sorted(hashes)       # the passes ran different binaries: no hash, the line will say stale
if stats:
    rows = [r for r in csv.DictReader(open(stats[0])) if "eval_kernel" in r["Name"]]
    if rows:
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        summary["kernel_ms_min"] = float(top["MinNs"]) * 1e-6
        summary["kernel_ms_avg_traced"] = float(top["AverageNs"]) * 1e-6
        summary["kernel_traced"] = top["Name"].split("(")[0]
json.dump(summary, open(os.path.join(dst, f"pmc_summary{sfx}.json"), "w"), indent=1)
print("wrote", dst, sorted(os.listdir(dst)))
