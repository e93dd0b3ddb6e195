"""Copy the judged part of a `scripts/gpu_batch.sh <tag> profiles` run into profiles/rN/: per directory kernel_stats.csv
(rocprofv3's --stats summary), bench_under_rocprof.json, and per counter pass the per-kernel aggregates of the per-dispatch
counter CSV (Kernel_Name, Counter_Name, Dispatches, Counter_Value_Sum, Counter_Value_Avg) -- the per-dispatch files are tens of MB.

    python scripts/profile_collect.py gpurun_out/<tag> profiles/r2
then  python scripts/profile_summary.py profiles/r2/summary.json '<key>|profiles/r2/<dir>' ...  (printed at the end)"""
import csv
import glob
import os
import shutil
import sys

KEYS = {"lapl_3375": "lapl_3375x3375", "lapl_3375_levels": "lapl_3375x3375:program=0", "gen_40_6": "gen:40:6", "gen_60_8": "gen:60:8",
        "gen_60_8_mixed": "gen:60:8:mixed", "gen_100_10_mixed": "gen:100:10:mixed", "gen_100_10": "gen:100:10"}


def short(name):
    return name.split("(")[0].split("<")[0].replace("void ", "").strip()


src, dst = sys.argv[1], sys.argv[2]
specs = []
for d in sorted(os.listdir(src)):
    sd = os.path.join(src, d)
    if not os.path.isdir(sd) or not glob.glob(os.path.join(sd, "stats", "*kernel_stats.csv")):
        continue
    dd = os.path.join(dst, d)
    os.makedirs(dd, exist_ok=True)
    shutil.copy(glob.glob(os.path.join(sd, "stats", "*kernel_stats.csv"))[0], os.path.join(dd, "kernel_stats.csv"))
    if os.path.exists(os.path.join(sd, "bench_under_rocprof.json")):
        shutil.copy(os.path.join(sd, "bench_under_rocprof.json"), os.path.join(dd, "bench_under_rocprof.json"))
    for p in sorted(glob.glob(os.path.join(sd, "pmc_*", "*counter_collection.csv"))):
        name = os.path.basename(os.path.dirname(p))  # pmc_SQ, pmc_FETCH_SIZE, pmc_WRITE_SIZE
        acc = {}
        for r in csv.DictReader(open(p)):
            s = acc.setdefault((short(r["Kernel_Name"]), r["Counter_Name"]), [0, 0.0])
            s[0] += 1
            s[1] += float(r["Counter_Value"])
        with open(os.path.join(dd, name + "_by_kernel.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Counter_Value_Sum", "Counter_Value_Avg"])
            for (k, c), (n, v) in sorted(acc.items()):
                w.writerow([k, c, n, v, f"{v / n:.3f}"])
    if d in KEYS:
        specs.append(f"'{KEYS[d]}|{dd}'")
print("python scripts/profile_summary.py", os.path.join(dst, "summary.json"), " ".join(specs))
