"""Build profiles/rN/summary.json -- every number bench.py's `roofline` object quotes from a profile -- out of the
rocprofv3 passes of one gpu_batch.sh run (or several).

    python scripts/profile_summary.py <out.json> '<run-key>|<batch dir>' ['<run-key>|<batch dir>' ...]

run-key = the bench case (`lapl_3375x3375`, `gen:40:6`, `gen:100:10:mixed`, ...).  A batch dir holds (any subset of)
    stats*/   *_kernel_stats.csv                      rocprofv3 --kernel-trace --stats
    pmc_SQ*/  *_counter_collection.csv                rocprofv3 --pmc SQ_... (one pass)
    pmc_FETCH_SIZE*/, pmc_WRITE_SIZE*/                separate passes, as MI355X_MICROARCH.md prescribes
Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE / WRITE_SIZE are in KiB and
FETCH_SIZE reports half of the bytes of wide coalesced reads, so hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024.
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs), the formula of rocprofiler-sdk's
derived_counters.xml (gfx94x section: the guide notes ROCm 7.2 ships no gfx950 one), summed over the launches of the kernel,
with GRBM_GUI_ACTIVE divided by 8: rocprofv3 reports it summed over the eight XCDs (checked against the executed flops of
k_update_mt: 512 * MOPS / duration / 78.6 TF/s = 0.476 where the corrected busy fraction says 0.48);
mfma_flops_per_launch = 512 * SQ_INSTS_VALU_MFMA_MOPS_F64 (the counter's unit) = flops EXECUTED on the matrix cores."""
import csv
import glob
import json
import os
import sys


def short(name):
    return name.split("(")[0].split("<")[0].replace("void ", "").strip()


def counters(path):
    acc = {}
    if path.endswith("_by_kernel.csv"):  # the per-kernel aggregates tracked under profiles/rN/ (Dispatches, Counter_Value_Sum)
        for r in csv.DictReader(open(path)):
            acc.setdefault(r["Kernel_Name"], {})[r["Counter_Name"]] = [int(r["Dispatches"]), float(r["Counter_Value_Sum"])]
        return acc
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        c = acc.setdefault(k, {})
        s = c.setdefault(r["Counter_Name"], [0, 0.0])
        s[0] += 1
        s[1] += float(r["Counter_Value"])
    return acc


def one_run(d, suffix=""):
    out = {"kernels": {}, "source": f"{d}: kernel_stats.csv = rocprofv3 --kernel-trace --stats; pmc_*_by_kernel.csv = per-kernel sums of the rocprofv3 --pmc "
                                    f"counter CSVs (one pass for the SQ/GRBM counters, FETCH_SIZE and WRITE_SIZE in passes of their own); the bench line "
                                    f"printed under the profiler is bench_under_rocprof.json (scripts/gpu_batch.sh profiles)"}
    ks = out["kernels"]
    for f in sorted(glob.glob(os.path.join(d, f"stats{suffix}", "*kernel_stats.csv")) + glob.glob(os.path.join(d, "kernel_stats.csv"))):
        for r in csv.DictReader(open(f)):  # (the instances of a template -- k_solve_panel<true, float>, <false, float> -- are summed under the short name)
            k = ks.setdefault(short(r["Name"]), {})
            tot = k.get("avg_launch_us", 0.0) * k.get("launches", 0) + float(r["AverageNs"]) * 1e-3 * int(r["Calls"])
            k["launches"] = k.get("launches", 0) + int(r["Calls"])
            k["avg_launch_us"] = round(tot / k["launches"], 3)
            k["share_of_device_time_pct"] = round(k.get("share_of_device_time_pct", 0.0) + float(r["Percentage"]), 6)
    for f in sorted(glob.glob(os.path.join(d, f"pmc_SQ*{suffix}", "*counter_collection.csv")) + glob.glob(os.path.join(d, "pmc_SQ*_by_kernel.csv"))):
        for kn, c in counters(f).items():
            k = ks.setdefault(kn, {})
            g = lambda n: c[n][1] if n in c else None  # noqa: E731
            gui, mfma = g("GRBM_GUI_ACTIVE"), g("SQ_VALU_MFMA_BUSY_CYCLES")
            if gui and mfma is not None:
                # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (one GRBM each): the cycles of the launch are 1/8 of it
                k["mfma_busy_frac"] = round(mfma / (gui / 8.0 * 256 * 4), 5)
            wave, wait, winst, act = g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY"), g("SQ_ACTIVE_INST_ANY")
            if wave:
                k["wave_cycles_split"] = {"wait_any": round(wait / wave, 4) if wait is not None else None,
                                          "wait_inst_any": round(winst / wave, 4) if winst is not None else None,
                                          "active_inst_any": round(act / wave, 4) if act is not None else None}
            if g("SQ_ACTIVE_INST_VALU") is not None and act:
                k["valu_share_of_active"] = round(g("SQ_ACTIVE_INST_VALU") / act, 4)
            for cn in c:
                if cn.startswith("SQ_INSTS_VALU_MFMA_MOPS_"):
                    k["mfma_flops_per_launch"] = round(512.0 * c[cn][1] / c[cn][0], 1)
                    k["mfma_flops_counter"] = cn
            k["pmc_launches"] = max(v[0] for v in c.values())
    fetch = glob.glob(os.path.join(d, f"pmc_FETCH_SIZE{suffix}", "*counter_collection.csv")) + glob.glob(os.path.join(d, "pmc_FETCH_SIZE_by_kernel.csv"))
    write = glob.glob(os.path.join(d, f"pmc_WRITE_SIZE{suffix}", "*counter_collection.csv")) + glob.glob(os.path.join(d, "pmc_WRITE_SIZE_by_kernel.csv"))
    if fetch and write:
        fc, wc = counters(fetch[0]), counters(write[0])
        for kn in set(fc) & set(wc):
            if "FETCH_SIZE" not in fc[kn] or "WRITE_SIZE" not in wc[kn]:
                continue
            f = fc[kn]["FETCH_SIZE"][1] / fc[kn]["FETCH_SIZE"][0]
            w = wc[kn]["WRITE_SIZE"][1] / wc[kn]["WRITE_SIZE"][0]
            k = ks.setdefault(kn, {})
            k["FETCH_SIZE_KiB_avg"] = round(f, 1)
            k["WRITE_SIZE_KiB_avg"] = round(w, 1)
            k["hbm_bytes_per_launch_corrected"] = int((2 * f + w) * 1024)
    out["kernels"] = {k: v for k, v in ks.items() if not k.startswith("__amd")}
    return out


if __name__ == "__main__":
    dst = sys.argv[1]
    doc = {"units": "durations us; FETCH/WRITE_SIZE KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction of MI355X_MICROARCH.md); "
                    "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 * 4) (MfmaUtil); mfma_flops = 512 * SQ_INSTS_VALU_MFMA_MOPS_F64", "runs": {}}
    if os.path.exists(dst):
        doc = json.load(open(dst))
    for spec in sys.argv[2:]:
        key, d = spec.split("|", 1) if "|" in spec else spec.split("=", 1)
        suffix = ""
        if "@" in d:
            d, suffix = d.split("@", 1)
        doc["runs"][key] = one_run(d, suffix)
    json.dump(doc, open(dst, "w"), indent=1)
    print(json.dumps(doc["runs"], indent=1))
