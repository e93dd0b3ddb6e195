"""Build profiles/rN/pmc_summary.json from two rocprofv3 counter-collection CSVs (FETCH_SIZE pass, WRITE_SIZE pass).

    python scripts/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are in KiB and
FETCH_SIZE reports half of the bytes of wide coalesced reads, so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import json
import sys


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        s = acc.setdefault(k, [0, 0.0])
        s[0] += 1
        s[1] += float(r["Counter_Value"])
    return acc


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, with --kernel-trace only) of "
              "`python bench.py --steps 20 --warmup 2 --no-cpu-baseline`, MI355X",
    "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB; gfx950 correction per MI355X_MICROARCH.md (HBM section): "
             "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 for wide coalesced reads; the kernels here use 8-byte-per-lane "
             "accesses for which the guide calls the absolute value uncalibrated",
    "kernels": {},
}
for k in sorted(set(fetch) & set(write)):
    if k.startswith("__amd"):
        continue
    f = fetch[k][1] / fetch[k][0]
    w = write[k][1] / write[k][0]
    out["kernels"][k] = {
        "launches": fetch[k][0],
        "FETCH_SIZE_KiB_avg": round(f, 1),
        "WRITE_SIZE_KiB_avg": round(w, 1),
        "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024),
        "hbm_bytes_per_launch_uncorrected": int((f + w) * 1024),
    }
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
