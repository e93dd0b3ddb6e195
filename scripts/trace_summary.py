"""Summarise a rocprofv3 kernel trace: per kernel, the durations of the last N launches (us)."""
import csv
import glob
import sys

pat = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = sorted(glob.glob(pat, recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in by.items():
    print(f"{k:28s} n={len(v):5d} avg={sum(v)/len(v):9.2f}us  last: " + " ".join(f"{x:.1f}" for x in v[-last:]))
# gaps between consecutive kernels of the last factorisation
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-13:]
t0 = int(tail[0]["Start_Timestamp"])
print("last 13 dispatches (start, dur) us:")
for r in tail:
    print(f"   {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  {r['Kernel_Name'].split('(')[0]}  grid={r['Grid_Size_X']} wg={r['Workgroup_Size_X']} vgpr={r['VGPR_Count']} scratch={r['Scratch_Size']}")
