import csv, glob, sys
f = sorted(glob.glob(sys.argv[1], recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last factorisation: find last k_set_slot
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_set_slot")]
start = idx[-2] if len(idx) > 1 else 0
end = idx[-1] if len(idx) > 1 else len(rows)
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:end]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} {(int(r['End_Timestamp'])-t0)/1e3:8.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}  q={r.get('Queue_Id','?'):>3} {r['Kernel_Name'].split('(')[0]:14s} grid={r['Grid_Size_X']}")
