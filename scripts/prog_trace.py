"""Development aid: timeline of one program launch (k_program): when each POTRF job was drawn, could start and ended; the spread
of the TRSM / update jobs; idle gaps on the critical chain.  python scripts/prog_trace.py [case] [opt=val ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import cholesky_amd as ca
from conftest import case_paths

case = sys.argv[1] if len(sys.argv) > 1 else "lapl_3375x3375"
m, o, c, _ = case_paths(case)
plan = ca.Plan(m, o, c)
dev = ca.Device(plan, 0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    dev.set_option(k, int(v))
a = dev.new_arena()
for _ in range(3):
    dev.fill(a); dev.factor(a)
dev.sync()
dev.fill(a); dev.sync()
tr = dev.program_trace(a)
us = tr[:, 1:4] * 0.01
kind = tr[:, 0]
print(f"{case}: {len(tr)} jobs, last end {us[:, 2].max():.1f} us")
names = {0: "POTRF", 10: "POTRF*", 1: "TRSM", 2: "UPD"}
for j in range(len(tr)):
    if kind[j] in (0, 10):
        print(f"  job {j:4d} {names[int(kind[j])]:6s} wg {tr[j, 4]:3d}  drawn {us[j, 0]:7.1f}  start {us[j, 1]:7.1f}  end {us[j, 2]:7.1f}  (busy {us[j, 2] - us[j, 1]:6.1f})")
if os.environ.get("TRACE_JOBS"):  # TRACE_JOBS=lo:hi -- every job of the range
    lo, hi = (int(v) for v in os.environ["TRACE_JOBS"].split(":"))
    for j in range(lo, min(hi, len(tr))):
        print(f"  job {j:4d} kind {int(kind[j]):2d} wg {tr[j, 4]:3d}  drawn {us[j, 0]:7.1f}  start {us[j, 1]:7.1f}  end {us[j, 2]:7.1f}")
for k in (1, 2):
    sel = kind == k
    if sel.any():
        d = us[sel]
        print(f"  {names[k]:5s} jobs {sel.sum():4d}: drawn {d[:, 0].min():6.1f}..{d[:, 0].max():6.1f}  start {d[:, 1].min():6.1f}..{d[:, 1].max():6.1f}  end {d[:, 2].min():6.1f}..{d[:, 2].max():6.1f}  mean busy {np.mean(d[:, 2] - d[:, 1]):5.1f} us, mean wait {np.mean(d[:, 1] - d[:, 0]):5.1f} us")
# per 20 us window: jobs running
for t0 in range(0, int(us[:, 2].max()) + 1, 20):
    run = ((us[:, 1] < t0 + 20) & (us[:, 2] > t0)).sum()
    wait = ((us[:, 0] < t0 + 20) & (us[:, 1] > t0)).sum()
    print(f"  t {t0:4d}-{t0 + 20:4d} us: {run:4d} jobs working, {wait:4d} waiting")

# what each POTRF job waited for last: the job whose signal completed its wait list
jobs, waits = plan.program_jobs(follow=not any(kv == "follow=0" for kv in sys.argv[2:]))
sig_jobs = {}
for j in range(len(jobs)):
    for c in (jobs[j, 5], jobs[j, 6]):
        if c >= 0:
            sig_jobs.setdefault(int(c), []).append(j)
heap_level = {int(plan.tree[h - 1]): h.bit_length() - 1 for h in range(1, plan.nsep + 1)}
print("critical waits of the POTRF jobs (last signalling job of every counter waited for):")
for j in range(len(jobs)):
    if jobs[j, 0] != 0:
        continue
    ws = waits[waits[:, 0] == j]
    worst = None
    for _, c, v in ws:
        js = sig_jobs.get(int(c), [])
        if js:
            last = max(js, key=lambda q: us[q, 2])
            if worst is None or us[last, 2] > us[worst, 2]:
                worst = last
    if worst is not None:
        print(f"  POTRF job {j} (sep {jobs[j, 1]} level {heap_level[int(jobs[j, 1])]} col0 {jobs[j, 2]}) start {us[j, 1]:6.1f}: last signal from job {worst} kind {jobs[worst, 0]} "
              f"target col-sep {jobs[worst, 1]} row-sep {jobs[worst, 2]}: drawn {us[worst, 0]:6.1f} start {us[worst, 1]:6.1f} end {us[worst, 2]:6.1f}")
        ws2 = waits[waits[:, 0] == worst]
        w2 = None
        for _, c, v in ws2:
            js = sig_jobs.get(int(c), [])
            if js:
                last = max(js, key=lambda q: us[q, 2])
                if w2 is None or us[last, 2] > us[w2, 2]:
                    w2 = last
        if w2 is not None:
            print(f"        ... which waited for job {w2} kind {jobs[w2, 0]} sep {jobs[w2, 1]} aux {jobs[w2, 2]}: drawn {us[w2, 0]:6.1f} start {us[w2, 1]:6.1f} end {us[w2, 2]:6.1f}")

# channel strips (TRSM jobs a follower follows): per pivot block, their start / end against the POTRF job's
print("followed strips (TRSM jobs with a channel) per source pivot block:")
pot = {(int(jobs[j, 1]), int(jobs[j, 2])): j for j in range(len(jobs)) if jobs[j, 0] == 0}
seen = {}
for j in range(len(jobs)):
    if jobs[j, 0] == 1 and jobs[j, 6] >= 0:
        seen.setdefault((int(jobs[j, 1]), int(jobs[j, 2]), int(jobs[j, 6])), []).append(j)
for (sep, col0, ch), js in sorted(seen.items(), key=lambda kv: us[pot[kv[0][:2]], 2]):
    pj = pot[(sep, col0)]
    if heap_level[sep] >= 4 and col0 == 0:
        continue
    print(f"  sep {sep:2d} (level {heap_level[sep]}) col0 {col0:3d} chan {ch:4d}: POTRF start {us[pj, 1]:6.1f} end {us[pj, 2]:6.1f} | strips drawn {min(us[q, 0] for q in js):6.1f} "
          f"start {min(us[q, 1] for q in js):6.1f}..{max(us[q, 1] for q in js):6.1f} end {min(us[q, 2] for q in js):6.1f}..{max(us[q, 2] for q in js):6.1f}")

# update jobs per target panel (column separator): spread of their starts and ends, by source phase order
print("update jobs per target panel:")
for sep in sorted(set(int(v) for v in jobs[jobs[:, 0] == 2, 1]), key=lambda s_: (heap_level[s_], s_), reverse=True):
    js = [j for j in range(len(jobs)) if jobs[j, 0] == 2 and jobs[j, 1] == sep]
    if heap_level[sep] == max(heap_level.values()):
        continue
    line = f"  panel {sep:2d} (level {heap_level[sep]}): {len(js):3d} jobs; "
    # group by queue position clusters (phases): split where job indices jump by > 50
    groups, cur = [], [js[0]]
    for j in js[1:]:
        if j - cur[-1] > 40:
            groups.append(cur); cur = [j]
        else:
            cur.append(j)
    groups.append(cur)
    for gq in groups:
        d = us[gq]
        line += f"[jobs {gq[0]}-{gq[-1]} n={len(gq)} tasks={int(jobs[gq, 4].sum())} drawn {d[:, 0].min():.0f}-{d[:, 0].max():.0f} start {d[:, 1].min():.0f}-{d[:, 1].max():.0f} end {d[:, 2].min():.0f}-{d[:, 2].max():.0f}] "
    print(line)

# TRACE_WAITS=lo:hi -- per job of the range: every wait (in list order = stage order) with the time its counter was complete
if os.environ.get("TRACE_WAITS"):
    lo, hi = (int(v) for v in os.environ["TRACE_WAITS"].split(":"))
    print("waits per job (counter: value needed, completed at = end of the last job signalling it; channel counters are raised by strips column by column: shown as the strips' end):")
    chan_jobs = {}
    for j in range(len(jobs)):
        if jobs[j, 0] == 1 and jobs[j, 6] >= 0:
            for t in range(32):
                chan_jobs.setdefault(int(jobs[j, 6]) + t, []).append(j)
    for j in range(lo, min(hi, len(jobs))):
        ws = waits[waits[:, 0] == j]
        line = f"  job {j} kind {jobs[j, 0]} sep {jobs[j, 1]} aux {jobs[j, 2]} tasks {jobs[j, 4]}: drawn {us[j, 0]:.1f} start {us[j, 1]:.1f} end {us[j, 2]:.1f} |"
        for _, c, v in ws:
            js = sig_jobs.get(int(c), []) or chan_jobs.get(int(c), [])
            t = max((us[q, 2] for q in js), default=-1.0)
            line += f" c{c}>={v}@{t:.1f}"
        print(line)
