#!/usr/bin/env python3
"""Timeline of the LAST n kernels of a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous kernel's end on
the same queue, stream / queue id, name.  usage: tools/step_timeline.py <kernel_trace.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
last_end = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:7.1f}  q{q:>3}  {r['Kernel_Name'][:90]}")
print(f"span {(max(int(r['End_Timestamp']) for r in rows) - t0) / 1e3:.1f} us")
