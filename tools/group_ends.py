#!/usr/bin/env python3
"""When each column group of a two-group solve ends: python3 tools/group_ends.py <kernel_trace.csv> (last solve of the trace, per HIP queue)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_prepare" in r["Kernel_Name"])
sv = rows[idx:]
t0 = int(sv[0]["Start_Timestamp"])
q = {}
for r in sv:
    n = r["Kernel_Name"]
    if "k_jn_gemm" in n or "k_transport" in n:
        e = q.setdefault(r.get("Queue_Id", "?"), {"n": 0, "end": 0, "busy": 0.0})
        e["n"] += 1; e["end"] = max(e["end"], (int(r["End_Timestamp"]) - t0) / 1e3); e["busy"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, e in q.items():
    print("queue %s: %d contraction / transport launches, last one ends at %.0f us, sum of durations %.0f us" % (k, e["n"], e["end"], e["busy"]))
