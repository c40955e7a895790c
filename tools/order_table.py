#!/usr/bin/env python3
"""Per-order table of one solve from a rocprofv3 kernel trace: python3 tools/order_table.py <kernel_trace.csv> [solve index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
solves, cur = [], []
for r in rows:
    name = r["Kernel_Name"]
    if "k_prepare" in name and cur and any("k_transport" in x["Kernel_Name"] for x in cur):
        solves.append(cur); cur = []
    cur.append(r)
solves.append(cur)
sv = solves[which]
t0 = int(sv[0]["Start_Timestamp"])
print("solve %d of %d: %d kernels, %.1f us from first start to last end" % (which % len(solves), len(solves), len(sv), (int(sv[-1]["End_Timestamp"]) - t0) / 1e3))
order, prev_end = 1, None
print("%5s %-28s %9s %9s %9s %8s" % ("order", "kernel", "start_us", "dur_us", "gap_us", "grid"))
for r in sv:
    name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("sosrt::", "").replace("(anonymous namespace)::", "")
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "k_jn_gemm" in name: order += 1
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    print("%5d %-28s %9.1f %9.1f %9.1f %8s" % (order, name[:28], (st - t0) / 1e3, (en - st) / 1e3, gap, r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    prev_end = en
