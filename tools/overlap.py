#!/usr/bin/env python3
"""Timeline of a solve with two column groups from a rocprofv3 kernel trace: for each contraction launch, how much of it overlaps a transport launch (diagnostic)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last solve: from the last k_prepare on
idx = max(i for i, r in enumerate(rows) if "k_prepare" in r["Kernel_Name"])
sv = rows[idx:]
t0 = int(sv[0]["Start_Timestamp"])
ev = []
for r in sv:
    n = r["Kernel_Name"]
    kind = "G" if "k_jn_gemm" in n else ("T" if "k_transport" in n else ("F" if "k_first_order" in n else "."))
    ev.append((kind, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?")))
print("solve length %.1f us, %d kernels" % (max(e[2] for e in ev), len(ev)))
G = [e for e in ev if e[0] == "G"]; T = [e for e in ev if e[0] == "T"]
def ov(a, b): return max(0.0, min(a[2], b[2]) - max(a[1], b[1]))
tot_g = sum(e[2] - e[1] for e in G); tot_t = sum(e[2] - e[1] for e in T)
o = sum(ov(a, b) for a in G for b in T)
gg = sum(ov(G[i], G[j]) for i in range(len(G)) for j in range(i + 1, len(G)))
tt = sum(ov(T[i], T[j]) for i in range(len(T)) for j in range(i + 1, len(T)))
print("sum contraction %.0f us, sum transport %.0f us, contraction x transport overlap %.0f us, contraction x contraction %.0f, transport x transport %.0f" % (tot_g, tot_t, o, gg, tt))
for e in ev[:60]:
    print("%s q%s %8.1f -> %8.1f  (%.1f)" % (e[0], e[3], e[1], e[2], e[2] - e[1]))
