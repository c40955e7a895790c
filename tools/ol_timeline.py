#!/usr/bin/env python3
"""Timeline of an order-loop launch from its event log (-DSOSRT_OL_STAMPS build, SOSRT_OL_LOG=<file>): per order of the launch,
when the transport workgroups of the first column learnt that their source function was complete, finished their sweeps and
published the verdict, and when the contraction workgroups saw their tiles' rows ready and had them stored.
usage: tools/ol_timeline.py <log file> [launch index]"""
import sys
from collections import defaultdict

EV = {1: "order top", 2: "Jn complete", 3: "down done", 4: "up done", 5: "verdict", 10: "tile ready", 11: "tile done"}
launches, cur = [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        cur = {"head": line.strip(), "ev": []}
        launches.append(cur)
    elif cur is not None:
        wg, k, ev, t = (int(x) for x in line.split())
        cur["ev"].append((wg, k, ev, t))
L = launches[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
print(L["head"])
t0 = min(e[3] for e in L["ev"])
by = defaultdict(list)
for wg, k, ev, t in L["ev"]:
    by[k].append((((t - t0) & 0xffffffff) / 100.0, wg, ev))
for k in sorted(by)[:12]:
    tr = sorted(x for x in by[k] if x[2] < 10 and x[1] < 2)
    ti = sorted(x for x in by[k] if x[2] >= 10)
    ready = [x[0] for x in ti if x[2] == 10]
    done = [x[0] for x in ti if x[2] == 11]
    print("order %2d: " % k + "  ".join("wg%d %s %.1f" % (wg, EV[ev], t) for t, wg, ev in tr))
    if ready:
        print("          tiles: ready %.1f .. %.1f, done %.1f .. %.1f (%d tiles)" % (min(ready), max(ready), min(done) if done else -1, max(done) if done else -1, len(done)))
