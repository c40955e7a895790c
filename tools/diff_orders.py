#!/usr/bin/env python3
"""First element where two transport kernels differ over a whole solve (diagnostic): per-order fields of a small batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sos-radiative-transfer_amd"))
import numpy as np
from sosrt.main import SOS_Aer_batch
rng = np.random.default_rng(11)
B = 24
mu0 = rng.uniform(0.2, 1.0, B); taer = rng.choice([0.02, 0.1, 0.35, 0.9, 2.5], B); rho = rng.uniform(0.0, 0.8, B)
N = int(os.environ.get("NANG", "128"))
SAVE = os.environ.get("SAVE", "1") == "1"
kw = dict(tauStar_atm=0.124, alb_aer=0.95, nb_layers=int(os.environ.get("NLAY", "57")), nb_angles=N, max_orders=200, save_orders=SAVE, raise_on_error=False)
res = {}
for mode in sys.argv[1:] or ["ring", "fast"]:
    os.environ["SOSRT_TRANSPORT"] = mode
    res[mode] = SOS_Aer_batch(mu0, taer, rho, **kw)
a, b = list(res.values())[:2]
print("n equal", np.array_equal(a.n, b.n), "idx", a.idx_up, a.idx_down)
if not SAVE:
    d = np.abs(a.I - b.I)
    print("n ring", a.n[:12], "n other", b.n[:12], "status", b.status[:12])
    print("final field max diff", d.max())
    bb, tt, mm = np.nonzero(d > 0)
    print("columns", np.unique(bb)[:10], "rows", np.unique(tt)[:50], "lanes", np.unique(mm)[:40], len(np.unique(mm)))
    sys.exit(0)
k = min(a.I_saved.shape[1], b.I_saved.shape[1])
print("n ring", a.n[:8], "n other", b.n[:8], "status", b.status[:8])
d = np.abs(a.I_saved[:, :k] - b.I_saved[:, :k])
print("max diff", d.max())
for o in range(d.shape[1]):
    if d[:, o].max() > 0:
        bb, tt, mm = np.nonzero(d[:, o] > 0)
        print("first differing order", o + 1, "columns", np.unique(bb)[:10], "rows", np.unique(tt), "lanes", np.unique(mm))
        for i in range(min(6, len(bb))):
            print("  col %d row %d lane %d: %.17g vs %.17g" % (bb[i], tt[i], mm[i], a.I_saved[bb[i], o, tt[i], mm[i]], b.I_saved[bb[i], o, tt[i], mm[i]]))
        break
