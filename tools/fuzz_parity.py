#!/usr/bin/env python3
"""Randomised parity: columns with random zone positions, optical depths, albedos and surfaces, and phase matrices whose rows
oscillate over the first J upward directions with a random amplitude -- so that the upward mu -> 0+ search (spec:403-406) runs
long in some rows and orders and short in others: rows finished one by one, the full row-by-row redo when the first row of a
zone is hit, the transposed fast path elsewhere -- through the ring kernel, the chunk-parallel kernel (one workgroup and
ceil(N/64) workgroups per column) and the general kernel: ring == chunk-parallel bit for bit, every kernel against the oracle at
1e-10 with equal order counts (or IndexError where the oracle raises it).
run_layers: the same for columns with two or three aerosol layers (five or seven zones) against the oracle's
zone-table path (parity unpinned by construction: the reference has one layer).
    python3 tools/fuzz_parity.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("sos-radiative-transfer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import sos_oracle as O
from sosrt import main as M
from sosrt.main import SOS_Aer_batch, SOS_Aer_layers
from util import rel_err

MODES = (("ring", {"SOSRT_TRANSPORT": "ring"}), ("scan", {"SOSRT_TRANSPORT": "scan"}),
         ("scan1", {"SOSRT_TRANSPORT": "scan", "SOSRT_SCAN_SPLIT": "0"}), ("general", {"SOSRT_TRANSPORT": "general"}))


def run(cases=24, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(cases):
        N = int(rng.choice([64, 100, 128, 128, 192, 256]))
        L = int(rng.integers(12, 100))
        z_up = float(rng.uniform(20, 100)); z_down = float(rng.uniform(5, z_up - 5))
        surface = str(rng.choice(["specular", "specular", "lambertian_readme"]))
        B = 3
        mu0 = rng.uniform(0.15, 1.0, B); taer = rng.choice([0.02, 0.1, 0.4, 1.0], B); rho = rng.uniform(0.0, 0.8, B)
        J = int(rng.choice([0, 20, 50, 66, 70, 90])); amp = float(rng.choice([0.003, 0.02, 0.3]))
        mu = O.make_mu(N)
        c = np.ones(2 * N)
        c[N:] = np.where(np.arange(N) < J, 1 + amp * (-1.0) ** np.arange(N), 1.0)
        P_atm = O.phase_rayleigh(N, mu, 0.5)[1] * c[:, None]
        P_aer = O.phase_hg(N, mu, 0.5, float(rng.choice([0.5, 0.7, 0.85])))[1] * c[:, None]
        P0a = np.stack([O.phase_rayleigh(N, mu, m)[0] for m in mu0]); P0r = np.stack([O.phase_hg(N, mu, m, 0.7)[0] for m in mu0])
        kw = dict(tauStar_atm=0.124, alb_aer=0.95, nb_layers=L, nb_angles=N, z_up=z_up, z_down=z_down, surface=surface,
                  P_atm=P_atm, P_aer=P_aer, P0_atm=P0a, P0_aer=P0r, max_orders=150, raise_on_error=False)
        out = {}
        for tag, env in MODES:
            if tag == "scan1" and N > 128:
                continue
            for k in ("SOSRT_TRANSPORT", "SOSRT_SCAN_SPLIT"):
                os.environ.pop(k, None)
            os.environ.update(env)
            for s_ in list(M._solvers.values()):
                s_.close()
            M._solvers.clear()
            try:
                out[tag] = SOS_Aer_batch(mu0, taer, rho, **kw)
            except ValueError as e:           # a slab that does not fit the column (idx_up < 1 ...): same for every mode
                out[tag] = e
        if isinstance(out["ring"], Exception):
            if verbose:
                print("case %2d L=%3d N=%3d: %s" % (case, L, N, out["ring"]))
            continue
        ring = out["ring"]
        msg = []
        for tag in out:
            r = out[tag]
            if tag in ("scan", "scan1") and not (np.array_equal(r.n, ring.n) and np.array_equal(r.status, ring.status)
                                                 and np.array_equal(r.I[ring.status == 0], ring.I[ring.status == 0])):
                if _ring_runs(L, N, out[tag].I.shape[0], 3):
                    msg.append("%s differs from ring in bits" % tag)
        worst, long_rows, note = 0.0, 0, ""
        for b in range(B):
            noise = None
            col = O.make_column(mu0[b], 120, z_up, z_down, L, 0.124, taer[b], rho[b], 1.0, 0.95, N, P0a[b], P_atm, P0r[b], P_aer, surface=surface)
            try:
                ref = O.solve_column(col, literal=False, max_orders=150)
            except IndexError:
                for tag in out:
                    if out[tag].status[b] != 1:
                        msg.append("%s column %d: oracle raises IndexError, status %d" % (tag, b, out[tag].status[b]))
                continue
            for In in ref.I_saved[1:]:
                reach = np.argmax(np.abs(np.diff(In[:, N:], 2, axis=1)) > 1e-12, axis=1) + 1
                long_rows += int((reach > 61).sum())
            for tag in out:
                r = out[tag]
                if r.status[b] == 2 and ref.n >= 150:
                    continue
                if r.status[b] != 0 or r.n[b] != ref.n:
                    msg.append("%s column %d: status %d n %d (oracle %d)" % (tag, b, r.status[b], r.n[b], ref.n))
                    continue
                e = rel_err(r.I[b], ref.I)
                worst = max(worst, e)
                if not e <= 1e-10:
                    # a direction just outside the reference's |mu0 - |mu|| < 1e-4 window: the reference's own first order carries
                    # rounding noise of that size there (oracle: first_order_extended); the bar is then three times that noise
                    if noise is None:
                        noise = rel_err(O.first_order(col), O.first_order_extended(col))
                    if e <= 3 * noise:
                        note = "  (bar %.1e: a direction %.1e from mu0, noise of the reference's first order %.1e)" % (
                            3 * noise, np.min(np.abs(np.abs(mu) - mu0[b])), noise)
                    else:
                        msg.append("%s column %d: rel err %.2e" % (tag, b, e))
        bad += bool(msg)
        if verbose or msg:
            print("case %2d L=%3d N=%3d %-17s J=%2d amp=%.3f  orders %s  rows with a search beyond lane 61: %4d  max rel err %.1e  %s" % (
                case, L, N, surface, J, amp, ring.n.tolist(), long_rows, worst, "; ".join(msg) if msg else "ok" + note))
    for k in ("SOSRT_TRANSPORT", "SOSRT_SCAN_SPLIT"):
        os.environ.pop(k, None)
    for s_ in list(M._solvers.values()):
        s_.close()
    M._solvers.clear()
    return bad


def _ring_runs(L, N, B, zones=3):
    """Whether SOSRT_TRANSPORT=ring runs the ring kernel at this shape (else the general kernel: rounding apart from the
    chunk-parallel one, which takes every N > 64 since round 4)."""
    from sosrt.solver import Solver
    from sosrt import _lib, inputs
    keep = os.environ.get("SOSRT_TRANSPORT")
    os.environ["SOSRT_TRANSPORT"] = "ring"
    s = Solver(L, N, device=-1)
    s.set_grid(inputs.direction_grid(N))
    ok = s.plan_launch(B, B, zones=zones)["transport"] == _lib.PLAN_TRANSPORT_RING
    s.close()
    if keep is None:
        os.environ.pop("SOSRT_TRANSPORT", None)
    else:
        os.environ["SOSRT_TRANSPORT"] = keep
    return ok


def _reset(env):
    for k in ("SOSRT_TRANSPORT", "SOSRT_SCAN_SPLIT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for s_ in list(M._solvers.values()):
        s_.close()
    M._solvers.clear()


def run_layers(cases=8, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(cases):
        # (70, 130, 200: direction counts whose rewritten mu -> 0- directions straddle two waves of a half row -- the split
        # chunk-parallel kernel's since round 4, in its zone-table instantiation here)
        N = int(rng.choice([64, 70, 100, 128, 128, 130, 131, 192, 200, 256, 300]))
        L = int(rng.integers(40, 120))
        nsl = int(rng.choice([2, 2, 3]))
        # layer tops and bottoms from the top of the atmosphere down, clear air between them
        edges = np.sort(rng.uniform(6, 112, 2 * nsl))[::-1]
        slabs = [(float(edges[2 * i]), float(edges[2 * i + 1]), float(rng.choice([0.02, 0.1, 0.4, 1.0])), float(rng.uniform(0.8, 1.0))) for i in range(nsl)]
        surface = str(rng.choice(["specular", "specular", "lambertian_readme"]))
        B = 3
        mu0 = rng.uniform(0.15, 1.0, B); rho = rng.uniform(0.0, 0.8, B)
        J = int(rng.choice([0, 20, 50, 66, 70, 90])); amp = float(rng.choice([0.003, 0.02, 0.3]))
        mu = O.make_mu(N)
        c = np.ones(2 * N)
        c[N:] = np.where(np.arange(N) < J, 1 + amp * (-1.0) ** np.arange(N), 1.0)
        P_atm = O.phase_rayleigh(N, mu, 0.5)[1] * c[:, None]
        P_aer = O.phase_hg(N, mu, 0.5, 0.7)[1] * c[:, None]
        try:
            cols = [O.make_column_slabs(mu0[b], 120, slabs, L, 0.124, rho[b], 1.0, N, O.phase_rayleigh(N, mu, mu0[b])[0], P_atm,
                                        O.phase_hg(N, mu, mu0[b], 0.7)[0], P_aer, surface=surface) for b in range(B)]
        except AssertionError as e:             # layers that touch on this grid
            if verbose:
                print("layers case %2d L=%3d N=%3d: %s" % (case, L, N, e))
            continue
        out = {}
        for tag, env in MODES:
            if tag == "scan1" and N > 128:
                continue
            _reset(env)
            out[tag] = SOS_Aer_layers(mu0, rho, slabs, tauStar_atm=0.124, nb_layers=L, nb_angles=N, aer_phase_fun="hg", g_aer=0.7,
                                      P_atm=P_atm, P_aer=P_aer, surface=surface, max_orders=150, raise_on_error=False)
        ring = out["ring"]
        msg = []
        for tag in ("scan", "scan1"):
            r = out.get(tag)
            if r is not None and not (np.array_equal(r.n, ring.n) and np.array_equal(r.status, ring.status)
                                      and np.array_equal(r.I[ring.status == 0], ring.I[ring.status == 0])):
                if _ring_runs(L, N, out[tag].I.shape[0], 2 * nsl + 1):
                    msg.append("%s differs from ring in bits" % tag)
        worst, note = 0.0, ""
        for b in range(B):
            noise = None
            try:
                ref = O.solve_column(cols[b], literal=False, max_orders=150)
            except IndexError:
                for tag in out:
                    if out[tag].status[b] != 1:
                        msg.append("%s column %d: oracle raises IndexError, status %d" % (tag, b, out[tag].status[b]))
                continue
            for tag in out:
                r = out[tag]
                if r.status[b] == 2 and ref.n >= 150:
                    continue
                if r.status[b] != 0 or r.n[b] != ref.n:
                    msg.append("%s column %d: status %d n %d (oracle %d)" % (tag, b, r.status[b], r.n[b], ref.n))
                    continue
                e = rel_err(r.I[b], ref.I)
                worst = max(worst, e)
                if not e <= 1e-10:
                    # (as in run(): a direction just outside the reference's limit window -- three times the reference's own noise)
                    if noise is None:
                        noise = rel_err(O.first_order(cols[b]), O.first_order_extended(cols[b]))
                    if e <= 3 * noise:
                        note = "  (bar %.1e: a direction %.1e from mu0, noise of the reference's first order %.1e)" % (
                            3 * noise, np.min(np.abs(np.abs(mu) - mu0[b])), noise)
                    else:
                        msg.append("%s column %d: rel err %.2e" % (tag, b, e))
        bad += bool(msg)
        if verbose or msg:
            print("layers case %2d L=%3d N=%3d %-17s zones %d J=%2d amp=%.3f  orders %s  max rel err %.1e  %s" % (
                case, L, N, surface, 2 * nsl + 1, J, amp, ring.n.tolist(), worst, "; ".join(msg) if msg else "ok" + note))
    _reset({})
    return bad


def run_wide(cases=10, seed=1, verbose=True):
    """The same for the shapes of the chunk-parallel kernel's WIDE instantiation (round 4: odd N, N > 256, more than 64 chunks per
    sweep -- the reference's shipped N = 501, L = 800 is all three): the default plan (WIDE, ceil(N / 64) workgroups per column)
    and the register-streaming kernel of rounds 1-3 (SOSRT_TRANSPORT=fast) against the oracle at 1e-10 with equal order counts,
    and against each other (different summation forms of the recurrence: rounding apart, no more)."""
    rng = np.random.default_rng(seed + 7)
    bad = 0
    for case in range(cases):
        N = int(rng.choice([129, 191, 257, 300, 333, 501]))
        L = int(rng.choice([rng.integers(12, 90), rng.integers(12, 90), rng.integers(520, 700)])) if N <= 200 else int(rng.integers(12, 110))
        z_up = float(rng.uniform(20, 100)); z_down = float(rng.uniform(5, z_up - 5))
        B = 2
        mu0 = rng.uniform(0.15, 1.0, B); taer = rng.choice([0.02, 0.1, 0.4, 1.0, 3.0], B); rho = rng.uniform(0.0, 0.8, B)
        J = int(rng.choice([0, 20, 66, 90])); amp = float(rng.choice([0.003, 0.02, 0.3]))
        mu = O.make_mu(N)
        c = np.ones(2 * N)
        c[N:] = np.where(np.arange(N) < J, 1 + amp * (-1.0) ** np.arange(N), 1.0)
        P_atm = O.phase_rayleigh(N, mu, 0.5)[1] * c[:, None]
        P_aer = O.phase_hg(N, mu, 0.5, float(rng.choice([0.5, 0.7, 0.85])))[1] * c[:, None]
        P0a = np.stack([O.phase_rayleigh(N, mu, m)[0] for m in mu0]); P0r = np.stack([O.phase_hg(N, mu, m, 0.7)[0] for m in mu0])
        kw = dict(tauStar_atm=0.124, alb_aer=0.95, nb_layers=L, nb_angles=N, z_up=z_up, z_down=z_down, surface="specular",
                  P_atm=P_atm, P_aer=P_aer, P0_atm=P0a, P0_aer=P0r, max_orders=60, raise_on_error=False)
        out, msg = {}, []
        for tag, env in (("wide", {}), ("fast", {"SOSRT_TRANSPORT": "fast"})):
            _reset(env)
            for s_ in list(M._solvers.values()):
                s_.close()
            M._solvers.clear()
            try:
                out[tag] = SOS_Aer_batch(mu0, taer, rho, **kw)
            except ValueError as e:
                out[tag] = e
        if isinstance(out["wide"], Exception):
            if verbose:
                print("wide case %2d L=%3d N=%3d: %s" % (case, L, N, out["wide"]))
            continue
        w, f = out["wide"], out["fast"]
        if not (np.array_equal(w.n, f.n) and np.array_equal(w.status, f.status)):
            msg.append("orders / statuses differ between the kernels: %s %s vs %s %s" % (w.n.tolist(), w.status.tolist(), f.n.tolist(), f.status.tolist()))
        worst, apart = 0.0, 0.0
        for b in range(B):
            col = O.make_column(mu0[b], 120, z_up, z_down, L, 0.124, taer[b], rho[b], 1.0, 0.95, N, P0a[b], P_atm, P0r[b], P_aer)
            try:
                ref = O.solve_column(col, literal=False, max_orders=60)
            except IndexError:
                for tag in out:
                    if out[tag].status[b] != 1:
                        msg.append("%s column %d: oracle raises IndexError, status %d" % (tag, b, out[tag].status[b]))
                continue
            for tag in out:
                r = out[tag]
                if r.status[b] == 2 and ref.n >= 60:
                    continue
                if r.status[b] != 0 or r.n[b] != ref.n:
                    msg.append("%s column %d: status %d n %d (oracle %d)" % (tag, b, r.status[b], r.n[b], ref.n))
                    continue
                e = rel_err(r.I[b], ref.I)
                worst = max(worst, e)
                if not e <= 1e-10:
                    noise = rel_err(O.first_order(col), O.first_order_extended(col))
                    if not e <= 3 * noise:
                        msg.append("%s column %d: rel err %.2e" % (tag, b, e))
            if w.status[b] == 0 and f.status[b] == 0:
                apart = max(apart, rel_err(w.I[b], f.I[b]))
        bad += bool(msg)
        if verbose or msg:
            print("wide case %2d L=%3d N=%3d J=%2d amp=%.3f  orders %s  max rel err %.1e  kernels %.1e apart  %s" % (
                case, L, N, J, amp, w.n.tolist(), worst, apart, "; ".join(msg) if msg else "ok"))
    _reset({})
    for s_ in list(M._solvers.values()):
        s_.close()
    M._solvers.clear()
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    b = run(n, seed) + run_layers(max(n // 2, 1), seed) + run_wide(max(n // 2, 1), seed)
    print("cases with a mismatch:", b)
    sys.exit(1 if b else 0)
