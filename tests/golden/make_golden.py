#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the reference itself.

Runs ONLY in the build container, where the read-only reference checkout is
mounted at /root/reference.  Nothing of the reference (source, bytecode) is
written into the repository: the outputs are .npz files that hold inputs and
expected outputs of the reference functions on the SOS hot path.

How the reference is driven:
  * `SOS_Aer_I1_In`, `SOS_Aer_In_limit`, `SOS_Aer_global_va` and
    `SOS_Aer_tau_profile` are imported directly from /root/reference.
  * `SOS_Aer_main_specular.py` is not importable as shipped (it imports two
    module names that do not exist, runs at import and has literal
    parameters).  Its source text is read at run time, the import lines of
    modules that are absent or unused are dropped, the parameter literals are
    replaced by the fixture's values, `phase_func(...)` is bound to a function
    that hands back pre-built (P0, P) arrays, the final bare `return` is made
    to return the locals, and the result is exec'd.  All arithmetic executed
    is the reference's own.
  * `SOS_Aer_main_lambertian.py` crashes at its lines 274-276 as shipped
    (SURVEY hazard H1).  For the Lambertian n>=2 fixture those three lines are
    replaced, at run time, by line 274 of the specular main (the specular
    first-order term); every fixture made this way is labelled
    `modified_reference=True`.
  * the phase-function builders `isotropic`, `rayleigh`, `henyey_greenstein`
    are taken out of `SOS_Aer_phase_func.py` by AST (the module itself
    imports `miepython`, which is not installed, and is never imported).

Usage:  python tests/golden/make_golden.py [--only g1,g2,...]
"""
import argparse
import ast
import io
import os
import re
import sys
import time
import types
import warnings
import contextlib

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

if not os.path.isdir(REF):
    raise SystemExit("the reference checkout is not present; fixtures can only be regenerated in the build container")

sys.path.insert(0, REF)
warnings.simplefilter("ignore")

import SOS_Aer_I1_In as R_I1In          # noqa: E402
import SOS_Aer_In_limit as R_lim        # noqa: E402
import SOS_Aer_global_va as R_gva       # noqa: E402


def _quiet():
    return contextlib.redirect_stdout(io.StringIO())


# --------------------------------------------------------------------------
# phase-function builders of the reference, extracted by AST
# --------------------------------------------------------------------------
def _load_phase_builders():
    src = open(os.path.join(REF, "SOS_Aer_phase_func.py")).read()
    tree = ast.parse(src)
    wanted = {"isotropic", "rayleigh", "henyey_greenstein", "fwc", "interpolate_fwc_phase"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    mod = ast.Module(body=body, type_ignores=[])
    import SOS_Aer_fwc_data as R_fwc      # two literal arrays (fwc:3,173), pure data
    ns = {"np": np, "tqdm": lambda it, **kw: it, "mu_fwc": R_fwc.mu_fwc, "phase_func_FWC": R_fwc.phase_func_FWC}
    exec(compile(mod, "<reference phase builders>", "exec"), ns)
    return ns


_PB = _load_phase_builders()


def ref_phase(name, N, mu, mu0, g=0.0):
    if name == "iso":
        return _PB["isotropic"](N, mu)
    if name == "rayleigh":
        return _PB["rayleigh"](N, mu, mu0)
    if name == "hg":
        return _PB["henyey_greenstein"](N, mu, mu0, g)
    if name == "fwc":
        return _PB["fwc"](N, mu, mu0)
    raise ValueError(name)


def _ref_P0_only(name, N, mu, mu0, g):
    """P0 of a reference builder without its O(D^2 25) P(mu, mu') loop: the builder's statements up to and
    including `P0 = P0/np.trapz(P0, mu) *2` are executed, the rest is dropped (run-time AST cut)."""
    fn = {"rayleigh": "rayleigh", "hg": "henyey_greenstein", "fwc": "fwc"}[name]
    src = open(os.path.join(REF, "SOS_Aer_phase_func.py")).read()
    tree = ast.parse(src)
    f = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == fn][0]
    cut = None
    for i, st in enumerate(f.body):
        if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name) and st.targets[0].id == "P0" and i > 3:
            cut = i
    assert cut is not None
    f.body = f.body[:cut + 1] + [ast.Return(value=ast.Name(id="P0", ctx=ast.Load()))]
    mod = ast.fix_missing_locations(ast.Module(body=[f], type_ignores=[]))
    ns = dict(_PB)
    exec(compile(mod, "<reference %s, P0 part>" % fn, "exec"), ns)
    return ns[fn](N, mu, mu0, g) if name == "hg" else ns[fn](N, mu, mu0)


def make_mu(N):
    return np.concatenate((np.linspace(-1, 0, N), np.linspace(0, 1, N)))


# --------------------------------------------------------------------------
# G1: step-level, single slab  (SOS_Aer_I1_In.py:13,62,77)
# --------------------------------------------------------------------------
def g1():
    cases = [
        # name, L, N, tauStar, phase, g, alb, n_orders_to_store
        ("L50_N32_t0224_iso", 50, 32, 0.224, "iso", 0.0, 1.0, 3),
        ("L50_N32_t0224_hg", 50, 32, 0.224, "hg", 0.7, 1.0, 3),
        ("L50_N32_t2_iso", 50, 32, 2.0, "iso", 0.0, 0.9, 3),       # idx=1 -> n_points<2 line
        ("L40_N128_t005_iso", 40, 128, 0.05, "iso", 0.0, 1.0, 3),  # idx=0, a4a windowed value survives
        ("L40_N128_t05_iso", 40, 128, 0.5, "iso", 0.0, 1.0, 3),    # idx=2 -> 2-point line
        ("L40_N100_t2_iso", 40, 100, 2.0, "iso", 0.0, 0.9, 3),     # idx=4 -> 4-point LSQ
        ("L40_N64_t45_hg", 40, 64, 4.5, "hg", 0.7, 0.9, 3),        # idx=3 -> exact quadratic
        ("L40_N128_t2_iso", 40, 128, 2.0, "iso", 0.0, 0.9, 3),     # idx=5 -> 5-point LSQ
        ("L30_N128_t45_iso", 30, 128, 4.5, "iso", 0.0, 0.9, 3),    # idx=7 -> 5-point LSQ
        ("L24_N256_t003_iso", 24, 256, 0.03, "iso", 0.0, 1.0, 2),  # idx=1, two a4a lanes, one is a source
    ]
    mu0 = 0.5
    for name, L, N, tauStar, ph, g, alb, nst in cases:
        mu = make_mu(N)
        # non-uniform tau grid (the reference accepts any increasing grid)
        x = np.linspace(0, 1, L)
        tau = tauStar * (0.35 * x + 0.65 * x * x)
        tau[-1] = tauStar
        P0, P = ref_phase(ph, N, mu, mu0, g)
        out = dict(tau=tau, mu=mu, tauStar=tauStar, mu0=mu0, alb=alb, N=N, P0=P0)
        if ph == "iso":
            out["P_const"] = np.float64(P[0, 0])
        else:
            out["P"] = P
        I1 = R_I1In.I1_NumInt(tau, mu, tauStar, mu0, P0, alb, N)
        out["I1"] = I1
        In_1 = I1
        err = ""
        for n in range(2, 2 + nst):
            Jn = R_I1In.Jn_NumInt(n, In_1, tau, mu, tauStar, mu0, P, alb, N)
            try:
                In = R_I1In.In_NumInt(n, Jn, In_1, tau, mu, tauStar, mu0, P, alb, N, 0, 0)
            except IndexError as e:  # unbounded while of the upward blend
                err = "IndexError at n=%d" % n
                out["Jn_%d" % n] = Jn
                break
            out["Jn_%d" % n] = Jn
            out["In_%d" % n] = In
            In_1 = In
        out["error"] = err
        out["numpy_version"] = np.__version__
        np.savez_compressed(os.path.join(OUT, "g1_%s.npz" % name), **out)
        print("g1", name, "stored orders", nst, err)


# --------------------------------------------------------------------------
# G2: helper level  (SOS_Aer_In_limit.py:70,113,155,145)
# --------------------------------------------------------------------------
def g2():
    rng = np.random.default_rng(20250905)
    asym = []
    for trial in range(40):
        n = int(rng.integers(1, 30))
        tau_slice = np.sort(rng.uniform(0, 0.5, n))
        J = rng.uniform(0.01, 1.0, n)
        tau_t = tau_slice[-1]
        for mu in (-5e-9, -5e-4, -2e-3, -7.874015748031496e-3, -9.9e-3):
            r = R_lim.improved_asymptotic_downward_radiance(J, tau_slice, tau_t, mu)
            asym.append((J, tau_slice, tau_t, mu, r))
    # window that excludes everything (tau_t far beyond the slice) -> -J[-1]
    J = np.array([0.3, 0.2]); ts = np.array([0.0, 0.01])
    asym.append((J, ts, 1.0, -2e-3, R_lim.improved_asymptotic_downward_radiance(J, ts, 1.0, -2e-3)))
    # nan in the integrand -> -J[-1]
    J = np.array([0.3, np.nan, 0.2]); ts = np.array([0.0, 0.005, 0.01])
    asym.append((J, ts, 0.01, -2e-3, R_lim.improved_asymptotic_downward_radiance(J, ts, 0.01, -2e-3)))
    out = {"n_asym": len(asym)}
    for i, (J, ts, tt, mu, r) in enumerate(asym):
        out["a%d_J" % i] = J; out["a%d_tau" % i] = ts
        out["a%d_par" % i] = np.array([tt, mu, r])
    # empty slice
    out["asym_empty"] = np.float64(R_lim.improved_asymptotic_downward_radiance(np.zeros(0), np.zeros(0), 0.1, -2e-3))

    lim = []
    for N in (32, 64, 100, 128, 256, 501):
        mu_down = np.linspace(-1, 0, N)
        for idx in sorted({int(c * N) for c in (0.005, 0.02, 0.04, 0.06)}):
            if idx == 0:
                continue
            row = np.exp(-0.3 / np.maximum(-mu_down, 1e-3)) * (1 + 0.1 * rng.standard_normal(N))
            vals = np.array([R_lim.improved_limit_mu_down(row, mu_down, N, idx, i) for i in range(idx)])
            vlin = np.array([R_lim.limit_mu_down(row, mu_down, N, idx, i) for i in range(idx)])
            lim.append((N, idx, row, vals, vlin))
    out["n_lim"] = len(lim)
    for i, (N, idx, row, vals, vlin) in enumerate(lim):
        out["l%d_Nidx" % i] = np.array([N, idx]); out["l%d_row" % i] = row
        out["l%d_vals" % i] = vals; out["l%d_lin" % i] = vlin
    mua = []
    for N in (32, 128, 256, 501):
        mua.append((N,) + tuple(R_lim.mu_approx_In(make_mu(N), N)))
    out["mu_approx"] = np.array(mua)
    out["thresholds"] = np.array([R_gva.MU_THRESHOLD, R_gva.MU_EXTREME_THRESHOLD, R_gva.MU_VERY_SMALL_THRESHOLD])
    out["numpy_version"] = np.__version__
    np.savez_compressed(os.path.join(OUT, "g2_helpers.npz"), **out)
    print("g2 helpers:", len(asym), "asymptotic,", len(lim), "limit cases")


# --------------------------------------------------------------------------
# full-column harness for the specular / lambertian mains
# --------------------------------------------------------------------------
_PARAM_NAMES = ["mu0", "z0", "z_up", "z_down", "nb_layers", "tauStar_atm", "tauStar_aer",
                "grd_alb", "alb_atm", "alb_aer", "nb_angles"]


def run_main(which, params, phases):
    """exec the reference main `which` ('specular' | 'lambertian') with
    `params` (dict of reference local names -> values) and `phases`
    {'atm': (P0,P), 'aer': (P0,P)}.  Returns the dict of its locals."""
    fn = os.path.join(REF, "SOS_Aer_main_%s.py" % which)
    lines = open(fn).read().split("\n")
    if which == "lambertian":
        spec_lines = open(os.path.join(REF, "SOS_Aer_main_specular.py")).read().split("\n")
        # H1 bypass: lam:274-276 (1-based) <- spec:274
        assert "scatt_surface = np.zeros(nb_angles)" in lines[273]
        lines[273:276] = [spec_lines[273]]
    src = "\n".join(lines)
    # imports: drop the ones that name absent/unused modules
    src = src.replace("from I1_In import", "from SOS_Aer_I1_In import")
    src = re.sub(r"^from SOS_Aer_vdh_extract import .*$", "", src, flags=re.M)
    src = re.sub(r"^from SOS_Aer_phase_func import .*$", "", src, flags=re.M)
    src = re.sub(r"^from SOS_Aer_graphe import .*$", "", src, flags=re.M)
    src = re.sub(r"^from tqdm import .*$", "", src, flags=re.M)
    # do not run at import
    src = re.sub(r"^SOS_Aer\(\)\s*$", "", src, flags=re.M)
    # parameter literals
    for k in _PARAM_NAMES:
        if k in params:
            src, nsub = re.subn(r"^(    %s = )[^#\n]*" % re.escape(k), lambda m: m.group(1) + repr(params[k]) + " ", src, count=1, flags=re.M)
            assert nsub == 1, k
    # plotting calls (lambertian main calls one) -> no-op
    src = re.sub(r"^(\s*)graphe_\w+\(.*$", r"\1pass", src, flags=re.M)
    # return the locals
    src, nsub = re.subn(r"^    return\s*$", "    return dict(locals())", src, flags=re.M)
    assert nsub == 1

    def phase_func(mol, *a, **k):
        return phases[mol]

    import matplotlib.pyplot as plt
    plt.show = lambda *a, **k: None
    ns = {"phase_func": phase_func, "__name__": "ref_main_" + which}
    with _quiet():
        exec(compile(src, "<reference main %s, parameters injected>" % which, "exec"), ns)
        t0 = time.perf_counter()
        loc = ns["SOS_Aer"]()
        dt = time.perf_counter() - t0
    plt.close("all")
    loc["_seconds"] = dt
    return loc


def _column_case(which, params, atm, aer, store="full"):
    N = params["nb_angles"]
    mu = make_mu(N)
    mu0 = params["mu0"]
    ph = {"atm": ref_phase(atm[0], N, mu, mu0, atm[1]), "aer": ref_phase(aer[0], N, mu, mu0, aer[1])}
    loc = run_main(which, params, ph)
    out = {k: np.asarray(v) for k, v in params.items()}
    out.update(surface=which, atm_phase=atm[0], atm_g=atm[1], aer_phase=aer[0], aer_g=aer[1],
               tau=loc["tau"], mu=loc["mu"], idx_up=loc["idx_up"], idx_down=loc["idx_down"], n=loc["n"],
               dtau_aer=loc["dtau_aer"], dtau_atm=loc["dtau_atm"],
               ref_seconds=loc["_seconds"], numpy_version=np.__version__,
               modified_reference=(which == "lambertian"))
    I = loc["I"]; Isv = np.stack(loc["I_saved"])
    L = I.shape[0]
    if store in ("full", "full+P"):
        out["I"] = I; out["I_saved"] = Isv
        out["P0_atm"] = ph["atm"][0]; out["P0_aer"] = ph["aer"][0]
        for mol, spec in (("atm", atm), ("aer", aer)):
            if spec[0] == "iso":
                out["P_%s_const" % mol] = np.float64(ph[mol][1][0, 0])
            else:
                out["P_%s" % mol] = ph[mol][1]
    if store in ("digest", "digest+P"):
        out["toa_up"] = I[0, N:]; out["sfc_down"] = I[L - 1, :N]
        out["col_sum"] = I.sum(axis=0); out["row_sum"] = I.sum(axis=1)
        out["order_toa_up_max"] = Isv[:, 0, N:].max(axis=1)
        out["order_sum"] = Isv.reshape(Isv.shape[0], -1).sum(axis=1)
        out["I_mid_row"] = I[L // 2]
        out["P0_atm"] = ph["atm"][0]; out["P0_aer"] = ph["aer"][0]
        if store == "digest+P":
            out["P_atm"] = ph["atm"][1]; out["P_aer"] = ph["aer"][1]
    return out, loc


def g3():
    base = dict(z0=120, z_up=25, z_down=17, tauStar_atm=0.104, tauStar_aer=0.120,
                alb_atm=1.0, alb_aer=1.0)
    # C1: iso / iso, L=50, N=32
    p = dict(base, mu0=0.5, nb_layers=50, nb_angles=32, grd_alb=0.15)
    out, loc = _column_case("specular", p, ("iso", 0.0), ("iso", 0.0))
    np.savez_compressed(os.path.join(OUT, "g3_spec_C1_iso.npz"), **out)
    print("g3 spec C1: n=%d idx=%d,%d TOA-up max %.16g  (%.2fs)" % (loc["n"], loc["idx_up"], loc["idx_down"], loc["I"][0, 32:].max(), loc["_seconds"]))
    # Rayleigh + HG with mu0 within 1e-4 of a grid node, absorbing aerosol, thicker slab
    N = 64
    mu = make_mu(N)
    mu0 = float(mu[N + 40] + 5e-5)
    p = dict(base, mu0=mu0, nb_layers=60, nb_angles=N, grd_alb=0.3, tauStar_aer=0.6, alb_aer=0.9)
    out, loc = _column_case("specular", p, ("rayleigh", 0.0), ("hg", 0.7))
    np.savez_compressed(os.path.join(OUT, "g3_spec_L60_N64_ray_hg_mu0node.npz"), **out)
    print("g3 spec L60 N64: n=%d idx=%d,%d TOA-up max %.16g  (%.2fs)" % (loc["n"], loc["idx_up"], loc["idx_down"], loc["I"][0, N:].max(), loc["_seconds"]))
    # black surface, thick slab spread over many layers (more orders; the a4b idx
    # bucket of the upper zone differs from the one of the slab / lower zone)
    p = dict(base, mu0=0.8, nb_layers=48, nb_angles=100, grd_alb=0.0, tauStar_aer=1.2, alb_aer=0.9,
             z_up=60, z_down=20)
    out, loc = _column_case("specular", p, ("iso", 0.0), ("iso", 0.0))
    np.savez_compressed(os.path.join(OUT, "g3_spec_L48_N100_thick_black.npz"), **out)
    print("g3 spec L48 N100 thick: n=%d idx=%d,%d TOA-up max %.16g  (%.2fs)" % (loc["n"], loc["idx_up"], loc["idx_down"], loc["I"][0, 100:].max(), loc["_seconds"]))


def g6():
    base = dict(z0=120, z_up=25, z_down=17, tauStar_atm=0.104, tauStar_aer=0.120,
                alb_atm=1.0, alb_aer=1.0)
    p = dict(base, mu0=0.5, nb_layers=50, nb_angles=32, grd_alb=0.15)
    out, loc = _column_case("lambertian", p, ("iso", 0.0), ("iso", 0.0))
    np.savez_compressed(os.path.join(OUT, "g6_lam_C1_iso_modified.npz"), **out)
    print("g6 lam (H1 bypassed) C1: n=%d, reflected upward I2 at surface %.6g" % (loc["n"], loc["I_saved"][1][-1, 40]))
    p = dict(base, mu0=0.6, nb_layers=40, nb_angles=64, grd_alb=0.5, tauStar_aer=0.4)
    out, loc = _column_case("lambertian", p, ("rayleigh", 0.0), ("hg", 0.6))
    np.savez_compressed(os.path.join(OUT, "g6_lam_L40_N64_ray_hg_modified.npz"), **out)
    print("g6 lam (H1 bypassed) L40 N64: n=%d" % loc["n"])


def g4():
    base = dict(z0=120, z_up=25, z_down=17, tauStar_atm=0.124, tauStar_aer=0.120,
                alb_atm=1.0, alb_aer=0.97)
    # C2-like: Rayleigh + HG(0.7) stand-in, L=200, N=128
    p = dict(base, mu0=0.5, nb_layers=200, nb_angles=128, grd_alb=0.15)
    out, loc = _column_case("specular", p, ("rayleigh", 0.0), ("hg", 0.7), store="digest+P")
    np.savez_compressed(os.path.join(OUT, "g4_spec_C2_L200_N128_digest.npz"), **out)
    print("g4 C2 digest: n=%d idx=%d,%d TOA-up max %.16g  reference %.2f s/column" % (
        loc["n"], loc["idx_up"], loc["idx_down"], loc["I"][0, 128:].max(), loc["_seconds"]))


def g5():
    for N in (32,):
        mu = make_mu(N)
        for mu0 in (0.5, 0.8125):
            out = {"mu": mu, "mu0": mu0, "N": N}
            out["ray_P0"], out["ray_P"] = ref_phase("rayleigh", N, mu, mu0)
            out["hg07_P0"], out["hg07_P"] = ref_phase("hg", N, mu, mu0, 0.7)
            out["hg03_P0"], out["hg03_P"] = ref_phase("hg", N, mu, mu0, 0.3)
            out["iso_P0"], out["iso_P"] = ref_phase("iso", N, mu, mu0)
            out["fwc_P0"], out["fwc_P"] = ref_phase("fwc", N, mu, mu0)
            np.savez_compressed(os.path.join(OUT, "g5_phase_N%d_mu0_%g.npz" % (N, mu0)), **out)
    # the BASELINE angular resolution: P0 for several mu0 of a sweep; P as digests (the matrices are 0.5 MB each)
    N = 128
    mu = make_mu(N)
    out = {"mu": mu, "N": N, "mu0": np.array([0.2, 0.5, 0.8125, 1.0])}
    for tag, name, g in (("ray", "rayleigh", 0.0), ("hg07", "hg", 0.7), ("fwc", "fwc", 0.0)):
        P0s = []
        for mu0 in out["mu0"]:
            if name == "fwc" and mu0 != 0.5:
                # the builder's P(mu, mu') loop dominates its cost and does not depend on mu0: P0 only
                P0s.append(_ref_P0_only(name, N, mu, float(mu0), g))
                continue
            P0, P = ref_phase(name, N, mu, float(mu0), g)
            P0s.append(P0)
            if mu0 == 0.5:
                out[tag + "_P_rows"] = P[[0, 1, N - 2, N - 1, N, N + 1, 2 * N - 2, 2 * N - 1]]
                out[tag + "_P_colsum"] = P.sum(axis=0)
                out[tag + "_P_diag"] = np.diag(P).copy()
                out[tag + "_P_anti"] = np.diag(P[:, ::-1]).copy()
        if name != "fwc":
            pass
        out[tag + "_P0"] = np.stack(P0s)
    np.savez_compressed(os.path.join(OUT, "g5_phase_N128.npz"), **out)
    # the tabulated fair-weather-cumulus phase function (fwc:3,173): data
    import SOS_Aer_fwc_data as R_fwc
    np.savez_compressed(os.path.join(OUT, "g5_fwc_table.npz"), mu_fwc=R_fwc.mu_fwc, phase_func_FWC=R_fwc.phase_func_FWC)
    # tau_profile
    import SOS_Aer_tau_profile as R_tp
    import matplotlib.pyplot as plt
    plt.show = lambda *a, **k: None
    out = {}
    for i, (ta, tr, z0, zu, zd, L) in enumerate([(0.104, 0.12, 120, 25, 17, 50), (0.124, 0.12, 120, 25, 17, 200),
                                                 (0.124, 0.0075, 120, 15, 14, 400), (0.2, 1.0, 120, 25, 17, 60)]):
        with _quiet():
            tau = R_tp.tau_profile(ta, tr, z0, zu, zd, L)
        plt.close("all")
        out["p%d" % i] = np.array([ta, tr, z0, zu, zd, L], dtype=float)
        out["tau%d" % i] = tau
    out["n"] = 4
    np.savez_compressed(os.path.join(OUT, "g5_tau_profile.npz"), **out)
    print("g5 phase + tau_profile fixtures written")


# --------------------------------------------------------------------------
# G7: what the callers consume -- fluxes, diffusivity, heating rate (SOS_Aer_graphe.py:10,41,74-91,157-158),
# TOA net flux, radiative forcing and critical albedo (SOS_Aer_critical_albedo.py:20-410)
# --------------------------------------------------------------------------
def _graphe_functions():
    """The arithmetic of the reference's plot functions: each function's statements up to (not including) its
    first `plt.` call are executed and its locals returned (run-time AST cut; nothing is stored)."""
    src = open(os.path.join(REF, "SOS_Aer_graphe.py")).read()
    tree = ast.parse(src)
    out = []
    for f in tree.body:
        if not isinstance(f, ast.FunctionDef):
            continue
        cut = len(f.body)
        for i, st in enumerate(f.body):
            if "plt." in ast.unparse(st) and not isinstance(st, ast.For):
                cut = i
                break
        ret = ast.parse("return dict(locals())").body[0]
        f.body = f.body[:cut] + [ret]
        out.append(f)
    mod = ast.fix_missing_locations(ast.Module(body=out, type_ignores=[]))
    import matplotlib.pyplot as plt       # graphe_successive_dif plots inside its loop over the orders
    ns = {"np": np, "plt": plt}
    exec(compile(mod, "<reference graphe arithmetic>", "exec"), ns)
    return ns


def _crit_functions(tauStar_tot):
    """SOS_Aer_radiative_forcing / SOS_Aer_critical_albedo: lines 1-410 of the reference file (its script body
    is not run), with the module-global `tauStar_tot` the function relies on (crit:39 vs crit:486) injected.
    A second copy of the forcing function returns its locals at the point of `return net_flux_toa`."""
    lines = open(os.path.join(REF, "SOS_Aer_critical_albedo.py")).read().split("\n")
    end = [i for i, l in enumerate(lines) if l.startswith("#                   MAIN SCRIPT")][0] - 1
    src = "\n".join(lines[:end])
    src = src.replace("from I1_In import", "from SOS_Aer_I1_In import")
    src = re.sub(r"^from SOS_Aer_phase_func import .*$", "", src, flags=re.M)
    src = re.sub(r"^from tqdm import .*$", "", src, flags=re.M)
    ns = {"__name__": "ref_crit", "tauStar_tot": tauStar_tot}
    exec(compile(src, "<reference critical_albedo, functions only>", "exec"), ns)
    src2, nsub = re.subn(r"^        return net_flux_toa\s*$", "        return dict(locals())", src, flags=re.M)
    assert nsub == 1
    ns2 = {"__name__": "ref_crit_locals", "tauStar_tot": tauStar_tot}
    exec(compile(src2, "<reference critical_albedo, locals>", "exec"), ns2)
    return ns, ns2


def g7():
    G = _graphe_functions()
    base = dict(z0=120, z_up=25, z_down=17, tauStar_atm=0.104, tauStar_aer=0.120, alb_atm=1.0, alb_aer=1.0)
    cases = [
        ("C1_iso", dict(base, mu0=0.5, nb_layers=50, nb_angles=32, grd_alb=0.15), ("iso", 0.0), ("iso", 0.0)),
        ("L60_N64_ray_hg", dict(base, mu0=0.6, nb_layers=60, nb_angles=64, grd_alb=0.3, tauStar_aer=0.6, alb_aer=0.9),
         ("rayleigh", 0.0), ("hg", 0.7)),
        ("L40_N128_ray_hg", dict(base, mu0=0.35, nb_layers=40, nb_angles=128, grd_alb=0.05, tauStar_atm=0.124, alb_aer=0.97),
         ("rayleigh", 0.0), ("hg", 0.7)),
    ]
    for name, p, atm, aer in cases:
        out, loc = _column_case("specular", p, atm, aer, store="full+P")
        L, N = p["nb_layers"], p["nb_angles"]
        I, mu, tau, mu0 = loc["I"], loc["mu"], loc["tau"], p["mu0"]
        z = np.linspace(p["z0"], 0, L)
        F0 = np.pi / mu0
        iu, idn = int(loc["idx_up"]), int(loc["idx_down"])
        with _quiet():
            r = G["graphe_diffusivity"](I, mu, z, L, "x")
            out["diffusivity"] = r["dif"]
            r = G["graphe_flux"](I, mu, z, L, N, tau, mu0, F0, p["grd_alb"], "x")
            out["flux_net_F0"] = r["flux"]
            r = G["graphe_flux_up_down"](I, mu, z, L, N, tau, mu0, F0, p["grd_alb"], "x")
            out["flux_up_F0"], out["flux_down_F0"] = r["flux_up"], r["flux_down"]
            r = G["graphe_heating_rate"](I, mu, z, L, N, iu, idn, F0, mu0, tau, p["grd_alb"], "x")
            out["flux_up_4pi"], out["flux_down_4pi"] = r["flux_up"], r["flux_down"]
            out["heating_rate"] = r["heating_rate"]
            r = G["graphe_successive_dif"](loc["I_saved"], mu, z, L, N, "x")
            out["diffusivity_last_order"] = r["dif"]
        # the function-form column of crit (own copy of the column arithmetic) on the same inputs
        tst = p["tauStar_atm"] + p["tauStar_aer"]
        C, C2 = _crit_functions(tst)
        args = (loc["dtau_aer"], p["tauStar_atm"], loc["dtau_atm"], loc["P_aer"], loc["P0_aer"], p["alb_aer"], loc["P_atm"],
                loc["P0_atm"], p["alb_atm"], p["grd_alb"], F0, mu, mu0, N, tau, L, iu, idn)
        with _quiet():
            lc = C2["SOS_Aer_radiative_forcing"](0, *args)          # tauStar_aer == 0 -> the branch that returns the net flux
            out["crit_net_flux_toa"] = np.float64(lc["net_flux_toa"])
            out["crit_flux_up"], out["crit_flux_down"] = lc["flux_up"], lc["flux_down"]
            out["crit_I_minus_spec_I_max"] = np.float64(np.max(np.abs(lc["I"] - I)))
            out["crit_n"] = lc["n"]
            out["crit_delta_F_coded"] = np.float64(C["SOS_Aer_radiative_forcing"](p["tauStar_aer"], *args))
            if name == "C1_iso":
                cargs = (p["tauStar_aer"], loc["dtau_aer"], p["tauStar_atm"], loc["dtau_atm"], loc["P_aer"], loc["P0_aer"],
                         loc["P_atm"], loc["P0_atm"], p["alb_atm"], p["grd_alb"], F0, mu, mu0, N, tau, L, iu, idn)
                out["crit_critical_albedo_coded"] = np.float64(C["SOS_Aer_critical_albedo"](*cargs))
                # the net flux at the albedos a bisection with a working baseline would visit
                for w in (0.5, 0.75, 0.875):
                    a2 = list(args); a2[5] = w
                    out["crit_net_flux_toa_alb%g" % w] = np.float64(C2["SOS_Aer_radiative_forcing"](0, *a2)["net_flux_toa"])
        # the aerosol-free column on its own grid (the baseline a forcing needs): spec exec
        p0 = dict(p, tauStar_aer=0.0)
        out0, loc0 = _column_case("specular", p0, atm, aer, store="digest")
        C0, C02 = _crit_functions(p["tauStar_atm"])
        a0 = (loc0["dtau_aer"], p["tauStar_atm"], loc0["dtau_atm"], loc0["P_aer"], loc0["P0_aer"], p["alb_aer"], loc0["P_atm"],
              loc0["P0_atm"], p["alb_atm"], p["grd_alb"], F0, mu, mu0, N, loc0["tau"], L, iu, idn)
        with _quiet():
            out["crit_net_flux_toa_no_aerosol"] = np.float64(C02["SOS_Aer_radiative_forcing"](0, *a0)["net_flux_toa"])
        out["n_no_aerosol"] = loc0["n"]
        out.pop("I_saved")                      # the per-order fields are pinned by g3; here only the total is used
        np.savez_compressed(os.path.join(OUT, "g7_epilogue_%s.npz" % name), **out)
        print("g7", name, "n=%d net TOA flux %.12g  coded dF %g  |I_crit - I_spec| %.2e" % (
            loc["n"], out["crit_net_flux_toa"], out["crit_delta_F_coded"], out["crit_I_minus_spec_I_max"]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="g1,g2,g3,g4,g5,g6,g7")
    a = ap.parse_args()
    cwd = os.getcwd()
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)  # the reference writes cache files into the CWD
        for g in a.only.split(","):
            globals()[g]()
        os.chdir(cwd)
