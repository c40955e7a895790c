"""Handle-level Python API over the C ABI (host NumPy arrays in, NumPy arrays out;
`solve_device` works on resident device buffers, e.g. torch tensors)."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from ._lib import check, lib


def _f64(a, shape=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a


def _i32(a, n, name):
    a = np.ascontiguousarray(a, dtype=np.int32)
    if a.shape != (n,):
        raise ValueError("%s has shape %s, expected (%d,)" % (name, a.shape, n))
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _vec(x, B, name):
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float64), (B,)))
    return a


@dataclass
class SolveResult:
    I: np.ndarray                  # [B, L, 2N]
    n: np.ndarray                  # [B] final order count (spec:307-310)
    status: np.ndarray             # [B] SOSRT_COL_*
    I_saved: Optional[np.ndarray]  # [B, max(n), L, 2N] (slots >= n[b] are zero) or None


class Solver:
    """One sosrt handle: fixed (nb_layers, nb_angles), a direction grid, a pair of phase
    matrices and a batch of columns."""

    def __init__(self, nb_layers: int, nb_angles: int, max_batch: int = 1, max_orders: int = 64, device: int = 0):
        self.L, self.N, self.D = int(nb_layers), int(nb_angles), 2 * int(nb_angles)
        self.max_batch, self.max_orders, self.device = int(max_batch), int(max_orders), int(device)
        self._h = ctypes.c_void_p()
        check(lib().sosrt_create(self.device, self.L, self.N, self.max_batch, self.max_orders, ctypes.byref(self._h)))
        self.B = 0
        self.order_budget = self.max_orders
        self.mu = None
        self._P = (None, None)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().sosrt_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup -------------------------------------------------------------
    def set_stream(self, stream_handle: Optional[int]):
        """Run on a caller's HIP stream (its integer handle, e.g. `torch.cuda.current_stream().cuda_stream`).  0 is the
        legacy default stream -- torch's default stream -- exactly as in HIP; None goes back to the handle's own stream (a
        blocking stream: it too orders against the default stream)."""
        if stream_handle is None:
            check(lib().sosrt_use_own_stream(self._h))
        else:
            check(lib().sosrt_set_stream(self._h, ctypes.c_void_p(stream_handle) if stream_handle else None))

    def synchronize(self):
        check(lib().sosrt_synchronize(self._h))

    def set_contraction(self, mode="f64"):
        """'f64' (default, the parity path: fp64 MFMA, using the flip symmetry of the folded matrices when they have it) |
        'f64_full' (fp64 MFMA, always the full 2N x 2N product) | 'f32' (float operands and accumulator in the Jn
        contraction: opt-in, about 3e-7 away from the fp64 result; BASELINE configs[4])."""
        m = {"f64": _lib.CONTRACT_F64, "f32": _lib.CONTRACT_F32, "f64_full": _lib.CONTRACT_F64_FULL}.get(mode)
        if m is None:
            raise ValueError("contraction must be 'f64', 'f64_full' or 'f32'")
        check(lib().sosrt_set_contraction(self._h, m))

    def set_order_budget(self, max_orders: int):
        """The solves that follow run at most `max_orders` orders (<= the handle's max_orders); a column still iterating then
        has status COL_MAXORDERS."""
        check(lib().sosrt_set_order_budget(self._h, int(max_orders)))
        self.order_budget = int(max_orders)

    def set_order_loop(self, on=True):
        """Whether the last orders of the last few live columns run in ONE launch (csrc/order_loop.hip; default on: same
        bits, no launches and no host round trip per order) or every order stays two launches."""
        check(lib().sosrt_set_order_loop(self._h, int(on) if on in (0, 1, 2) else (1 if on else 0)))

    def order_loop_stats(self, column_orders=False):
        """(order-loop launches of the last solve, launches that found their grid not resident and handed back[, the (column,
        order) pairs that ran inside them -- synchronises])"""
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
        check(lib().sosrt_order_loop_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c) if column_orders else None))
        return (a.value, b.value, c.value) if column_orders else (a.value, b.value)

    def phase_asymmetry(self):
        """(max |W[k][m] - W[2N-1-k][2N-1-m]| / max |W| of the folded matrices, whether the next solve uses the symmetry)"""
        a, u = ctypes.c_double(), ctypes.c_int()
        check(lib().sosrt_phase_asymmetry(self._h, ctypes.byref(a), ctypes.byref(u)))
        return a.value, bool(u.value)

    def set_first_order(self, mode="coded"):
        """'coded' (default): spec:104-292, what both mains of the reference compute (specularly reflected beam).
        'readme': the Lambertian first order of the reference's README.md:126-171 (direct beam + the beam reflected
        isotropically by the ground + isotropic reflection of the downward first order) -- PARITY UNPINNED, the
        reference has no runnable code for it; meant for surface='lambertian_readme', after set_columns."""
        m = {"coded": _lib.FIRST_ORDER_CODED, "readme": _lib.FIRST_ORDER_README}.get(mode)
        if m is None:
            raise ValueError("first_order must be 'coded' or 'readme'")
        check(lib().sosrt_set_first_order(self._h, m))

    def set_grid(self, mu):
        mu = _f64(mu, (self.D,), "mu")
        check(lib().sosrt_set_grid(self._h, _ptr(mu)))
        self.mu = mu.copy()
        self._P = (None, None)

    def set_phase(self, P_atm, P_aer=None):
        Pa = _f64(P_atm, (self.D, self.D), "P_atm")
        Pr = None if P_aer is None else _f64(P_aer, (self.D, self.D), "P_aer")
        check(lib().sosrt_set_phase(self._h, _ptr(Pa), _ptr(Pr)))
        self._P = (Pa.copy(), None if Pr is None else Pr.copy())

    def same_grid(self, mu):
        return self.mu is not None and np.array_equal(self.mu, np.asarray(mu, dtype=np.float64))

    def same_phase(self, P_atm, P_aer=None):
        a, r = self._P
        if a is None or not np.array_equal(a, P_atm):
            return False
        if (r is None) != (P_aer is None):
            return False
        return r is None or np.array_equal(r, P_aer)

    def set_columns(self, idx_up, idx_down, mu0, grd_alb, alb_atm, alb_aer, dtau_atm, dtau_aer, tauStar_tot,
                    surface="specular"):
        """Three-zone columns (spec:23-53).  Scalars broadcast over the batch."""
        B = int(np.size(idx_up))
        # 'lambertian' is the reference's file as coded (lam:399/401: negative reflected radiance);
        # 'lambertian_readme' the same term with the sign of the reference's README.md:215
        sf = {"specular": _lib.SURFACE_SPECULAR, "lambertian": _lib.SURFACE_LAMBERTIAN,
              "lambertian_readme": _lib.SURFACE_LAMBERTIAN_README}.get(surface)
        if sf is None:
            raise ValueError("surface must be 'specular', 'lambertian' or 'lambertian_readme', got %r" % (surface,))
        iu = _i32(np.reshape(idx_up, (B,)), B, "idx_up")
        idn = _i32(np.reshape(idx_down, (B,)), B, "idx_down")
        v = [_vec(x, B, n) for x, n in ((mu0, "mu0"), (grd_alb, "grd_alb"), (alb_atm, "alb_atm"), (alb_aer, "alb_aer"),
                                        (dtau_atm, "dtau_atm"), (dtau_aer, "dtau_aer"), (tauStar_tot, "tauStar_tot"))]
        check(lib().sosrt_set_columns(self._h, B, _lib.GEOM_THREE_ZONE, sf, _ptr(iu), _ptr(idn), *[_ptr(x) for x in v]))
        self.B = B

    def set_columns_zones(self, zone_r0, zone_mix, mu0, grd_alb, alb_atm, dtau_atm, zone_alb_aer, zone_dtau_aer, tauStar_tot,
                          nz=None, surface="specular"):
        """Columns described by a zone table: `zone_r0`, `zone_mix` [B, nzmax] (first row of each zone, 1 for an
        aerosol zone), `zone_alb_aer`, `zone_dtau_aer` [B, nzmax] (per aerosol zone), `nz` [B] zones per column (default:
        all nzmax).  (clear, slab, clear) is `set_columns`."""
        zr0 = np.ascontiguousarray(np.atleast_2d(zone_r0), dtype=np.int32)
        B, nzmax = zr0.shape
        zmix = np.ascontiguousarray(np.broadcast_to(np.asarray(zone_mix, dtype=np.int32), (B, nzmax)))
        zwr = np.ascontiguousarray(np.broadcast_to(np.asarray(zone_alb_aer, dtype=np.float64), (B, nzmax)))
        zdt = np.ascontiguousarray(np.broadcast_to(np.asarray(zone_dtau_aer, dtype=np.float64), (B, nzmax)))
        nzv = np.full(B, nzmax, dtype=np.int32) if nz is None else _i32(np.reshape(nz, (B,)), B, "nz")
        sf = {"specular": _lib.SURFACE_SPECULAR, "lambertian": _lib.SURFACE_LAMBERTIAN,
              "lambertian_readme": _lib.SURFACE_LAMBERTIAN_README}.get(surface)
        if sf is None:
            raise ValueError("surface must be 'specular', 'lambertian' or 'lambertian_readme', got %r" % (surface,))
        v = [_vec(x, B, n) for x, n in ((mu0, "mu0"), (grd_alb, "grd_alb"), (alb_atm, "alb_atm"), (dtau_atm, "dtau_atm"),
                                        (tauStar_tot, "tauStar_tot"))]
        check(lib().sosrt_set_columns_zones(self._h, B, sf, nzmax, _ptr(nzv), _ptr(zr0), _ptr(zmix), _ptr(v[0]), _ptr(v[1]),
                                            _ptr(v[2]), _ptr(v[3]), _ptr(zwr), _ptr(zdt), _ptr(v[4])))
        self.B = B

    def set_columns_single_slab(self, mu0, alb, tauStar):
        """Single homogeneous slab over a black surface (I1_In:13-130)."""
        B = int(np.size(mu0))
        m, a, t = _vec(mu0, B, "mu0"), _vec(alb, B, "alb"), _vec(tauStar, B, "tauStar")
        check(lib().sosrt_set_columns(self._h, B, _lib.GEOM_SINGLE_SLAB, _lib.SURFACE_NONE, None, None, _ptr(m), None,
                                      _ptr(a), None, None, None, _ptr(t)))
        self.B = B

    # ---- step level ----------------------------------------------------------
    def first_order(self, tau, P0_atm, P0_aer=None):
        B = self.B
        tau = _f64(tau, (B, self.L), "tau")
        Pa = _f64(P0_atm, (B, self.D), "P0_atm")
        Pr = None if P0_aer is None else _f64(P0_aer, (B, self.D), "P0_aer")
        out = np.empty((B, self.L, self.D))
        check(lib().sosrt_first_order(self._h, B, _ptr(tau), _ptr(Pa), _ptr(Pr), _ptr(out)))
        return out

    def source(self, In_1):
        B = self.B
        x = _f64(In_1, (B, self.L, self.D), "In_1")
        out = np.empty_like(x)
        check(lib().sosrt_source(self._h, B, _ptr(x), _ptr(out)))
        return out

    def transport(self, tau, Jn):
        B = self.B
        tau = _f64(tau, (B, self.L), "tau")
        J = _f64(Jn, (B, self.L, self.D), "Jn")
        out = np.empty_like(J)
        st = np.zeros(B, dtype=np.int32)
        check(lib().sosrt_transport(self._h, B, _ptr(tau), _ptr(J), _ptr(out), _ptr(st)))
        return out, st

    # ---- column level --------------------------------------------------------
    def solve(self, tau, P0_atm=None, P0_aer=None, tol=1e-4, I1=None, save_orders=False, fetch_field=True) -> SolveResult:
        """fetch_field=False leaves the radiance field on the device (SolveResult.I is None) for `epilogue`."""
        B = self.B
        tau = _f64(tau, (B, self.L), "tau")
        Pa = None if P0_atm is None else _f64(P0_atm, (B, self.D), "P0_atm")
        Pr = None if P0_aer is None else _f64(P0_aer, (B, self.D), "P0_aer")
        I1a = None if I1 is None else _f64(I1, (B, self.L, self.D), "I1")
        I = np.empty((B, self.L, self.D)) if fetch_field else None
        n = np.zeros(B, dtype=np.int32)
        st = np.zeros(B, dtype=np.int32)
        sv = None
        if save_orders:
            # The per-order history has n entries (spec:304-305,458), n known only afterwards: a first pass without
            # it (the field stays on the device) gives n, the second stores exactly max(n) orders per column --
            # not max_orders of them (1.6 GB per column at the reference's shipped size with a 256-order budget).
            check(lib().sosrt_solve(self._h, B, _ptr(tau), _ptr(Pa), _ptr(Pr), float(tol), _ptr(I1a), None, None, _ptr(n), _ptr(st)))
            slots = int(max(1, n.max()))
            check(lib().sosrt_set_saved_orders(self._h, slots))
            sv = np.zeros((B, slots, self.L, self.D))
        try:
            check(lib().sosrt_solve(self._h, B, _ptr(tau), _ptr(Pa), _ptr(Pr), float(tol), _ptr(I1a), _ptr(I), _ptr(sv),
                                    _ptr(n), _ptr(st)))
        finally:
            if save_orders:
                check(lib().sosrt_set_saved_orders(self._h, self.max_orders))
        return SolveResult(I=I, n=n, status=st, I_saved=sv)

    def solve_device(self, d_tau: int, d_P0_atm: int, d_P0_aer: int, d_I_out: int, tol=1e-4, d_I1: int = 0,
                     d_I_saved: int = 0, d_n_orders: int = 0, d_status: int = 0):
        """All arguments are device addresses (e.g. torch `tensor.data_ptr()`); work is enqueued on
        the handle's stream (set_stream) and is complete after synchronize()."""
        vp = lambda x: ctypes.c_void_p(x) if x else None
        check(lib().sosrt_solve_dev(self._h, self.B, vp(d_tau), vp(d_P0_atm), vp(d_P0_aer), float(tol), vp(d_I1),
                                    vp(d_I_out), vp(d_I_saved), vp(d_n_orders), vp(d_status)))

    def last_solve_stats(self):
        mo = ctypes.c_int()
        so = ctypes.c_longlong()
        check(lib().sosrt_last_solve_stats(self._h, ctypes.byref(mo), ctypes.byref(so)))
        return mo.value, so.value

    def fluxes(self, tau, I, beam_norm="crit"):
        B = self.B
        tau = _f64(tau, (B, self.L), "tau")
        I = _f64(I, (B, self.L, self.D), "I")
        fd = np.empty((B, self.L))
        fu = np.empty((B, self.L))
        check(lib().sosrt_fluxes(self._h, B, _ptr(tau), _ptr(I), 0 if beam_norm == "crit" else 1, _ptr(fd), _ptr(fu)))
        return fd, fu

    def epilogue(self, z_profile=None, beam_norm="crit", want=("flux_down", "flux_up", "diffusivity", "heating_rate", "net_toa")):
        """Fluxes, diffusivity, heating rate and TOA net flux of the field the last `solve` left on the device
        (graphe:10,74-91,157-158, crit:377-382); only these [B, L] / [B] arrays cross PCIe."""
        B = self.B
        want = set(want)
        if z_profile is None:
            want.discard("heating_rate")
        z = None if z_profile is None else _f64(z_profile, (self.L,), "z_profile")
        arr = {k: (np.empty(B) if k == "net_toa" else np.empty((B, self.L))) if k in want else None
               for k in ("flux_down", "flux_up", "diffusivity", "heating_rate", "net_toa")}
        check(lib().sosrt_epilogue(self._h, B, 0 if beam_norm == "crit" else 1, _ptr(z), _ptr(arr["flux_down"]),
                                   _ptr(arr["flux_up"]), _ptr(arr["diffusivity"]), _ptr(arr["heating_rate"]),
                                   _ptr(arr["net_toa"])))
        return {k: v for k, v in arr.items() if v is not None}

    def epilogue_device(self, d_tau: int, d_I: int, d_z: int = 0, beam_norm="crit", d_flux_down: int = 0, d_flux_up: int = 0,
                        d_diffusivity: int = 0, d_heating_rate: int = 0, d_net_toa: int = 0):
        """Same on caller-owned device buffers (addresses), enqueued on the handle's stream."""
        vp = lambda x: ctypes.c_void_p(x) if x else None
        check(lib().sosrt_epilogue_dev(self._h, self.B, vp(d_tau), vp(d_I), 0 if beam_norm == "crit" else 1, vp(d_z),
                                       vp(d_flux_down), vp(d_flux_up), vp(d_diffusivity), vp(d_heating_rate), vp(d_net_toa)))

    # ---- phase functions on the device (phase:68-292) ----------------------------
    _KINDS = {"iso": _lib.PHASE_ISO, "rayleigh": _lib.PHASE_RAYLEIGH, "hg": _lib.PHASE_HG, "table": _lib.PHASE_TABLE}

    def set_phase_table(self, tab_mu, tab_p):
        tm = np.ascontiguousarray(tab_mu, dtype=np.float64)
        tp = _f64(tab_p, tm.shape, "tab_p")
        check(lib().sosrt_phase_table(self._h, _ptr(tm), _ptr(tp), int(tm.size)))

    def phase_p0(self, kind, mu0, g=0.0):
        """P0(mu, mu0[b]) for an array of mu0 -> [len(mu0), 2N]."""
        m = np.ascontiguousarray(np.atleast_1d(mu0), dtype=np.float64)
        out = np.empty((m.size, self.D))
        step = max(1, self.max_batch)
        for i in range(0, m.size, step):
            mm = np.ascontiguousarray(m[i:i + step])
            oo = np.empty((mm.size, self.D))
            check(lib().sosrt_phase_p0(self._h, int(mm.size), self._KINDS[kind], float(g), _ptr(mm), _ptr(oo)))
            out[i:i + step] = oo
        return out

    def phase_p0_device(self, kind, d_mu0: int, d_P0_out: int, B: int, g=0.0):
        check(lib().sosrt_phase_p0_dev(self._h, int(B), self._KINDS[kind], float(g), ctypes.c_void_p(d_mu0), ctypes.c_void_p(d_P0_out)))

    def phase_matrix(self, kind, g=0.0):
        out = np.empty((self.D, self.D))
        check(lib().sosrt_phase_matrix(self._h, self._KINDS[kind], float(g), _ptr(out)))
        return out

    # ---- multi-GPU gather over RCCL (one process per GPU) -------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte RCCL id, made by one rank and handed to the others by any means."""
        buf = ctypes.create_string_buffer(128)
        check(lib().sosrt_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        check(lib().sosrt_comm_init(self._h, int(rank), int(world), ctypes.c_char_p(unique_id)))

    def gather_device(self, root: int, counts, d_send: int, d_recv: int):
        """counts[world] doubles per rank; d_send / d_recv device addresses; enqueued on the handle's stream."""
        c = np.ascontiguousarray(counts, dtype=np.int64)
        check(lib().sosrt_gather(self._h, int(root), _ptr(c), ctypes.c_void_p(d_send) if d_send else None,
                                 ctypes.c_void_p(d_recv) if d_recv else None))

    def comm_destroy(self):
        check(lib().sosrt_comm_destroy(self._h))

    # ---- helper level ----------------------------------------------------------
    def limit_mu_down(self, rows, idx):
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        if rows.ndim != 2 or rows.shape[1] != self.N:
            raise ValueError("rows must be [R, nb_angles]")
        out = np.empty((rows.shape[0], idx))
        if idx:
            check(lib().sosrt_limit_mu_down(self._h, rows.shape[0], int(idx), _ptr(rows), _ptr(out)))
        return out

    def asymptotic_down(self, J, tau, lens, tau_t, mu):
        J = np.ascontiguousarray(J, dtype=np.float64)
        tau = _f64(tau, J.shape, "tau")
        R, stride = J.shape
        lens = _i32(lens, R, "len")
        tt = _f64(tau_t, (R,), "tau_t")
        m = _f64(mu, (R,), "mu")
        out = np.empty(R)
        check(lib().sosrt_asymptotic_down(self._h, R, stride, _ptr(lens), _ptr(J), _ptr(tau), _ptr(tt), _ptr(m), _ptr(out)))
        return out

    # ---- plan introspection (host only) ----------------------------------------
    def plan_weights(self):
        w = np.empty(self.D)
        check(lib().sosrt_plan_weights(self._h, _ptr(w)))
        return w

    def plan_fold(self, which=0):
        W = np.empty((self.D, self.D))
        check(lib().sosrt_plan_fold(self._h, which, _ptr(W)))
        return W

    def plan_fix_table(self, idx):
        s0, ns = ctypes.c_int(), ctypes.c_int()
        C = np.zeros((max(idx, 1), 5))
        check(lib().sosrt_plan_fix_table(self._h, int(idx), ctypes.byref(s0), ctypes.byref(ns), _ptr(C)))
        return s0.value, ns.value, C.reshape(-1)[: idx * ns.value].reshape(idx, ns.value).copy()

    def plan_launch(self, batch, live, surface="specular", zones=3, cus=0):
        """The kernels an order of the order loop launches for a batch of `batch` columns (up to `zones` zones each) of whose
        first column group `live` are live -- sosrt_plan_launch, the one function the order loop decides with; host only."""
        sf = {"none": _lib.SURFACE_NONE, "specular": _lib.SURFACE_SPECULAR, "lambertian": _lib.SURFACE_LAMBERTIAN,
              "lambertian_readme": _lib.SURFACE_LAMBERTIAN_README}[surface]
        out = (ctypes.c_int * 9)()
        check(lib().sosrt_plan_launch(self._h, int(batch), int(live), sf, int(zones), int(cus), out))
        keys = ("groups", "gemm", "tail_cols", "transport", "parts", "repair", "order_loop", "ol_parts", "ol_grid")
        return dict(zip(keys, list(out)))

    def microbench(self, which):
        """0: FP64 MFMA TFLOP/s, 1: streaming copy GB/s, 2: FP64 FMA TFLOP/s, measured on this device."""
        r = ctypes.c_double()
        check(lib().sosrt_microbench(self._h, int(which), ctypes.byref(r)))
        return r.value

    # ---- profiling ---------------------------------------------------------------
    def profile_enable(self, on=True):
        check(lib().sosrt_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        check(lib().sosrt_profile_reset(self._h))

    def profile_get(self, kernel):
        ms, cnt, w = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
        check(lib().sosrt_profile_get(self._h, kernel, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(w)))
        return ms.value, cnt.value


def fix_count(tau_ref: float, nb_angles: int) -> int:
    """Number of downward angles next to mu=0 the reference rewrites (I1_In:124-127)."""
    return lib().sosrt_plan_fix_count(float(tau_ref), int(nb_angles))
