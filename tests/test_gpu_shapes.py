"""More shapes through the HIP path: the BASELINE configs C3 (N=256) and C5 (L=400, N=256), direction
counts that are not a multiple of the wavefront, thin slabs, the order budget.  Needs an MI355X."""
import os

import numpy as np
import pytest

import sos_oracle as O
from sosrt import _lib, inputs
from sosrt.main import SOS_Aer_batch
from sosrt.solver import Solver
from util import RTOL, assert_close, rel_err

pytestmark = pytest.mark.gpu


def _oracle(mu0, taer, rho, L, N, P_atm, P_aer, z_up, z_down, alb_aer, surface="specular"):
    mu = inputs.direction_grid(N)
    P0a = inputs.phase_function("rayleigh", N, mu, mu0)[0]
    P0r = inputs.phase_function("hg", N, mu, mu0, 0.7)[0]
    col = O.make_column(mu0, 120, z_up, z_down, L, 0.124, taer, rho, 1.0, alb_aer, N, P0a, P_atm, P0r, P_aer, surface=surface)
    return O.solve_column(col, literal=False)


@pytest.mark.parametrize("L,N,zs,cols", [
    (200, 256, (25, 17), [(0.5, 0.12, 0.15)]),                                   # C3: EVA-like column, D = 512
    (400, 256, (15, 14), [(0.5, 0.0075, 0.15), (0.3, 0.05, 0.6)]),               # C5: wildfire slab (thin), L = 400
    (64, 48, (30, 10), [(0.7, 0.3, 0.2), (0.25, 1.0, 0.0)]),                     # 48 directions: one partly filled wave
    (50, 70, (40, 20), [(0.6, 0.2, 0.1), (0.9, 0.5, 0.2)]),                      # 70: extrapolation straddles two waves of a half row: split chunk-parallel kernel (part 0 holds them in one)
    (90, 192, (25, 17), [(0.4, 0.12, 0.3)]),                                     # three waves per sweep
    (600, 32, (25, 17), [(0.5, 0.12, 0.15)]),                                    # more than 64 chunks per sweep: every chunk takes the ring kernel's general body
    (520, 64, (60, 30), [(0.6, 0.3, 0.2)]),                                      # 65 chunks, zone boundaries far down
    # the WIDE instantiation of the chunk-parallel kernel's split form (round 4): odd N, N > 256, more than 64 chunks per sweep
    (100, 501, (25, 17), [(0.5, 0.12, 0.3), (0.8, 0.6, 0.0)]),                   # the reference's shipped direction count: eight workgroups per column, 4-byte stage fills
    (96, 300, (40, 20), [(0.6, 0.2, 0.1), (0.3, 1.0, 0.5)]),                     # five workgroups per column, a ragged last part
    (800, 128, (25, 17), [(0.5, 0.12, 0.15)]),                                   # the shipped layer count at N = 128: 100 chunks per sweep, two mask words
    (64, 257, (30, 10), [(0.7, 0.3, 0.2), (0.25, 1.0, 0.0)]),                    # odd, one direction into a fifth part
    (700, 129, (25, 17), [(0.5, 0.12, 0.15)]),                                   # odd N just above two parts, 88 chunks
])
def test_shapes_match_oracle(L, N, zs, cols):
    mu = inputs.direction_grid(N)
    P_atm = inputs.phase_function("rayleigh", N, mu, 0.5)[1]
    P_aer = inputs.phase_function("hg", N, mu, 0.5, 0.7)[1]
    m0, ta, rh = (np.array(x) for x in zip(*cols))
    r = SOS_Aer_batch(m0, ta, rh, tauStar_atm=0.124, alb_aer=0.97, nb_layers=L, nb_angles=N, z_up=zs[0], z_down=zs[1],
                      P_atm=P_atm, P_aer=P_aer, max_orders=200, raise_on_error=False)
    for b, (a, t, g) in enumerate(cols):
        try:
            ref = _oracle(a, t, g, L, N, P_atm, P_aer, zs[0], zs[1], 0.97)
        except IndexError:
            assert r.status[b] == _lib.COL_INDEXERROR
            continue
        assert r.status[b] == _lib.COL_OK and r.n[b] == ref.n, (b, r.status[b], r.n[b], ref.n)
        assert_close(r.I[b], ref.I, RTOL, "L=%d N=%d column %d" % (L, N, b))


def test_order_budget_is_reported():
    L, N = 40, 32
    mu = inputs.direction_grid(N)
    P = inputs.phase_function("iso", N, mu, 0.5)[1]
    r = SOS_Aer_batch([0.5, 0.5], [0.05, 1.0], [0.1, 0.7], nb_layers=L, nb_angles=N, z_up=60, z_down=20, P_atm=P, P_aer=P,
                      atm_phase_fun="iso", aer_phase_fun="iso", max_orders=6, raise_on_error=False)
    assert r.status[1] == _lib.COL_MAXORDERS and r.n[1] == 6
    ok = SOS_Aer_batch([0.5], [0.05], [0.1], nb_layers=L, nb_angles=N, z_up=60, z_down=20, P_atm=P, P_aer=P,
                       atm_phase_fun="iso", aer_phase_fun="iso", max_orders=64)
    if r.status[0] == _lib.COL_OK:
        assert r.n[0] == ok.n[0] and np.array_equal(r.I[0], ok.I[0])
    # the budget is the CALL's, not that of the cached handle, which was just remade for 64 orders (sosrt_set_order_budget)
    again = SOS_Aer_batch([0.5, 0.5], [0.05, 1.0], [0.1, 0.7], nb_layers=L, nb_angles=N, z_up=60, z_down=20, P_atm=P, P_aer=P,
                          atm_phase_fun="iso", aer_phase_fun="iso", max_orders=6, raise_on_error=False)
    assert again.status[1] == _lib.COL_MAXORDERS and again.n[1] == 6 and np.array_equal(again.I, r.I)
    s = Solver(L, N, max_orders=8, device=-1)
    with pytest.raises(ValueError):
        s.set_order_budget(9)
    s.close()


def test_saved_orders_on_device_path_and_stats():
    """solve_dev with an I_saved buffer (device pointers through torch)."""
    import torch
    L, N, B = 50, 32, 3
    mu = inputs.direction_grid(N)
    P0, P = inputs.phase_function("iso", N, mu, 0.5)
    iu, idn = inputs.slab_indices(120, 25, 17, L)
    taer = np.array([0.05, 0.12, 0.4])
    tau = np.stack([inputs.tau_profile(0.104, t, 120, 25, 17, L) for t in taer])
    s = Solver(L, N, max_batch=B, max_orders=40)
    dev = torch.device("cuda", 0)
    s.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    s.set_grid(mu)
    s.set_phase(P, P)
    s.set_columns(np.full(B, iu), np.full(B, idn), 0.5, 0.15, 1.0, 1.0, 0.104 / L, taer / (idn + 1 - iu), 0.104 + taer)
    d_tau = torch.from_numpy(tau).to(dev)
    d_P0 = torch.from_numpy(np.tile(P0, (B, 1))).to(dev)
    d_I = torch.empty((B, L, 2 * N), dtype=torch.float64, device=dev)
    d_sv = torch.zeros((B, 40, L, 2 * N), dtype=torch.float64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    s.solve_device(d_tau.data_ptr(), d_P0.data_ptr(), d_P0.data_ptr(), d_I.data_ptr(), d_I_saved=d_sv.data_ptr(),
                   d_n_orders=d_n.data_ptr())
    torch.cuda.synchronize()
    n = d_n.cpu().numpy()
    mo, so = s.last_solve_stats()
    assert so == int((n - 1).sum()) and mo >= n.max()
    I, sv = d_I.cpu().numpy(), d_sv.cpu().numpy()
    for b in range(B):
        assert_close(sv[b, :n[b]].sum(axis=0), I[b], 1e-13, "sum of saved orders")
        assert not sv[b, n[b]:].any()
    g = np.load(__import__("util").golden("g3_spec_C1_iso.npz")[0])
    assert n[1] == int(g["n"])
    assert_close(I[1], g["I"], RTOL, "C1 column through the device path")
    s.close()


def test_forcing_and_critical_albedo_drivers():
    """crit:377-410 on top of the batched solve; fluxes checked against the oracle's."""
    from sosrt import forcing
    L, N = 60, 64      # at N = 32 the reference itself raises IndexError for the bright columns of this test
    mu = inputs.direction_grid(N)
    P0a, Pa = inputs.phase_function("rayleigh", N, mu, 0.5)
    P0r, Pr = inputs.phase_function("hg", N, mu, 0.5, 0.7)
    kw = dict(nb_layers=L, nb_angles=N, z_up=40, z_down=12, phases=(P0a, Pa, P0r, Pr))
    taer, walb = [0.05, 0.3], [0.95, 0.8]
    got = forcing.toa_net_flux(0.5, 0.124, taer, 0.15, 1.0, walb, **kw)
    for b in range(2):
        col = O.make_column(0.5, 120, 40, 12, L, 0.124, taer[b], 0.15, 1.0, walb[b], N, P0a, Pa, P0r, Pr)
        I = O.solve_column(col, literal=False).I
        fd, fu = O.fluxes(I, mu, col.tau, N, 0.5, 0.15, beam_norm="crit")
        assert got[b] == pytest.approx(-fd[0] - fu[0], rel=1e-10)
    dF = forcing.radiative_forcing(0.5, 0.124, taer, 0.15, 1.0, walb, **kw)
    assert dF.shape == (2,) and np.all(np.isfinite(dF)) and np.all(dF != 0)
    # a conservative bright aerosol sends more light back (less net flux in); a dark one absorbs
    bright = forcing.radiative_forcing(0.5, 0.124, [0.3], 0.15, 1.0, [1.0], **kw)[0]
    dark = forcing.radiative_forcing(0.5, 0.124, [0.3], 0.15, 1.0, [0.3], **kw)[0]
    assert bright < dark
    wc = forcing.critical_albedo(0.5, 0.124, [0.1, 0.3], 0.15, 1.0, **kw)
    assert wc.shape == (2,) and np.all((wc > 0) & (wc < 1))


def test_large_ragged_batch_of_small_columns():
    """600 columns (more than two 256-wide blocks of the live-column searches, ragged last block) that
    converge at very different orders: batch invariance (reversed order gives the same bits) and a
    sample against the oracle."""
    import sos_oracle as O
    from sosrt import inputs
    from sosrt.main import SOS_Aer_batch
    from util import RTOL, assert_close
    L, N, B = 24, 64, 600
    rng = np.random.default_rng(7)
    mu0 = rng.uniform(0.15, 1.0, B)
    taer = 10 ** rng.uniform(-2.5, 0.4, B)
    rho = rng.uniform(0.0, 0.95, B)
    mu = inputs.direction_grid(N)
    P_atm = inputs.phase_function("rayleigh", N, mu, 0.5)[1]
    P_aer = inputs.phase_function("hg", N, mu, 0.5, 0.7)[1]
    kw = dict(tauStar_atm=0.124, alb_aer=0.97, nb_layers=L, nb_angles=N, z_up=40, z_down=12, P_atm=P_atm, P_aer=P_aer,
              max_orders=300, raise_on_error=False)
    r = SOS_Aer_batch(mu0, taer, rho, **kw)
    r2 = SOS_Aer_batch(mu0[::-1], taer[::-1], rho[::-1], **kw)
    assert np.array_equal(r2.n[::-1], r.n) and np.array_equal(r2.status[::-1], r.status)
    ok = r.status == 0
    assert ok.sum() > 0.5 * B and r.n[ok].max() > 3 * r.n[ok].min()
    assert np.array_equal(r2.I[::-1][ok], r.I[ok])
    for b in (0, 255, 256, 511, 512, 599):
        P0a = inputs.phase_function("rayleigh", N, mu, mu0[b])[0]
        P0r = inputs.phase_function("hg", N, mu, mu0[b], 0.7)[0]
        col = O.make_column(mu0[b], 120, 40, 12, L, 0.124, taer[b], rho[b], 1.0, 0.97, N, P0a, P_atm, P0r, P_aer)
        try:
            ref = O.solve_column(col, literal=False)
        except IndexError:
            assert r.status[b] == 1, b
            continue
        assert r.status[b] == 0 and r.n[b] == ref.n, (b, r.status[b], r.n[b], ref.n)
        assert_close(r.I[b], ref.I, RTOL, "column %d" % b)


def test_c5_sized_batch_of_the_hg_stand_in_on_device():
    """A batch of the SIZE of BASELINE's largest configuration -- 4096 columns, L = 400, N = 256 (33 GB of fields on the
    device) -- with the HG(0.7) stand-in on the EVA slab (the wildfire scenario itself: 64 of its columns against the oracle in
    tests/test_gpu_mie.py, the whole 4096-column sweep with a sampled column in bench.py's `extras.c5`), through the
    device-pointer entry point: every column converges, the run is deterministic, order counts are monotone along the sweep
    axes, and one sampled column matches the oracle."""
    import os
    import sys
    import torch
    import sos_oracle as O
    from sosrt import inputs
    from sosrt.solver import Solver
    from util import RTOL, assert_close
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    L, N = 400, 256
    w = bench.build_sweep(4096, L, N, 0, 1)
    B = w["B"]
    assert B == 4096
    dev = torch.device("cuda", 0)
    s = Solver(L, N, max_batch=B, max_orders=128)
    s.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    s.set_grid(w["mu"]); s.set_phase(w["P_atm"], w["P_aer"])
    s.set_columns(np.full(B, w["idx_up"]), np.full(B, w["idx_down"]), w["mu0"], w["rho"], 1.0, w["alb_aer"],
                  w["tau_atm"] / L, w["taer"] / (w["idx_down"] + 1 - w["idx_up"]), w["tau_atm"] + w["taer"])
    d_tau = torch.from_numpy(w["tau"]).to(dev)
    # P0(mu, mu0) of the 4096 columns built on the device (16 distinct mu0)
    d_mu0 = torch.from_numpy(np.ascontiguousarray(w["mu0"])).to(dev)
    d_P0a = torch.empty((B, 2 * N), dtype=torch.float64, device=dev); d_P0r = torch.empty((B, 2 * N), dtype=torch.float64, device=dev)
    s.phase_p0_device("rayleigh", d_mu0.data_ptr(), d_P0a.data_ptr(), B)
    s.phase_p0_device("hg", d_mu0.data_ptr(), d_P0r.data_ptr(), B, g=0.7)
    d_I = torch.empty((B, L, 2 * N), dtype=torch.float64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    digests = []
    for _ in range(2):
        d_I.zero_()              # on torch's default stream = the legacy default stream = what set_stream(0) names: ordered
        s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), d_I.data_ptr(), d_n_orders=d_n.data_ptr(),
                       d_status=d_st.data_ptr())
        torch.cuda.synchronize()
        digests.append((d_I[:, 0, N:].clone(), d_I[:, L - 1, :N].clone(), d_n.clone()))
    assert int(d_st.abs().sum().item()) == 0
    assert all(torch.equal(a, b) for a, b in zip(*digests))                 # bit-reproducible
    n = d_n.cpu().numpy().reshape(16, 16, 16)                                # axes: mu0, tau*_aer, grd_alb
    assert n.min() >= 2 and (np.diff(n, axis=1) >= 0).all() and (np.diff(n, axis=2) >= 0).all()
    assert torch.isfinite(d_I[::97]).all() and float(d_I[:, 0, N + 1:].min()) > 0
    b = 16 * 16 * 5 + 16 * 9 + 4
    mu = w["mu"]
    col = O.make_column(w["mu0"][b], 120, 25, 17, L, w["tau_atm"], w["taer"][b], w["rho"][b], 1.0, w["alb_aer"], N,
                        O.phase_rayleigh(N, mu, w["mu0"][b])[0], w["P_atm"], O.phase_hg(N, mu, w["mu0"][b], 0.7)[0], w["P_aer"])
    ref = O.solve_column(col, literal=False)
    assert int(d_n[b].item()) == ref.n
    assert_close(d_I[b].cpu().numpy(), ref.I, RTOL, "column %d" % b)
    s.close()


@pytest.mark.parametrize("stream", ["default", "own", "torch"])
def test_solve_is_ordered_against_the_callers_fills_without_a_synchronise(stream):
    """The failure of round 2 (gpurun_out/split_full.txt): a torch caller fills its buffers on torch's default stream, whose
    handle is 0, and solves without a synchronise.  sosrt_set_stream(h, 0) now names that stream (round 2: the handle's own
    non-blocking stream, which raced the fill), the handle's own stream is a blocking stream, and an explicit torch stream
    is ordered by the caller.  Large fills right before the solve, field and order counts bit-reproducible over repeats and
    equal to a synchronised run."""
    import torch
    import bench
    L, N = 200, 128
    w = bench.build_sweep(216, L, N, 0, 1)
    B = w["B"]
    dev = torch.device("cuda", 0)
    P0a_h, P0r_h = bench.host_p0(w)
    s = Solver(L, N, max_batch=B, max_orders=128)
    side = torch.cuda.Stream(device=dev) if stream == "torch" else None
    s.set_stream({"default": torch.cuda.current_stream(dev).cuda_stream, "own": None}.get(stream, side.cuda_stream if side else None))
    s.set_grid(w["mu"]); s.set_phase(w["P_atm"], w["P_aer"])
    s.set_columns(np.full(B, w["idx_up"]), np.full(B, w["idx_down"]), w["mu0"], w["rho"], 1.0, w["alb_aer"],
                  w["tau_atm"] / L, w["taer"] / (w["idx_down"] + 1 - w["idx_up"]), w["tau_atm"] + w["taer"])
    h_tau, h_P0a, h_P0r = (torch.from_numpy(np.ascontiguousarray(x)).pin_memory() for x in (w["tau"], P0a_h, P0r_h))
    d_tau = torch.empty((B, L), dtype=torch.float64, device=dev)
    d_P0a = torch.empty((B, 2 * N), dtype=torch.float64, device=dev); d_P0r = torch.empty_like(d_P0a)
    d_I = torch.empty((B, L, 2 * N), dtype=torch.float64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    junk = torch.empty((1 << 27,), dtype=torch.float64, device=dev)          # 1 GiB: the fills take a while
    outs = []
    for rep in range(3):
        ctx = torch.cuda.stream(side) if side is not None else torch.cuda.stream(torch.cuda.current_stream(dev))
        with ctx:
            # poison, then the real inputs, all asynchronous on the caller's stream, and NO synchronise before the solve
            junk.fill_(float(rep)); d_I.fill_(float("nan")); d_tau.fill_(float("nan")); d_n.fill_(-1)
            d_tau.copy_(h_tau, non_blocking=True); d_P0a.copy_(h_P0a, non_blocking=True); d_P0r.copy_(h_P0r, non_blocking=True)
            s.solve_device(d_tau.data_ptr(), d_P0a.data_ptr(), d_P0r.data_ptr(), d_I.data_ptr(), d_n_orders=d_n.data_ptr())
            # the caller reads on its stream right away, again without a synchronise of its own
            outs.append((d_I.clone(), d_n.clone()))
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    assert bool(torch.isfinite(outs[0][0]).all()) and int(outs[0][1].min()) >= 2
    ref = SOS_Aer_batch(w["mu0"], w["taer"], w["rho"], tauStar_atm=w["tau_atm"], alb_aer=w["alb_aer"], nb_layers=L, nb_angles=N,
                        P_atm=w["P_atm"], P_aer=w["P_aer"], P0_atm=P0a_h, P0_aer=P0r_h, max_orders=128)
    assert np.array_equal(outs[0][1].cpu().numpy(), ref.n)
    assert np.array_equal(outs[0][0].cpu().numpy(), ref.I)                  # the synchronous host-pointer solve, bit for bit
    s.close()


def test_many_distinct_slab_coefficients_fall_back_to_two_passes():
    """More distinct (ca, cr) slab coefficient pairs than the combined-matrix cache holds (32): the
    live-column slab tiles then take the two-pass path.  Same answers as with few pairs."""
    import sos_oracle as O
    from sosrt import inputs
    from sosrt.main import SOS_Aer_batch
    from util import RTOL, assert_close
    L, N, B = 40, 64, 48
    mu = inputs.direction_grid(N)
    P_atm = inputs.phase_function("rayleigh", N, mu, 0.5)[1]
    P_aer = inputs.phase_function("hg", N, mu, 0.5, 0.7)[1]
    mu0 = np.full(B, 0.6)
    taer = np.geomspace(0.02, 1.5, B)            # 48 distinct optical depths -> 48 distinct (f_atm, f_aer)
    rho = np.linspace(0.05, 0.7, B)
    walb = np.linspace(0.85, 1.0, B)
    kw = dict(tauStar_atm=0.124, alb_aer=walb, nb_layers=L, nb_angles=N, z_up=40, z_down=12, P_atm=P_atm, P_aer=P_aer,
              max_orders=200, raise_on_error=False)
    r = SOS_Aer_batch(mu0, taer, rho, **kw)
    P0a = inputs.phase_function("rayleigh", N, mu, 0.6)[0]
    P0r = inputs.phase_function("hg", N, mu, 0.6, 0.7)[0]
    checked = 0
    for b in (0, 17, 33, 47):
        col = O.make_column(0.6, 120, 40, 12, L, 0.124, taer[b], rho[b], 1.0, walb[b], N, P0a, P_atm, P0r, P_aer)
        try:
            ref = O.solve_column(col, literal=False)
        except IndexError:
            assert r.status[b] == 1, b
            continue
        assert r.status[b] == 0 and r.n[b] == ref.n, (b, r.status[b], r.n[b], ref.n)
        assert_close(r.I[b], ref.I, RTOL, "column %d" % b)
        checked += 1
    assert checked >= 2
    # the same columns in groups of 8 (few pairs per batch: the combined-matrix path) agree to rounding
    for g0 in (0, 40):
        sl = slice(g0, g0 + 8)
        rs = SOS_Aer_batch(mu0[sl], taer[sl], rho[sl], **dict(kw, alb_aer=walb[sl]))
        ok = (r.status[sl] == 0) & (rs.status == 0)
        assert np.array_equal(rs.n[ok], r.n[sl][ok])
        assert_close(rs.I[ok], r.I[sl][ok], 1e-12, "two-pass vs combined-matrix slab tiles")


def test_sharded_solve_equals_single_rank_bit_for_bit():
    """SOS_Aer_batch(devices=[0, 0]): two worker processes (gloo, sharing the one GPU of the test box) solve their
    shards of a ragged sweep and gather; a column's result does not depend on the batch it was solved in."""
    rng = np.random.default_rng(7)
    B = 37
    mu0 = rng.uniform(0.2, 1.0, B)
    taer = rng.choice([0.02, 0.1, 0.35, 0.9], B)          # four optical-depth profiles -> four slab coefficient pairs
    rho = rng.uniform(0.0, 0.8, B)
    kw = dict(tauStar_atm=0.124, alb_aer=0.95, nb_layers=40, nb_angles=64, max_orders=200)
    one = SOS_Aer_batch(mu0, taer, rho, **kw)
    two = SOS_Aer_batch(mu0, taer, rho, devices=[0, 0], **kw)
    assert np.array_equal(one.n, two.n) and np.array_equal(one.status, two.status)
    assert np.array_equal(one.tau, two.tau)
    assert np.array_equal(one.I, two.I)                    # bit for bit
    # and the same holds for a sub-batch solved on its own (different tilings of the contraction)
    sub = SOS_Aer_batch(mu0[:5], taer[:5], rho[:5], **kw)
    assert np.array_equal(sub.I, one.I[:5])


def test_two_column_groups_on_two_streams_are_bit_identical(monkeypatch):
    """SOSRT_GROUPS=2: the order loop runs per half of the batch, the second half on an internal stream, contraction and
    transport of the two halves side by side on the CUs.  Same bits as the single-group loop -- and the tilings differ, so this
    is also a batch-invariance check."""
    from sosrt import main as M
    rng = np.random.default_rng(11)
    B = 600
    mu0 = rng.uniform(0.2, 1.0, B)
    taer = rng.choice([0.02, 0.12, 0.6], B)
    rho = rng.uniform(0.0, 0.8, B)
    kw = dict(tauStar_atm=0.124, alb_aer=0.9, nb_layers=40, nb_angles=64, max_orders=200)
    out = []
    for groups in ("1", "2"):
        monkeypatch.setenv("SOSRT_GROUPS", groups)
        for s_ in list(M._solvers.values()):
            s_.close()
        M._solvers.clear()
        out.append(SOS_Aer_batch(mu0, taer, rho, **kw))
    for s_ in list(M._solvers.values()):
        s_.close()
    M._solvers.clear()
    a, b = out
    assert (a.status == 0).all()
    assert np.array_equal(a.n, b.n) and np.array_equal(a.status, b.status)
    assert np.array_equal(a.I, b.I)                         # bit for bit


def test_default_grouping_of_a_large_batch_keeps_the_bits(monkeypatch):
    """The library's own choice (SOSRT_GROUPS unset: two column groups above 256 columns) against one group, on an odd batch size
    (groups of 150 and 151 columns) of two-layer columns -- zone tables of five zones, the zone-table instantiation of the ring /
    chunk-parallel kernels, a surface that reflects -- and every column against a batch of its own kind solved alone."""
    from sosrt import main as M
    from sosrt.main import SOS_Aer_layers
    rng = np.random.default_rng(31)
    B = 301
    mu0 = rng.uniform(0.2, 1.0, B)
    rho = rng.uniform(0.0, 0.8, B)
    slabs = [(60, 50, 0.12, 0.90), (25, 17, 0.20, 0.97)]
    kw = dict(tauStar_atm=0.124, nb_layers=48, nb_angles=64, aer_phase_fun="hg", g_aer=0.7, max_orders=200)
    out = []
    for groups in ("1", None):
        if groups is None:
            monkeypatch.delenv("SOSRT_GROUPS", raising=False)
        else:
            monkeypatch.setenv("SOSRT_GROUPS", groups)
        for s_ in list(M._solvers.values()):
            s_.close()
        M._solvers.clear()
        out.append(SOS_Aer_layers(mu0, rho, slabs, **kw))
    sub = SOS_Aer_layers(mu0[148:153], rho[148:153], slabs, **kw)      # five columns across the cut, alone (one group)
    for s_ in list(M._solvers.values()):
        s_.close()
    M._solvers.clear()
    a, b = out
    assert (a.status == 0).all()
    assert np.array_equal(a.n, b.n) and np.array_equal(a.status, b.status) and np.array_equal(a.I, b.I)
    assert np.array_equal(sub.n, b.n[148:153]) and np.array_equal(sub.I, b.I[148:153])


def test_transport_kernel_follows_the_live_count_with_the_same_bits(monkeypatch):
    """Default mode: the ring kernel for launches with many live columns, the chunk-parallel kernel (one or two workgroups per
    column) for launches with few.  The two share their arithmetic (chunk-local recurrence), so (1) the ring kernel alone,
    the chunk-parallel kernel alone and any threshold between them give the same bits; (2) a column solved alone, in a
    sub-batch or in the whole batch has the same bits."""
    from sosrt import main as M
    rng = np.random.default_rng(23)
    B = 40
    mu0 = rng.uniform(0.2, 1.0, B)
    taer = rng.choice([0.02, 0.12, 0.6], B)
    rho = rng.uniform(0.0, 0.8, B)
    kw = dict(tauStar_atm=0.124, alb_aer=0.95, nb_layers=72, nb_angles=64, max_orders=200)

    def fresh(**env):
        for k in ("SOSRT_TRANSPORT", "SOSRT_SCAN_COLS", "SOSRT_SCAN_SPLIT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for s_ in list(M._solvers.values()):
            s_.close()
        M._solvers.clear()

    fresh(SOSRT_TRANSPORT="ring")
    ring = SOS_Aer_batch(mu0, taer, rho, **kw)
    assert (ring.status == 0).all() and ring.n.max() >= 14 and ring.n.min() < 14
    for env in ({"SOSRT_TRANSPORT": "scan"}, {"SOSRT_TRANSPORT": "scan", "SOSRT_SCAN_SPLIT": "0"}, {"SOSRT_SCAN_COLS": "12"}, {}):
        fresh(**env)
        whole = SOS_Aer_batch(mu0, taer, rho, **kw)
        assert np.array_equal(whole.n, ring.n) and (whole.status == 0).all()
        assert np.array_equal(whole.I, ring.I), env             # bit for bit
        sub = SOS_Aer_batch(mu0[7:19], taer[7:19], rho[7:19], **kw)
        assert np.array_equal(sub.I, whole.I[7:19])
        slow = int(np.argmax(ring.n))
        one = SOS_Aer_batch(mu0[slow:slow + 1], taer[slow:slow + 1], rho[slow:slow + 1], **kw)
        assert np.array_equal(one.I[0], whole.I[slow]) and one.n[0] == whole.n[slow]
    fresh()


@pytest.mark.parametrize("L,N,B,surface", [(200, 128, 1, "specular"), (200, 128, 24, "specular"), (72, 64, 40, "specular"),
                                           (200, 256, 3, "specular"), (50, 32, 7, "specular"), (90, 192, 5, "specular"),
                                           (72, 128, 9, "lambertian"), (41, 100, 6, "specular"), (200, 128, 150, "specular")])
def test_order_loop_kernel_keeps_the_bits(L, N, B, surface, monkeypatch):
    """The last orders of the last few live columns in ONE launch (csrc/order_loop.hip: the chunk-parallel transport of each live
    column and the tiles of their source functions as roles of one grid, tied by per-column counters) against the same solve
    with every order as two launches: same order counts, same statuses, the same bits -- for a lone column (the launch starts at
    the second order), for batches that enter it when enough columns have converged, in the split form (two and four
    workgroups per column), with one workgroup per column (N <= 64, a Lambertian surface) and on shapes whose last part or
    last chunk is ragged; and the launches really ran."""
    from sosrt import main as M
    rng = np.random.default_rng(1000 * L + 7 * N + B)
    mu0 = rng.uniform(0.2, 1.0, B)
    taer = rng.choice([0.02, 0.12, 0.6, 1.0], B)
    rho = rng.uniform(0.0, 0.8, B)
    kw = dict(tauStar_atm=0.124, alb_aer=0.95, nb_layers=L, nb_angles=N, max_orders=200, surface=surface, raise_on_error=False)
    out = {}
    for on in ("0", "1"):
        monkeypatch.setenv("SOSRT_ORDER_LOOP", on)
        for s_ in list(M._solvers.values()):
            s_.close()
        M._solvers.clear()
        out[on] = SOS_Aer_batch(mu0, taer, rho, **kw)
        launches = [s_.order_loop_stats() for s_ in M._solvers.values()]
        assert launches, "no cached solver"
        if on == "1":
            assert launches[0][0] >= 1 and launches[0][1] == 0, launches        # it ran, and was never refused
        else:
            assert launches[0][0] == 0
    for s_ in list(M._solvers.values()):
        s_.close()
    M._solvers.clear()
    a, b = out["0"], out["1"]
    assert np.array_equal(a.n, b.n), (a.n, b.n)
    assert np.array_equal(a.status, b.status)
    assert (a.status == 0).any()
    assert np.array_equal(a.I, b.I)                                            # bit for bit


@pytest.mark.parametrize("L,N,B,surface,pairs", [(200, 128, 1, "specular", 1), (200, 128, 9, "specular", 3), (200, 256, 3, "specular", 2),
                                                  (72, 64, 5, "lambertian", 2), (40, 64, 48, "specular", 48), (24, 128, 300, "specular", 4),
                                                  (35, 192, 4, "specular", 2)])
def test_register_resident_contraction_tile_keeps_the_bits(L, N, B, surface, pairs, monkeypatch):
    """The contraction of the last few live columns (csrc/jn_gemm_tile.hpp: gemm_tile_lone -- 16-row tiles, a lane's fragments of the
    folded matrix in registers, no barrier in the k-loop; api.hip: SOSRT_PLAN_GEMM_LIVE16_REGS) against the staged live-column
    tilings (SOSRT_GEMM_REGS=0): same order counts, same statuses, the same bits -- a lone column, several columns, N = 256 (more
    than 64 KB of LDS), a Lambertian surface, more slab coefficient pairs than combined matrices (two passes), a group of more than
    256 columns (two rounds of candidates in the search for the tile's column) and a shape whose last tiles are ragged."""
    from sosrt import main as M
    rng = np.random.default_rng(100 * L + N + B)
    mu0 = rng.uniform(0.2, 1.0, B)
    taer = np.geomspace(0.02, 1.2, pairs)[rng.integers(0, pairs, B)]
    rho = rng.uniform(0.0, 0.8, B)
    walb = np.linspace(0.85, 1.0, pairs)[rng.integers(0, pairs, B)] if pairs > 32 else 0.95
    if pairs > 32:
        taer = np.geomspace(0.02, 1.5, B)
    kw = dict(tauStar_atm=0.124, alb_aer=walb, nb_layers=L, nb_angles=N, max_orders=200, surface=surface, raise_on_error=False)
    out = {}
    for regs in ("0", "64"):
        monkeypatch.setenv("SOSRT_GEMM_REGS", regs)
        monkeypatch.setenv("SOSRT_GROUPS", "1")
        for s_ in list(M._solvers.values()):
            s_.close()
        M._solvers.clear()
        out[regs] = SOS_Aer_batch(mu0, taer, rho, **kw)
        plans = [s_.plan_launch(B, min(B, 2), surface=surface) for s_ in M._solvers.values()]
        assert plans and plans[0]["gemm"] == (_lib.PLAN_GEMM_LIVE16_REGS if regs == "64" else _lib.PLAN_GEMM_LIVE32_DEEP), plans
    for s_ in list(M._solvers.values()):
        s_.close()
    M._solvers.clear()
    a, b = out["0"], out["64"]
    assert np.array_equal(a.n, b.n), (a.n, b.n)
    assert np.array_equal(a.status, b.status)
    assert (a.status == 0).any()
    assert np.array_equal(a.I, b.I)                                            # bit for bit


@pytest.mark.parametrize("case", ["three_zones", "two_layers", "two_groups"])
def test_transport_over_the_dense_tilings_live_list_keeps_the_bits(case, monkeypatch):
    """While most columns of a group are live the contraction keeps its dense tiling, whose extra workgroup writes the live list
    (csrc/jn_gemm.hip: write_live_list), and the ring-class transport runs over that list -- its kernel chosen by the live count --
    instead of over every column of the group; columns of more than three zones, which have the dense tiling only, get their tail
    onto the chunk-parallel kernel that way.  Same order counts, statuses and bits as with SOSRT_DENSE_LIVE_LIST=0."""
    from sosrt import main as M
    from sosrt.main import SOS_Aer_layers
    rng = np.random.default_rng(5)
    B, L, N = (300, 40, 128) if case != "two_layers" else (90, 48, 128)
    mu0 = rng.uniform(0.2, 1.0, B); rho = rng.uniform(0.0, 0.8, B)
    taer = rng.choice([0.02, 0.12, 0.6, 1.0], B)
    out = {}
    for on in ("0", "1"):
        monkeypatch.setenv("SOSRT_DENSE_LIVE_LIST", on)
        monkeypatch.setenv("SOSRT_GROUPS", "2" if case == "two_groups" else "1")
        for s_ in list(M._solvers.values()):
            s_.close()
        M._solvers.clear()
        if case == "two_layers":
            out[on] = SOS_Aer_layers(mu0, rho, [(60, 50, 0.2, 0.90), (25, 17, 0.4, 0.97)], nb_layers=L, nb_angles=N, max_orders=200, raise_on_error=False)
        else:
            out[on] = SOS_Aer_batch(mu0, taer, rho, tauStar_atm=0.124, alb_aer=0.95, nb_layers=L, nb_angles=N, max_orders=200, raise_on_error=False)
    for s_ in list(M._solvers.values()):
        s_.close()
    M._solvers.clear()
    a, b = out["0"], out["1"]
    assert np.array_equal(a.n, b.n) and np.array_equal(a.status, b.status) and (a.status == 0).any()
    assert a.n.max() > a.n.min() + 3                                           # the columns do leave one by one
    assert np.array_equal(a.I, b.I)                                            # bit for bit


@pytest.mark.parametrize("B,groups", [(1, "1"), (7, "1"), (300, "2")])
def test_single_slab_batches_through_the_new_contraction_paths(B, groups, monkeypatch):
    """The single-slab geometry (I1_In:13-129: no aerosol rows, so no per-column slab descriptors and the group's operands offset by
    the launcher) through the register-resident tile and through the dense tiling's live list, two column groups included: same
    bits as with both switched off, and the first column against the oracle."""
    L, N = 30, 64
    rng = np.random.default_rng(B)
    mu = inputs.direction_grid(N)
    mu0 = rng.uniform(0.2, 1.0, B); alb = rng.uniform(0.6, 1.0, B); tstar = rng.choice([0.1, 0.3, 1.0, 2.0], B)
    P = inputs.phase_function("hg", N, mu, 0.5, 0.7)[1]
    P0 = np.stack([inputs.phase_function("hg", N, mu, m, 0.7)[0] for m in mu0])
    tau = np.stack([np.linspace(0.0, t, L) for t in tstar])
    out = {}
    for on in ("0", "1"):
        monkeypatch.setenv("SOSRT_GEMM_REGS", "0" if on == "0" else "64")
        monkeypatch.setenv("SOSRT_DENSE_LIVE_LIST", on)
        monkeypatch.setenv("SOSRT_GROUPS", groups)
        s = Solver(L, N, max_batch=B, max_orders=200)
        s.set_grid(mu); s.set_phase(P)
        s.set_columns_single_slab(mu0, alb, tstar)
        out[on] = s.solve(tau, P0)
        s.close()
    a, b = out["0"], out["1"]
    assert np.array_equal(a.n, b.n) and np.array_equal(a.status, b.status) and (a.status == 0).all()
    assert np.array_equal(a.I, b.I)
    ref = O.solve_single_slab(tau[0], mu, float(tstar[0]), float(mu0[0]), P0[0], P, float(alb[0]), N, literal=False)
    assert int(b.n[0]) == ref.n
    assert_close(b.I[0], ref.I, RTOL, "single slab, column 0")


def test_a_refused_order_loop_launch_hands_the_orders_back():
    """The residency handshake of an order-loop launch, made to fail on purpose (mode 2: a grid of twice the device's CUs): every
    workgroup that is on the machine waits 4 ms for the ones that are not, one of them declares the launch NOT RESIDENT, nothing
    has been touched, and the host runs the same orders as two launches each -- same bits, one refused launch per solve (no
    second attempt within a solve)."""
    rng = np.random.default_rng(17)
    B, L, N = 6, 72, 128
    mu0 = rng.uniform(0.2, 1.0, B); taer = rng.choice([0.02, 0.12, 0.6], B); rho = rng.uniform(0.0, 0.8, B)
    mu = inputs.direction_grid(N)
    P0a, Pa = inputs.phase_function("rayleigh", N, mu, 0.5); P0r, Pr = inputs.phase_function("hg", N, mu, 0.5, 0.7)
    P0a = np.stack([inputs.phase_function("rayleigh", N, mu, m)[0] for m in mu0]); P0r = np.stack([inputs.phase_function("hg", N, mu, m, 0.7)[0] for m in mu0])
    iu, idn = inputs.slab_indices(120, 25, 17, L)
    tau = np.stack([inputs.tau_profile(0.124, t, 120, 25, 17, L) for t in taer])
    out = {}
    for mode in (0, 2, 1):
        s = Solver(L, N, max_batch=B, max_orders=100)
        s.set_grid(mu); s.set_phase(Pa, Pr)
        s.set_columns(np.full(B, iu), np.full(B, idn), mu0, rho, 1.0, 0.95, 0.124 / L, taer / (idn + 1 - iu), 0.124 + taer)
        s.set_order_loop(mode)
        out[mode] = (s.solve(tau, P0a, P0r), s.order_loop_stats(True))
        s.close()
    ref = out[0][0]
    assert out[0][1] == (0, 0, 0)
    assert out[2][1] == (1, 1, 0), out[2][1]                 # launched once, refused, no (column, order) ran inside it
    assert out[1][1][0] == 1 and out[1][1][1] == 0 and out[1][1][2] == int((ref.n - 1).sum())
    for mode in (1, 2):
        r = out[mode][0]
        assert np.array_equal(r.n, ref.n) and np.array_equal(r.status, ref.status) and np.array_equal(r.I, ref.I), mode


_OL_WORKER = r"""
import json, os, sys
import numpy as np
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, "sos-radiative-transfer_amd"))
from sosrt import main as M
from sosrt.main import SOS_Aer_batch
rng = np.random.default_rng(5)
B = 24
mu0 = rng.uniform(0.2, 1.0, B); taer = rng.choice([0.02, 0.12, 0.6], B); rho = rng.uniform(0.0, 0.8, B)
kw = dict(tauStar_atm=0.124, alb_aer=0.95, nb_layers=200, nb_angles=128, max_orders=200)
os.environ["SOSRT_ORDER_LOOP"] = "0"
ref = SOS_Aer_batch(mu0, taer, rho, **kw)
for s_ in list(M._solvers.values()):
    s_.close()
M._solvers.clear()
os.environ["SOSRT_ORDER_LOOP"] = "1"
launches = refused = 0
for it in range(int(sys.argv[2])):
    r = SOS_Aer_batch(mu0, taer, rho, **kw)
    assert np.array_equal(r.n, ref.n) and np.array_equal(r.status, ref.status) and np.array_equal(r.I, ref.I), it
    st = list(M._solvers.values())[0].order_loop_stats()
    launches += st[0]; refused += st[1]
print(json.dumps({"launches": launches, "refused": refused}))
"""


def test_order_loop_launches_of_two_processes_on_one_gpu():
    """An order-loop launch is workgroups that wait for each other, so all of them must be on the machine -- which another process's
    kernels can prevent (the per-device CU budget only spans the handles of ONE process).  Two processes solve with
    SOSRT_ORDER_LOOP=1 on the one GPU at the same time, forty solves each: whenever a launch finds its grid not resident (its
    handshake times out; nothing has been touched) the host takes the orders back with two launches each, and every solve of both
    processes has the bits of the two-launch loop.  The refused launches are reported, not asserted (they depend on how the two
    processes' launches interleave)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", _OL_WORKER, root, "40"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    stats = [json.loads([l for l in so.splitlines() if l.startswith("{")][-1]) for so, _ in outs]
    print("order-loop launches / refused per process:", stats)
    assert all(st["launches"] >= 40 for st in stats)


@pytest.mark.parametrize("L,N", [(3, 8), (4, 64), (5, 128), (8, 32), (9, 128), (17, 100), (33, 64), (65, 128),
                                 (40, 256), (26, 192), (21, 200), (200, 256)])
def test_chunk_parallel_transport_on_small_and_ragged_shapes(L, N, monkeypatch):
    """The chunk-parallel kernel for every order on columns of fewer chunks than it has waves, ragged last chunks, one or
    two lane groups, one, two, three or four workgroups per column (N = 192, 200 -- a last part of 8 lanes -- and 256: the
    split form only), against the ring kernel: same order counts and statuses (IndexError, order budget), same bits."""
    from sosrt import main as M
    rng = np.random.default_rng(100 * L + N)
    B = 5
    mu0 = rng.uniform(0.2, 1.0, B)
    taer = rng.choice([0.02, 0.12, 0.6], B)
    rho = rng.uniform(0.0, 0.8, B)
    kw = dict(tauStar_atm=0.124, alb_aer=0.9, nb_layers=L, nb_angles=N, max_orders=120, raise_on_error=False, z_up=80, z_down=40)
    out = {}
    for mode in ("ring", "scan"):
        monkeypatch.setenv("SOSRT_TRANSPORT", mode)
        for s_ in list(M._solvers.values()):
            s_.close()
        M._solvers.clear()
        out[mode] = SOS_Aer_batch(mu0, taer, rho, **kw)
    for s_ in list(M._solvers.values()):
        s_.close()
    M._solvers.clear()
    a, b = out["ring"], out["scan"]
    assert np.array_equal(a.n, b.n) and np.array_equal(a.status, b.status)
    live = a.status == 0
    # where the ring kernel takes the shape the two share their arithmetic: the same bits.  N = 200: the rewritten mu -> 0-
    # directions straddle two waves of a half row, "ring" falls to the general kernel (the serial form of the recurrence) while the
    # split chunk-parallel kernel holds them in one wave (round 4): the same numbers to rounding
    monkeypatch.setenv("SOSRT_TRANSPORT", "ring")
    probe = Solver(L, N, device=-1)
    probe.set_grid(inputs.direction_grid(N))
    ring_runs = probe.plan_launch(B, B)["transport"] == _lib.PLAN_TRANSPORT_RING
    probe.close()
    monkeypatch.delenv("SOSRT_TRANSPORT")
    if live.any():
        if ring_runs:
            assert np.array_equal(b.I[live], a.I[live])
        else:
            assert rel_err(b.I[live], a.I[live]) <= 1e-12


def test_three_zone_columns_keep_their_bits_beside_columns_of_more_zones():
    """A batch that mixes (clear, slab, clear) columns with two-layer columns: the batch takes the zone-table instantiation of
    the ring / chunk-parallel kernels (the boundaries beyond the reference's two held in scalars as well), whose arithmetic for a
    three-zone column is the three-zone instantiation's: such a column has the bits it has in a batch of its own; the two-layer
    column matches the oracle."""
    L, N = 48, 64
    mu = inputs.direction_grid(N)
    P0a, Pa = inputs.phase_function("rayleigh", N, mu, 0.6)
    P0r, Pr = inputs.phase_function("hg", N, mu, 0.6, 0.7)
    tau1, r1, mix1, dta1 = inputs.tau_profile_slabs(0.124, [(25, 17, 0.3)], 120, L)
    two = [(60, 50, 0.2, 0.90), (25, 17, 0.4, 0.97)]
    c2 = O.make_column_slabs(0.6, 120, two, L, 0.124, 0.3, 1.0, N, P0a, Pa, P0r, Pr)
    nzmax = len(c2.zones)
    assert nzmax == 5 and len(r1) == 3

    def pad(x, fill=0):
        return list(x) + [fill] * (nzmax - len(x))

    s = Solver(L, N, max_batch=3, max_orders=100)
    s.set_grid(mu); s.set_phase(Pa, Pr)
    # the three-zone columns alone
    rho = [0.1, 0.5]
    s.set_columns_zones(np.tile(r1, (2, 1)), mix1, 0.6, rho, 1.0, 0.124 / L, 0.95, dta1, 0.424)
    alone = s.solve(np.tile(tau1, (2, 1)), np.tile(P0a, (2, 1)), np.tile(P0r, (2, 1)))
    # the same two around a two-layer column
    zr0 = np.array([pad(r1), [z.r0 for z in c2.zones], pad(r1)], dtype=np.int32)
    zmix = np.array([pad(mix1), [z.kind == "mix" for z in c2.zones], pad(mix1)], dtype=np.int32)
    zwr = np.array([pad([0.95 if m else 0.0 for m in mix1]), [z.alb_aer for z in c2.zones], pad([0.95 if m else 0.0 for m in mix1])])
    zdt = np.array([pad(dta1), [z.dtau_aer for z in c2.zones], pad(dta1)])
    s.set_columns_zones(zr0, zmix, 0.6, [rho[0], c2.grd_alb, rho[1]], 1.0, 0.124 / L, zwr, zdt, [0.424, c2.tauStar_tot, 0.424], nz=[3, 5, 3])
    mixed = s.solve(np.stack([tau1, c2.tau, tau1]), np.tile(P0a, (3, 1)), np.tile(P0r, (3, 1)))
    assert (mixed.status == 0).all() and (alone.status == 0).all()
    assert mixed.n[0] == alone.n[0] and mixed.n[2] == alone.n[1]
    assert np.array_equal(mixed.I[0], alone.I[0]) and np.array_equal(mixed.I[2], alone.I[1])      # bit for bit
    ref = O.solve_column(c2, literal=False)
    assert int(mixed.n[1]) == ref.n
    assert_close(mixed.I[1], ref.I, RTOL, "two-layer column beside three-zone columns")
    s.close()


def test_reference_shipped_size_L800_N501():
    """The size the reference ships (spec:33,57: nb_layers = 800, nb_angles = 501 -- odd and > 256, so the
    register-streaming transport kernel and the N >= 501 extrapolation tables) against the oracle, one column, at
    the north-star tolerance."""
    L, N = 800, 501
    mu = inputs.direction_grid(N)
    P0a, Pa = inputs.phase_function("rayleigh", N, mu, 0.5)
    P0r, Pr = inputs.phase_function("hg", N, mu, 0.5, 0.7)
    r = SOS_Aer_batch([0.5], [0.120], [0.3], tauStar_atm=0.104, alb_atm=1.0, alb_aer=1.0, nb_layers=L, nb_angles=N,
                      P_atm=Pa, P_aer=Pr, P0_atm=P0a[None], P0_aer=P0r[None], max_orders=64)
    c = O.make_column(0.5, 120, 25, 17, L, 0.104, 0.120, 0.3, 1.0, 1.0, N, P0a, Pa, P0r, Pr)
    ref = O.solve_column(c, literal=False)
    assert int(r.n[0]) == ref.n and int(r.status[0]) == 0
    assert_close(r.I[0], ref.I, RTOL, "I at the shipped size")
    # SOS_Aer(): the per-order history has n entries, not max_orders of them (1.6 GB per column otherwise)
    from sosrt.main import SOS_Aer
    one = SOS_Aer(nb_layers=L, nb_angles=N, grd_alb=0.3, P_atm=Pa, P0_atm=P0a, P_aer=Pr, P0_aer=P0r, max_orders=256)
    assert one.I_saved.shape == (ref.n, L, 2 * N) and one.n == ref.n
    assert_close(one.I_saved.sum(axis=0), one.I, 1e-13, "sum of the orders")


def test_the_literal_shipped_call():
    """`SOS_Aer()` with NO arguments: the literals of SOS_Aer_main_specular.py:23-96 -- 800 layers, 501 directions per hemisphere,
    a perfectly reflecting ground (grd_alb = 1), conservative scattering, Rayleigh + the EVA log-normal Mie aerosol -- up to an
    order budget of 8 (the column needs 13 orders; the oracle takes 7 s per order at this size), against the oracle on identical
    (P0, P): same order count, the budget reported, every order's field and their sum at 1e-10.  The transport is the WIDE
    instantiation of the chunk-parallel kernel (eight workgroups per column), with the |mu| < 0.01 lanes of the thin upper zone
    kept from k_smallmu (idx = 2 there: directions N-5 .. N-3 survive)."""
    from sosrt.main import SOS_Aer, DEFAULTS
    L, N = DEFAULTS["nb_layers"], DEFAULTS["nb_angles"]
    assert (L, N, DEFAULTS["grd_alb"], DEFAULTS["aer_phase_fun"]) == (800, 501, 1, "eva")
    from sosrt import main as M
    for s_ in list(M._solvers.values()):                     # (a cached handle of this shape would bring its own, larger, order budget)
        s_.close()
    M._solvers.clear()
    one = SOS_Aer(max_orders=8)
    mu = inputs.direction_grid(N)
    P0a, Pa = inputs.phase_function_device("rayleigh", N, mu, 0.5)
    P0r, Pr = inputs.phase_function_device("eva", N, mu, 0.5)
    c = O.make_column(0.5, 120, 25, 17, L, 0.104, 0.120, 1.0, 1.0, 1.0, N, P0a, Pa, P0r, Pr)
    ref = O.solve_column(c, literal=False, max_orders=8)
    assert one.n == ref.n == 8 and one.status == _lib.COL_MAXORDERS
    assert np.array_equal(one.tau, c.tau) and (one.idx_up, one.idx_down) == (c.idx_up, c.idx_down)
    assert_close(one.I, ref.I, RTOL, "I of the shipped call after 8 orders")
    assert one.I_saved.shape == (8, L, 2 * N)
    for k in range(8):
        assert_close(one.I_saved[k], ref.I_saved[k], RTOL, "order %d of the shipped call" % (k + 1))


def test_rccl_entry_points_world_of_one():
    """sosrt_comm_unique_id / sosrt_comm_init / sosrt_gather / sosrt_comm_destroy: RCCL bound at run time.  One GPU
    here, so a communicator of one rank; the driver's multi-GPU runs use torch.distributed over the same RCCL."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    s = Solver(8, 8, max_batch=2, max_orders=2)
    s.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    uid = Solver.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    s.comm_init(0, 1, uid)
    with pytest.raises(_lib.SosrtError):
        s.comm_init(0, 1, uid)                       # one communicator per handle
    x = torch.arange(1000, dtype=torch.float64, device=dev) * 0.5
    y = torch.zeros(1000, dtype=torch.float64, device=dev)
    s.gather_device(0, [1000], x.data_ptr(), y.data_ptr())
    s.synchronize()
    assert torch.equal(x, y)
    with pytest.raises(ValueError):
        s.gather_device(3, [1000], x.data_ptr(), y.data_ptr())
    s.comm_destroy()
    s.close()


def test_sharded_solve_through_the_abi_gather_world_of_one():
    """`dist.solve_sharded(gather="abi")`: the shard is solved into device memory, packed, sent through sosrt_gather
    (RCCL; a communicator of ONE rank here -- RCCL refuses two ranks on one GPU, so more than one rank of this path stays
    unmeasured on this box) into the root's device buffer, restored to global order, and copied to the host once.  In a
    child process (it initialises a process group); same bits as `SOS_Aer_batch`."""
    import subprocess
    import sys
    code = r"""
import os, sys
import numpy as np
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, 'sos-radiative-transfer_amd'))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2], RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from sosrt.main import SOS_Aer_batch
from sosrt.dist import solve_sharded
rng = np.random.default_rng(11)
B = 9
mu0, taer, rho = rng.uniform(0.2, 1.0, B), rng.choice([0.02, 0.3, 0.9], B), rng.uniform(0.0, 0.8, B)
kw = dict(tauStar_atm=0.124, alb_aer=0.95, nb_layers=40, nb_angles=64, max_orders=200)
one = SOS_Aer_batch(mu0, taer, rho, **kw)
for via in ('abi', 'torch'):
    two = solve_sharded(mu0, taer, rho, gather=via, **kw)
    assert np.array_equal(one.n, two.n) and np.array_equal(one.status, two.status) and np.array_equal(one.tau, two.tau), via
    assert np.array_equal(one.I, two.I), via
dist.destroy_process_group()
print('SHARDED_OK')
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code, root, str(30500 + os.getpid() % 1000)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARDED_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_float_contraction_is_an_opt_in_that_misses_the_parity_bar():
    """BASELINE configs[4] (fp64 -> fp32 mixed with tolerance study) on the device: the float contraction runs, gives the
    same order counts on this sweep, and lands 1e-8 .. 1e-5 away from the fp64 result -- outside the 1e-10 bar, which
    is why it is not the default."""
    import io
    import study_mixed_precision_gpu as S
    r = S.study(64, 60, 64, out=io.StringIO(), steps=1)
    # typical element 1e-8 .. 1e-5 away; isolated rows further, where the perturbation moves a data-dependent search (spec:403-406)
    assert 1e-9 < r["med"] < 1e-5 and 1e-9 < r["p999"] < 1e-3 and r["rel"] > 1e-10
    # and the step-level entry point follows the mode
    N, L = 32, 20
    mu = inputs.direction_grid(N)
    P = inputs.phase_function("hg", N, mu, 0.5, 0.6)[1]
    s = Solver(L, N, max_batch=2, max_orders=4)
    s.set_grid(mu); s.set_phase(P)
    s.set_columns_single_slab([0.5, 0.7], 0.9, 0.3)
    x = np.random.default_rng(0).uniform(0.1, 1.0, (2, L, 2 * N))
    j64 = s.source(x)
    s.set_contraction("f32")
    j32 = s.source(x)
    s.set_contraction("f64")
    assert np.array_equal(s.source(x), j64)
    e = np.max(np.abs(j32 - j64)) / np.max(np.abs(j64))
    assert 1e-9 < e < 1e-5
    s.close()


@pytest.mark.parametrize("L,N,slabs", [
    (96, 201, [(60, 50, 0.2, 0.90), (25, 17, 0.4, 0.97)]),                          # odd N, five zones: four workgroups per column
    (80, 300, [(80, 70, 0.05, 1.0), (55, 40, 0.3, 0.9), (25, 10, 0.6, 0.85)]),      # N > 256, seven zones
    (560, 66, [(100, 90, 0.1, 0.95), (12, 5, 0.5, 0.9)]),                           # 70 chunks per sweep: boundaries in both mask words
])
def test_zone_table_columns_at_the_wide_shapes(L, N, slabs):
    """Columns with several aerosol layers at the shapes of the chunk-parallel kernel's WIDE instantiation (odd N, N > 256, more than
    64 chunks per sweep): the launch plan takes that kernel with the whole zone table (no longer the register-streaming fallback),
    and the field matches the oracle's zone-table path (parity unpinned by construction: the reference has one layer)."""
    from sosrt.main import SOS_Aer_layers, get_solver
    mu = inputs.direction_grid(N)
    m0 = np.array([0.45, 0.8]); rho = np.array([0.1, 0.5])
    r = SOS_Aer_layers(m0, rho, slabs, nb_layers=L, nb_angles=N, max_orders=120)
    nz = 2 * len(slabs) + 1
    p = get_solver(L, N, 2, 120, 0).plan_launch(2, 2, zones=nz)
    assert (p["transport"], p["parts"]) == (_lib.PLAN_TRANSPORT_SCAN, (N + 63) // 64)
    Pa = inputs.phase_function("rayleigh", N, mu, 0.5)[1]; Pr = inputs.phase_function("hg", N, mu, 0.5, 0.7)[1]
    for i in range(2):
        P0a = inputs.phase_function("rayleigh", N, mu, m0[i])[0]; P0r = inputs.phase_function("hg", N, mu, m0[i], 0.7)[0]
        c = O.make_column_slabs(m0[i], 120, slabs, L, 0.124, rho[i], 1.0, N, P0a, Pa, P0r, Pr)
        ref = O.solve_column(c, literal=False)
        assert int(r.n[i]) == ref.n and int(r.status[i]) == 0
        assert_close(r.I[i], ref.I, RTOL, "%d-zone column %d at L=%d, N=%d" % (nz, i, L, N))


def test_zone_table_columns():
    """columns described by a zone table.  (clear, slab, clear) through sosrt_set_columns_zones equals
    sosrt_set_columns bit for bit; columns with two aerosol layers (five zones; the ring / chunk-parallel kernels in their
    zone-table instantiation, every transport mode in the randomised test below) match the oracle's zone-table path, fluxes and
    heating rate included (the 'erase_pics' fix-up at both layers).  More than one slab is parity unpinned by construction (the reference has one)."""
    L, N, B = 48, 64, 5
    mu = inputs.direction_grid(N)
    P0a, Pa = inputs.phase_function("rayleigh", N, mu, 0.6)
    P0r, Pr = inputs.phase_function("hg", N, mu, 0.6, 0.7)
    # (a) the reference's three zones, both entry points
    tau, r0, mix, dta = inputs.tau_profile_slabs(0.124, [(25, 17, 0.3)], 120, L)
    iu, idn = inputs.slab_indices(120, 25, 17, L)
    assert np.array_equal(tau, inputs.tau_profile(0.124, 0.3, 120, 25, 17, L)) and list(r0) == [0, iu, idn + 1]
    s = Solver(L, N, max_batch=B, max_orders=100)
    s.set_grid(mu); s.set_phase(Pa, Pr)
    rho = np.linspace(0.0, 0.6, B)
    s.set_columns(np.full(B, iu), np.full(B, idn), 0.6, rho, 1.0, 0.95, 0.124 / L, dta[1], 0.424)
    a = s.solve(np.tile(tau, (B, 1)), np.tile(P0a, (B, 1)), np.tile(P0r, (B, 1)))
    s.set_columns_zones(np.tile(r0, (B, 1)), mix, 0.6, rho, 1.0, 0.124 / L, 0.95, dta, 0.424)
    b = s.solve(np.tile(tau, (B, 1)), np.tile(P0a, (B, 1)), np.tile(P0r, (B, 1)))
    assert np.array_equal(a.n, b.n) and np.array_equal(a.I, b.I)
    # (b) two aerosol layers with different albedos and optical depths, different per column
    slabs = [[(60, 50, 0.2, 0.90), (25, 17, 0.4, 0.97)], [(70, 55, 0.05, 1.0), (30, 10, 1.2, 0.85)], [(60, 50, 0.6, 0.95), (25, 17, 0.1, 0.8)]]
    cols = [O.make_column_slabs(0.6, 120, sl, L, 0.124, 0.1 + 0.2 * i, 1.0, N, P0a, Pa, P0r, Pr) for i, sl in enumerate(slabs)]
    Bz = len(cols)
    zr0 = np.array([[z.r0 for z in c.zones] for c in cols], dtype=np.int32)
    zmix = np.array([[z.kind == "mix" for z in c.zones] for c in cols], dtype=np.int32)
    zwr = np.array([[z.alb_aer for z in c.zones] for c in cols])
    zdt = np.array([[z.dtau_aer for z in c.zones] for c in cols])
    for c, sl in zip(cols, slabs):      # the product's grid builder gives the oracle's grid and table
        t2, r2, m2, d2 = inputs.tau_profile_slabs(0.124, [x[:3] for x in sl], 120, L)
        assert np.array_equal(t2, c.tau) and list(r2) == [z.r0 for z in c.zones] and np.array_equal(d2, [z.dtau_aer for z in c.zones])
    s.set_columns_zones(zr0, zmix, 0.6, [c.grd_alb for c in cols], 1.0, 0.124 / L, zwr, zdt, [c.tauStar_tot for c in cols])
    tau3 = np.stack([c.tau for c in cols])
    r = s.solve(tau3, np.tile(P0a, (Bz, 1)), np.tile(P0r, (Bz, 1)), fetch_field=True)
    z = np.linspace(120, 0, L)
    e = s.epilogue(z_profile=z)
    for i, c in enumerate(cols):
        ref = O.solve_column(c, literal=False)
        assert int(r.n[i]) == ref.n and int(r.status[i]) == 0
        assert_close(r.I[i], ref.I, RTOL, "two-slab column %d" % i)
        fd, fu = O.fluxes(ref.I, mu, c.tau, N, 0.6, c.grd_alb)
        assert_close(e["flux_down"][i], fd, RTOL, "flux down")
        pairs = [(zn.r0, zn.r1) for zn in c.zones if zn.kind == "mix"]
        hr = O.heating_rate(ref.I, mu, c.tau, N, 0.6, c.grd_alb, z, 0, 0, slabs=pairs)
        scale = np.max(np.abs(fd + fu)) / (1.225 * 1004 * abs(z[1] - z[0]))
        assert np.max(np.abs(e["heating_rate"][i] - hr)) <= RTOL * scale
        for iu2, id2 in pairs:
            assert e["heating_rate"][i][iu2 - 1] == e["heating_rate"][i][iu2 - 2] and e["heating_rate"][i][id2] == e["heating_rate"][i][id2 - 1]
    # the column-level entry point: one layer = SOS_Aer_batch bit for bit (P0 built on the device for both here), two layers = oracle
    from sosrt.main import SOS_Aer_layers
    m0 = np.array([0.35, 0.6, 0.9])
    one = SOS_Aer_layers(m0, 0.2, [(25, 17, 0.3, 0.95)], nb_layers=L, nb_angles=N, max_orders=100)
    P0a3 = s.phase_p0("rayleigh", m0); P0r3 = s.phase_p0("hg", m0, 0.7)
    ref1 = SOS_Aer_batch(m0, 0.3, 0.2, alb_aer=0.95, nb_layers=L, nb_angles=N, max_orders=100, P0_atm=P0a3, P0_aer=P0r3)
    assert np.array_equal(one.I, ref1.I) and np.array_equal(one.n, ref1.n)
    two = SOS_Aer_layers([0.6], [0.1], slabs[0], nb_layers=L, nb_angles=N, max_orders=100)
    assert_close(two.I[0], O.solve_column(cols[0], literal=False).I, RTOL, "SOS_Aer_layers, two layers")
    s.close()
    s = Solver(L, N, max_batch=B, max_orders=100)
    s.set_grid(mu); s.set_phase(Pa, Pr)
    # argument checks
    with pytest.raises(ValueError):
        s.set_columns_zones([[0, 10, 20]], [[1, 0, 0]], 0.6, 0.1, 1.0, 0.124 / L, 0.9, 0.01, 0.3)      # aerosol zone at the top
    with pytest.raises(ValueError):
        s.set_columns_zones([[0, 20, 10]], [[0, 1, 0]], 0.6, 0.1, 1.0, 0.124 / L, 0.9, 0.01, 0.3)      # not ascending
    s.close()


def test_randomised_parity_across_transport_kernels():
    """tools/fuzz_parity.py: random zone positions, optical depths, albedos, surfaces and N in {64, 100, 128, 192, 256}, phase
    matrices whose rows oscillate over a random number of upward directions with a random amplitude (long searches in some rows
    and orders, short ones in others: rows finished one by one, the full redo, the fast path): ring == chunk-parallel bit for
    bit, every transport kernel against the oracle at 1e-10 with equal order counts or the same IndexError."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    assert fuzz_parity.run(cases=14, seed=20251005, verbose=False) == 0
    # columns of two or three aerosol layers (five / seven zones) through the same kernels, against the oracle's zone tables
    assert fuzz_parity.run_layers(cases=6, seed=20251005, verbose=False) == 0
