"""The row-parallel interior-point LamMuZ kernel (rda_planner_amd/csrc/lammuz_ip_device.h) is ONE template, instantiated on the
device with DPP lane operations and here - tests/emu/rip_emu.cpp - with a 16-wide host vector class: the very arithmetic of the
kernel runs on the CPU and is pinned against the oracle's interior-point restatement (oracle/lmz_ipm.c, itself pinned on the
reference's one-stage problems in tests/test_reference_pinned.py).  Both end on the central path at the same barrier parameter,
which is a unique point: they must agree to the accuracy the centring test leaves (1e-7 relative in s o z)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers as hp
from rda_planner_amd import scenarios as sc
from rda_planner_amd._capi import c_double_p, c_int_p, dptr

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    src, out = os.path.join(HERE, "emu", "rip_emu.cpp"), os.path.join(HERE, "emu", "_build", "librip_emu.so")
    hdr = os.path.join(HERE, "..", "rda_planner_amd", "csrc", "lammuz_ip_device.h")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", out, src])
    lib = C.CDLL(out)
    lib.rip_emu_solve.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, C.c_int, C.c_int, c_double_p, C.c_double, c_double_p, c_double_p,
                                  c_double_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p,
                                  c_double_p, C.c_int, c_int_p]
    lib.rip_emu_solve.restype = C.c_int
    return lib


def _oracle(orc):
    L = orc.lib
    L.orc_lammuz_ipm_one.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, C.c_int, C.c_int, c_double_p, C.c_double, c_double_p,
                                     c_double_p, c_double_p, C.c_double, C.c_double, C.c_double, C.c_int, c_double_p, c_double_p,
                                     c_double_p, c_double_p, c_int_p]
    L.orc_lammuz_ipm_one.restype = C.c_int
    L.orc_set_lmz_ipm_mu.argtypes = [C.c_double]
    return L


def _case(rng, E, circle_obstacle):
    p = rng.uniform(-5, 5, 2)
    phi = rng.uniform(-np.pi, np.pi)
    dist = rng.choice([0.5, 1.5, 3.0, 6.0, 15.0])
    th = rng.uniform(0, 2 * np.pi)
    cen = p + dist * np.array([np.cos(th), np.sin(th)])
    if circle_obstacle:
        A = np.zeros((E, 2)); A[0] = [1, 0]; A[1] = [0, 1]
        b = np.zeros(E); b[0:2] = cen; b[2] = -rng.uniform(0.3, 1.5)
    else:
        A, b = hp.random_polygon(rng, cen, int(rng.integers(3, E + 1)), rng.uniform(0.4, 2.0), E)
    return np.ascontiguousarray(A), np.ascontiguousarray(b), p, phi, rng.normal(0, 0.3, 2), rng.normal(0, 0.5), rng.uniform(0.1, 1.0)


ROBOTS = {"rectangle": (hp.G, hp.H, 0),
          "circle": (np.ascontiguousarray([[1.0, 0.0], [0.0, 1.0], [0.0, 0.0]]), np.ascontiguousarray([0.0, 0.0, -0.8]), 1)}


@pytest.mark.parametrize("robot", ["rectangle", "circle"])
@pytest.mark.parametrize("accelerated", [1, 0])
@pytest.mark.parametrize("mu", [1e-3, 1e-6])
def test_row_parallel_kernel_arithmetic_equals_the_oracle_on_the_central_path(emu, orc, robot, accelerated, mu):
    L = _oracle(orc)
    L.orc_set_lmz_ipm_mu(mu)
    G, h, rn2 = ROBOTS[robot]
    R = G.shape[0]
    rng = np.random.default_rng(7 + 13 * accelerated + (0 if robot == "rectangle" else 101))
    worst, n_ok = 0.0, 0
    try:
        for trial in range(60):
            circ = trial % 3 == 2
            E = 4
            A, b, p, phi, xi, zeta, dbar = _case(rng, E, circ)
            lo, mo, zo, cmh, it = np.zeros(E), np.zeros(R), C.c_double(0), np.zeros(4), C.c_int(0)
            st_o = L.orc_lammuz_ipm_one(E, R, dptr(A), dptr(b), int(circ), rn2, dptr(np.ascontiguousarray(p)), float(phi), dptr(G), dptr(h),
                                        dptr(np.ascontiguousarray(xi)), float(zeta), float(dbar), 1.0, accelerated, dptr(lo), dptr(mo),
                                        C.cast(C.byref(zo), c_double_p), dptr(cmh), C.cast(C.byref(it), c_int_p))
            le, me, ze, xe = np.zeros(E), np.zeros(R), C.c_double(0), np.zeros(16)
            st_e = emu.rip_emu_solve(E, R, dptr(A), dptr(b), int(circ), rn2, dptr(np.ascontiguousarray(p)), float(phi), dptr(G), dptr(h),
                                     dptr(np.ascontiguousarray(xi)), float(zeta), float(dbar), 1.0, accelerated, mu, dptr(le), dptr(me),
                                     C.cast(C.byref(ze), c_double_p), dptr(xe), None, 0, None)
            assert (st_o == 2) == (st_e == 2), (trial, st_o, st_e)
            if st_o == 2:
                continue
            n_ok += 1
            err = max(np.abs(le - lo).max(), np.abs(me - mo).max(), abs(ze.value - zo.value))
            scale = 1.0 + max(np.abs(lo).max(), np.abs(mo).max(), abs(zo.value))
            worst = max(worst, err / scale)
            assert err <= 2e-6 * scale, (trial, circ, err, lo, le, mo, me, zo.value, ze.value)
    finally:
        L.orc_set_lmz_ipm_mu(1e-6)
    assert n_ok >= 50, n_ok


@pytest.mark.parametrize("robot", ["rectangle", "circle"])
def test_warm_start_from_the_previous_central_point_reaches_the_same_point_in_fewer_iterations(emu, robot):
    """the kernel keeps (x, s, z) of every (slot, stage) and starts the next ADMM iteration's solve there (the cones do not change,
    so the point stays interior; its gap is deg mu*, so the iteration is in its centring phase at once): same end point as the cold
    start, a fraction of the iterations where consecutive problems are close, and still the same point when they are far apart"""
    G, h, rn2 = ROBOTS[robot]
    R = G.shape[0]
    rng = np.random.default_rng(5)
    mu = 1e-3
    tot_cold = tot_warm = 0
    for trial in range(40):
        circ = trial % 3 == 2
        A, b, p, phi, xi, zeta, dbar = _case(rng, 4, circ)
        state = np.zeros(80)
        le, me, ze, it0 = np.zeros(4), np.zeros(R), C.c_double(0), C.c_int(0)
        args = lambda p_, phi_, xi_, zeta_: (4, R, dptr(A), dptr(b), int(circ), rn2, dptr(np.ascontiguousarray(p_)), float(phi_), dptr(G), dptr(h),
                                             dptr(np.ascontiguousarray(xi_)), float(zeta_), float(dbar), 1.0, 1, mu)
        st = emu.rip_emu_solve(*args(p, phi, xi, zeta), dptr(le), dptr(me), C.cast(C.byref(ze), c_double_p), None, dptr(state), 0, C.cast(C.byref(it0), c_int_p))
        if st != 0:
            continue
        # the next ADMM iteration: pose and multipliers have moved a little (every tenth trial: a lot)
        far = trial % 10 == 9
        sc_ = 20.0 if far else 1.0
        p2, phi2 = p + sc_ * rng.normal(0, 0.03, 2), phi + sc_ * rng.normal(0, 0.01)
        xi2, zeta2 = xi + sc_ * rng.normal(0, 0.02, 2), zeta + sc_ * rng.normal(0, 0.05)
        lc, mc, zc, itc = np.zeros(4), np.zeros(R), C.c_double(0), C.c_int(0)
        lw, mw, zw, itw = np.zeros(4), np.zeros(R), C.c_double(0), C.c_int(0)
        sc0 = emu.rip_emu_solve(*args(p2, phi2, xi2, zeta2), dptr(lc), dptr(mc), C.cast(C.byref(zc), c_double_p), None, dptr(np.zeros(80)), 0, C.cast(C.byref(itc), c_int_p))
        sw0 = emu.rip_emu_solve(*args(p2, phi2, xi2, zeta2), dptr(lw), dptr(mw), C.cast(C.byref(zw), c_double_p), None, dptr(state), 1, C.cast(C.byref(itw), c_int_p))
        assert sc0 == sw0 == 0, (trial, sc0, sw0)
        scale = 1.0 + max(np.abs(lc).max(), np.abs(mc).max(), abs(zc.value))
        assert max(np.abs(lw - lc).max(), np.abs(mw - mc).max(), abs(zw.value - zc.value)) <= 2e-6 * scale, (trial, far, lc, lw)
        if not far:
            tot_cold += itc.value; tot_warm += itw.value
    print(f"{robot}: interior-point iterations cold {tot_cold}, warm {tot_warm}")
    assert tot_warm < 0.6 * tot_cold, (tot_warm, tot_cold)


def test_warm_start_from_another_obstacles_point_is_harmless(emu):
    """the obstacle list is re-ordered every tick (mpc.py:205-206): the kept point of a slot may belong to a different obstacle - a
    triangle (one zero-padded edge row) where a quadrilateral was, a circle where a polygon was.  The answer must be the cold one."""
    G, h, rn2 = ROBOTS["rectangle"]
    rng = np.random.default_rng(9)
    mu = 1e-6
    for trial in range(40):
        kind_a, kind_b = trial % 3, (trial // 3) % 3                 # 0 quadrilateral-ish, 1 triangle, 2 circle
        def make(kind):
            while True:
                A, b, p, phi, xi, zeta, dbar = _case(rng, 4, kind == 2)
                nz = int(np.sum(np.any(A != 0, axis=1)))
                if kind == 2 or (kind == 1 and nz == 3) or (kind == 0 and nz == 4):
                    return A, b, p, phi, xi, zeta, dbar
        Aa, ba, p, phi, xi, zeta, dbar = make(kind_a)
        Ab, bb = make(kind_b)[0:2]
        state = np.zeros(80)
        out = lambda: (np.zeros(4), np.zeros(4), C.c_double(0), C.c_int(0))
        l0, m0, z0, i0 = out()
        st = emu.rip_emu_solve(4, 4, dptr(Aa), dptr(ba), int(kind_a == 2), rn2, dptr(np.ascontiguousarray(p)), float(phi), dptr(G), dptr(h), dptr(np.ascontiguousarray(xi)),
                               float(zeta), float(dbar), 1.0, 1, mu, dptr(l0), dptr(m0), C.cast(C.byref(z0), c_double_p), None, dptr(state), 0, C.cast(C.byref(i0), c_int_p))
        if st != 0:
            continue
        lc, mc, zc, ic = out(); lw, mw, zw, iw = out()
        args = (4, 4, dptr(Ab), dptr(bb), int(kind_b == 2), rn2, dptr(np.ascontiguousarray(p)), float(phi), dptr(G), dptr(h), dptr(np.ascontiguousarray(xi)),
                float(zeta), float(dbar), 1.0, 1, mu)
        sc0 = emu.rip_emu_solve(*args, dptr(lc), dptr(mc), C.cast(C.byref(zc), c_double_p), None, dptr(np.zeros(80)), 0, C.cast(C.byref(ic), c_int_p))
        sw0 = emu.rip_emu_solve(*args, dptr(lw), dptr(mw), C.cast(C.byref(zw), c_double_p), None, dptr(state.copy()), 1, C.cast(C.byref(iw), c_int_p))
        assert sc0 == sw0, (trial, kind_a, kind_b, sc0, sw0)
        if sc0 == 0:
            scale = 1.0 + max(np.abs(lc).max(), np.abs(mc).max(), abs(zc.value))
            assert max(np.abs(lw - lc).max(), np.abs(mw - mc).max(), abs(zw.value - zc.value)) <= 2e-6 * scale, (trial, kind_a, kind_b, lc, lw)
