"""-m gpu : the device-side obstacle pipeline (rda_upload_scene / rda_step_scene) against the host
caller code it replaces (MPC.convert_rda_obstacle + RDA_solver._stage, mirrors of reference mpc.py:189-218,440-549
and rda_solver.py:483-526).  Integer / geometry staging work: the bar is BIT-EXACT slots."""
import ctypes as C

import numpy as np
import pytest

from rda_planner_amd import scenarios as sc
from rda_planner_amd._capi import dptr, iptr
from rda_planner_amd.mpc import MPC

pytestmark = pytest.mark.gpu


def _random_scene(rng, n, moving=0.0, circles=0.3, cw=0.5, E=4, around=(20.0, 20.0), spread=15.0):
    obs = []
    for _ in range(n):
        c = np.array(around) + rng.uniform(-spread, spread, 2)
        vel = rng.uniform(-1, 1, 2) if rng.random() < moving else (0.0, 0.0)
        if rng.random() < circles:
            obs.append(sc.circle(c[0], c[1], rng.uniform(0.3, 1.5), vel))
        else:
            k = int(rng.integers(3, E + 1))
            o = sc.regular_polygon(c[0], c[1], k, rng.uniform(0.5, 1.5), rng.uniform(-np.pi, np.pi), vel)
            if rng.random() < cw:                           # clockwise input: the reference reverses it
                o = o._replace(vertex=o.vertex[:, ::-1].copy())
            obs.append(o)
    return obs


def _host_slots(mpc, obs, state, order):
    """what the host path stages: (A [N][nt][E][2], b [N][nt][E], cone [N], nt)"""
    mpc.state = state
    lst = mpc.convert_rda_obstacle(obs, state, order)
    use, A, b, cone, per_t = mpc.rda._stage(lst)
    N = mpc.rda.max_obs_num
    assert use == N or use == 0
    return A, b, cone, (mpc.rda.T + 1 if per_t else 1)


def _device_slots(mpc, obs, state, order):
    api, h = mpc.rda._be.api, mpc.rda._be.handle
    n, kind, nvert, geom, vel = mpc.rda.flatten_scene(obs)
    T, N, E = mpc.rda.T, mpc.rda.max_obs_num, mpc.rda.max_edge_num
    bad = np.zeros(1, np.int32)
    rob = np.ascontiguousarray(np.asarray(state, float).ravel()[0:2])
    rc = api.upload_scene(h, n, iptr(np.ascontiguousarray(kind)), iptr(np.ascontiguousarray(nvert)), dptr(np.ascontiguousarray(geom)),
                          dptr(np.ascontiguousarray(vel)), dptr(rob), int(order), iptr(bad))
    assert rc == 0, rc
    A = np.zeros((N, T + 1, E, 2)); b = np.zeros((N, T + 1, E)); cone = np.zeros(N, np.int32); nt = np.zeros(1, np.int32)
    assert api.get_obstacles(h, dptr(A), dptr(b), iptr(cone), iptr(nt)) == 0
    nt = int(nt[0])
    return A.ravel()[: N * nt * E * 2].reshape(N, nt, E, 2), b.ravel()[: N * nt * E].reshape(N, nt, E), cone, nt, int(bad[0])


def _mpc(N, E=4, T=10):
    car = sc.rectangle_robot()
    path = sc.line_path([0, 20, 0], [60, 20, 0])
    return MPC(car, path, receding=T, max_edge_num=E, max_obs_num=N, iter_num=2)


@pytest.mark.parametrize("n,N,moving,order,E", [(7, 7, 0.0, True, 4), (5, 9, 0.0, True, 4), (30, 8, 0.0, True, 4), (12, 12, 0.5, True, 4),
                                                (40, 16, 0.3, True, 5), (9, 9, 0.4, False, 4), (3, 11, 1.0, False, 8), (200, 64, 0.2, True, 4)])
def test_slots_bit_identical(n, N, moving, order, E):
    rng = np.random.default_rng(1000 * n + N)
    mpc = _mpc(N, E)
    state = np.array([[18.0], [19.0], [0.3]])
    obs = _random_scene(rng, n, moving=moving, E=E)
    Ah, bh, ch, nth = _host_slots(mpc, list(obs), state, order)
    Ad, bd, cd, ntd, bad = _device_slots(mpc, obs, state, order)
    assert bad == 0
    assert np.array_equal(ch, cd)
    if nth == ntd:
        assert np.array_equal(Ah, Ad) and np.array_equal(bh, bd)
    else:                                                    # device replicates over t as soon as ANY obstacle of the scene moves
        assert nth == 1 and ntd == mpc.rda.T + 1
        assert np.array_equal(np.broadcast_to(Ah, Ad.shape), Ad) and np.array_equal(np.broadcast_to(bh, bd.shape), bd)


def test_ordering_ties_are_stable_and_nonconvex_is_reported():
    mpc = _mpc(4)
    state = np.array([[0.0], [0.0], [0.0]])
    # four identical-distance obstacles + two nearer ones at the end; order must be [4, 5, 0, 1]
    obs = [sc.circle(3.0, 0.0, 0.5), sc.circle(0.0, 3.0, 0.6), sc.circle(-3.0, 0.0, 0.7), sc.circle(0.0, -3.0, 0.8),
           sc.circle(1.0, 0.0, 0.2), sc.circle(0.0, 2.0, 0.3)]
    Ad, bd, cd, ntd, bad = _device_slots(mpc, obs, state, True)
    assert ntd == 1 and bad == 0
    assert np.array_equal(-bd[:, 0, 2], [0.2, 0.3, 0.5, 0.6])
    Ah, bh, ch, nth = _host_slots(mpc, list(obs), state, True)
    assert np.array_equal(bh, bd)
    # a dart (non-convex) polygon: the reference warns and keeps the input order
    dart = sc.Obstacle(None, None, np.array([[5.0, 7.0, 5.5, 7.0], [5.0, 6.0, 6.0, 4.0]]), "Rpositive", np.zeros((2, 1)))
    obs2 = [dart, sc.circle(9, 9, 1.0)]
    Ad, bd, cd, ntd, bad = _device_slots(mpc, obs2, state, False)
    Ah, bh, ch, nth = _host_slots(mpc, list(obs2), state, False)
    assert bad == 1 and np.array_equal(Ah, Ad) and np.array_equal(bh, bd) and np.array_equal(ch, cd)


def test_nonconvex_warning_travels_with_the_step_result(capsys):
    """the default (device-staged, no extra synchronisation) control path prints the reference's convexity warning (mpc.py:524)"""
    mpc = _mpc(4)
    state = np.array([[0.0], [20.0], [0.0]])
    dart = sc.Obstacle(None, None, np.array([[15.0, 17.0, 15.5, 17.0], [25.0, 26.0, 26.0, 24.0]]), "Rpositive", np.zeros((2, 1)))
    mpc.control(state, 4.0, [dart, sc.circle(19, 29, 1.0)])
    assert "not convex" in capsys.readouterr().out
    mpc.control(state, 4.0, [sc.circle(19, 29, 1.0)])
    assert "not convex" not in capsys.readouterr().out


def test_empty_scene_skips_dual_side():
    mpc = _mpc(5)
    state = np.array([[0.0], [20.0], [0.0]])
    u, info = mpc.control(state, 4.0, [])
    assert np.isfinite(u).all() and info["iters"] >= 1


@pytest.mark.parametrize("moving", [0.0, 0.5])
def test_closed_loop_identical_to_host_staging(moving):
    """the whole MPC step with device-side conversion == with the host conversion, bit for bit"""
    rng = np.random.default_rng(7)
    car = sc.rectangle_robot()
    path = sc.line_path([0, 20, 0], [60, 20, 0])
    obs = _random_scene(rng, 25, moving=moving, around=(25.0, 20.0), spread=12.0)
    runs = []
    for dev in (True, False):
        mpc = MPC(car, [p.copy() for p in path], receding=15, max_edge_num=4, max_obs_num=12, iter_num=3, device_obstacles=dev)
        assert mpc.rda.has_scene
        state = np.array([[0.0], [20.0], [0.0]])
        us = []
        for k in range(25):
            cur = [o if o.cone_type != "Rpositive" or np.linalg.norm(o.velocity) <= 0.01 else
                   o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in obs]
            cur = [o if o.cone_type != "norm2" or np.linalg.norm(o.velocity) <= 0.01 else
                   o._replace(center=o.center + o.velocity * (0.1 * k)) for o in cur]
            u, info = mpc.control(state, 4.0, cur)
            us.append(u.ravel().copy())
            state = sc.kinematic_step(state, u, car, 0.1)
        runs.append(np.array(us))
    assert np.array_equal(runs[0], runs[1])
