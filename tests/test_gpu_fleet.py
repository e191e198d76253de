"""GPU tests of the fleet path (include/rda_hip.h `rda_fleet_*`, rda_planner_amd/fleet.py): B independent egos advanced by
one set of launches per ADMM iteration must give, ego by ego, exactly what the members give when stepped on their own
(the same device code runs on the same data; only the grid has one more dimension) - BASELINE config C5."""
import ctypes as C

import numpy as np
import pytest

import helpers as hp
from rda_planner_amd import scenarios as sc
from rda_planner_amd._capi import Info, dptr, iptr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from rda_planner_amd._lib import hip_api
    return hip_api()


def _members(B, T, N, iter_num=3, per_ego_robot=False):
    """B twins (solo, fleet member) of solvers with different kinematics / weights / obstacle sets"""
    from rda_planner_amd.rda_solver import RDA_solver
    from rda_planner_amd.mpc import MPC
    solo, memb, staged = [], [], []
    for i in range(B):
        dyn = ["acker", "diff", "omni"][i % 3]
        car_t = sc.rectangle_robot(dynamics=dyn, wheelbase=3.0 if dyn == "acker" else 0,
                                   length=4.6 + (0.2 * i if per_ego_robot else 0))
        kw = dict(iter_num=iter_num, time_print=False, ro1=[200.0, 100.0, 300.0][i % 3], slack_gain=8.0 + i)
        solo.append(RDA_solver(T, car_t, 4, N, **kw))
        memb.append(RDA_solver(T, car_t, 4, N, **kw))
        n_obs = [N, N // 2, 0, N + 3][i % 4]                     # full, padded (Q3), empty (Q9), truncated
        obstacles = sc.scene_polygons(n_obs, lo=(5, -8), hi=(25, 8), seed=11 + i, moving=(i % 2 == 1)) if n_obs else []
        conv = MPC.__new__(MPC); conv.receding = T; conv.dt = 0.1; conv.state = np.zeros((3, 1))
        staged.append(MPC.convert_rda_obstacle(conv, obstacles, np.zeros((3, 1)), True))
    return solo, memb, staged


def _inputs(rng, B, T, N, K):
    noms, nomu, refs = np.zeros((K, B, 3, T + 1)), np.zeros((K, B, 2, T)), np.zeros((K, B, 3, T + 1))
    for k in range(K):
        for i in range(B):
            si = hp.su_inputs(rng, hp.make_cfg(T=T, N=N, dynamics=i % 3))
            noms[k, i], nomu[k, i], refs[k, i] = si["nom_s"], si["nom_u"].reshape(2, T), si["ref"]
    return noms, nomu, refs


def _fleet(hip, memb):
    B = len(memb)
    arr = (C.c_void_p * B)(*[m._be.handle for m in memb])
    F = C.c_void_p()
    assert hip.fleet_create(arr, B, C.byref(F)) == 0
    assert hip.fleet_size(F) == B
    return F


def test_fleet_step_equals_member_steps(hip):
    """7 egos (three kinematics, different weights, robots, obstacle counts incl. none / padded / truncated, static and
    moving), 5 steps: controls, states, iteration counts, residuals and the whole dual state are bit-identical"""
    rng = np.random.default_rng(5)
    B, T, N, K = 7, 10, 12, 5
    solo, memb, staged = _members(B, T, N, per_ego_robot=True)
    noms, nomu, refs = _inputs(rng, B, T, N, K)
    speed = np.linspace(2.0, 5.0, B)
    F = _fleet(hip, memb)
    for k in range(K):
        want = []
        for i in range(B):
            u, info = solo[i].iterative_solve(noms[k, i], nomu[k, i], [refs[k, i][:, j:j + 1] for j in range(T + 1)], speed[i], list(staged[i]))
            want.append((u, info))
            memb[i].upload_obstacles(list(staged[i]))
        ou, os_ = np.zeros((B, 2, T)), np.zeros((B, 3, T + 1))
        infos = (Info * B)()
        assert hip.fleet_step(F, dptr(noms[k]), dptr(nomu[k]), dptr(refs[k]), dptr(speed), dptr(ou), dptr(os_), infos) == 0
        for i in range(B):
            u, info = want[i]
            assert np.array_equal(ou[i], u), (k, i, np.abs(ou[i] - u).max())
            assert np.array_equal(os_[i], np.hstack(info["opt_state_list"])), (k, i)
            assert infos[i].iters == info["iters"] and infos[i].su_status == info["status"], (k, i)
            assert infos[i].resi_dual == info["resi_dual"] and infos[i].resi_pri == info["resi_pri"], (k, i)
            assert infos[i].su_ipm_iters == info["su_ipm_iters"], (k, i)
    for i in range(B):
        a, b = solo[i].get_state(), memb[i].get_state()
        for key in a:
            assert np.array_equal(a[key], b[key]), (i, key)
    hip.fleet_destroy(F)
    # the members are still ordinary solvers afterwards
    u, info = memb[0].iterative_solve(noms[0, 0], nomu[0, 0], [refs[0, 0][:, j:j + 1] for j in range(T + 1)], 3.0, list(staged[0]))
    assert np.isfinite(u).all()


def test_fleet_trace_replay_equals_member_replay(hip):
    """the device-resident replay that bench.py times: rda_fleet_enqueue_range == rda_enqueue_range on every member"""
    rng = np.random.default_rng(8)
    B, T, N, K = 5, 20, 24, 6
    solo, memb, staged = _members(B, T, N, iter_num=4)
    noms, nomu, refs = _inputs(rng, B, T, N, K)
    speed = np.full(K, 4.0)
    for i in range(B):
        n, A, b, cone, per_t = solo[i]._stage(list(staged[i]))
        for s in (solo[i], memb[i]):
            h = s._be.handle
            assert hip.lib.rda_upload_obstacles(h, n, dptr(A), dptr(b), iptr(cone), per_t) == 0
            tr = [np.ascontiguousarray(x[:, i]) for x in (noms, nomu, refs)]
            assert hip.lib.rda_upload_trace(h, K, dptr(tr[0]), dptr(tr[1]), dptr(tr[2]), dptr(speed)) == 0
        assert hip.lib.rda_enqueue_range(solo[i]._be.handle, 0, K) == 0
    F = _fleet(hip, memb)
    assert hip.fleet_enqueue_range(F, 0, 2) == 0
    assert hip.fleet_enqueue_range(F, 2, K) == 0
    assert hip.fleet_enqueue_range(F, 0, K + 1) != 0              # beyond the uploaded trace
    assert hip.fleet_sync(F) == 0
    for i in range(B):
        for k in range(K):
            ua, sa, ia = np.zeros((2, T)), np.zeros((3, T + 1)), Info()
            ub, sb, ib = np.zeros((2, T)), np.zeros((3, T + 1)), Info()
            assert hip.lib.rda_fetch_result(solo[i]._be.handle, k, dptr(ua), dptr(sa), C.byref(ia)) == 0
            assert hip.lib.rda_fetch_result(memb[i]._be.handle, k, dptr(ub), dptr(sb), C.byref(ib)) == 0
            assert np.array_equal(ua, ub) and np.array_equal(sa, sb), (i, k)
            assert (ia.iters, ia.su_status, ia.resi_dual, ia.resi_pri) == (ib.iters, ib.su_status, ib.resi_dual, ib.resi_pri)
    hip.fleet_destroy(F)


def test_fleet_rejects_mismatched_members(hip):
    from rda_planner_amd.rda_solver import RDA_solver
    car_t = sc.rectangle_robot()
    a = RDA_solver(10, car_t, 4, 8, iter_num=2, time_print=False)
    for other in (RDA_solver(12, car_t, 4, 8, iter_num=2, time_print=False), RDA_solver(10, car_t, 5, 8, iter_num=2, time_print=False),
                  RDA_solver(10, car_t, 4, 9, iter_num=2, time_print=False), RDA_solver(10, car_t, 4, 8, iter_num=3, time_print=False)):
        arr = (C.c_void_p * 2)(a._be.handle, other._be.handle)
        F = C.c_void_p()
        assert hip.fleet_create(arr, 2, C.byref(F)) != 0
    F = C.c_void_p()
    assert hip.fleet_create(None, 0, C.byref(F)) != 0


@pytest.mark.parametrize("mixed", [True, False])
def test_fleet_control_equals_member_control(mixed):
    """`Fleet.control` over 6 closed loops (different paths, kinematics, static / moving scenes, device and host obstacle
    staging) == six `MPC.control` loops, bit for bit, until every member has arrived.  mixed: one member converts its
    obstacles on the host, which sends every member through the per-member staging; otherwise all scenes are flattened in
    one pass and staged by one call (rda_fleet_upload_scenes)"""
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd.fleet import Fleet
    B = 6
    solo, memb, cars, scenes, states = [], [], [], [], []
    for i in range(B):
        dyn = ["acker", "diff", "omni"][i % 3]
        car_t = sc.rectangle_robot(dynamics=dyn, wheelbase=3.0 if dyn == "acker" else 0)
        y = 20.0 + 3 * i
        path = sc.line_path([4, y, 0], [24 + 2 * i, y, 0], 0.1)
        clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
        scene = sc.scene_polygons([12, 6][i % 2], lo=(6, y - 10), hi=(30, y + 10), seed=40 + i, keep_clear=clear, clear_radius=3.0, moving=(i % 2 == 0))
        scene.append(sc.circle(15.0, y + 4.0, 0.8, (0.0, -0.2)))
        kw = dict(receding=10, iter_num=3, max_edge_num=4, max_obs_num=10, device_obstacles=(i != 4 or not mixed))
        solo.append(MPC(car_t, [p.copy() for p in path], **kw))
        memb.append(MPC(car_t, [p.copy() for p in path], **kw))
        cars.append(car_t); scenes.append(scene)
        st = path[0].copy().reshape(3, 1)
        if dyn == "omni":
            st[2, 0] = 0.0
        states.append(st)
    fleet = Fleet(memb)
    assert len(fleet) == B
    arrived = [False] * B
    for k in range(90):
        cur = [[o if not o.velocity.any() else (o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) if o.cone_type == "Rpositive"
                                                else o._replace(center=o.center + o.velocity * (0.1 * k))) for o in scenes[i]] for i in range(B)]
        res = fleet.control([s.copy() for s in states], [3.0 + 0.2 * i for i in range(B)], [list(c) for c in cur])
        for i in range(B):
            u, info = solo[i].control(states[i].copy(), 3.0 + 0.2 * i, list(cur[i]))
            uf, inf = res[i]
            assert np.array_equal(u, uf), (k, i, np.abs(u - uf).max())
            assert info["iters"] == inf["iters"] and info["arrive"] == inf["arrive"] and info["resi_dual"] == inf["resi_dual"]
            assert np.array_equal(np.hstack(info["opt_state_list"]), np.hstack(inf["opt_state_list"]))
            arrived[i] = arrived[i] or info["arrive"]
            states[i] = sc.kinematic_step(states[i], u, cars[i], 0.1)
        if all(arrived):
            break
    assert sum(arrived) >= 3
    assert (fleet.batched_ticks == 0) if mixed else (fleet.batched_ticks > 0)
    fleet.close()


@pytest.mark.parametrize("T,E,robot_k", [(40, 4, 4), (12, 8, 8), (33, 6, 3), (64, 5, 5)])
def test_fleet_control_equals_member_control_other_shapes(T, E, robot_k):
    """the same bit-for-bit statement on shapes the examples do not use: horizons without a compile-time su instantiation (generic T), obstacle
    polygons with up to E vertices, robot bodies with 3 / 5 / 8 edges (E + R + 1 > 16: the fleet form of the one-row-per-wave LamMuZ kernel)"""
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd.fleet import Fleet
    B = 5
    rng = np.random.default_rng(100 * T + E)
    solo, memb, cars, scenes, states = [], [], [], [], []
    for i in range(B):
        dyn = ["acker", "diff", "omni"][i % 3]
        car_t = sc.rectangle_robot(dynamics=dyn, wheelbase=3.0 if dyn == "acker" else 0)
        if robot_k != 4:
            ang = 2 * np.pi * (np.arange(robot_k) + 0.5) / robot_k
            Gk, hk = sc.polygon_halfspaces(np.vstack(((1.5 if dyn == "acker" else 0.0) + 2.3 * np.cos(ang), 0.9 * np.sin(ang))))
            car_t = car_t._replace(G=Gk, h=hk)
        y = 20.0 + 3 * i
        path = sc.line_path([4, y, 0], [30, y, 0], 0.1)
        clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
        scene = []
        while len(scene) < 14:
            c = rng.uniform((6, y - 10), (34, y + 10))
            if np.min(np.linalg.norm(clear - c, axis=1)) < 3.0:
                continue
            vel = rng.uniform(-0.4, 0.4, 2) if i % 2 == 0 else (0.0, 0.0)
            scene.append(sc.regular_polygon(c[0], c[1], int(rng.integers(3, E + 1)), rng.uniform(0.5, 1.0), rng.uniform(-np.pi, np.pi), vel))
        scene.append(sc.circle(15.0, y + 4.0, 0.8, (0.0, -0.2)))
        kw = dict(receding=T, iter_num=3, max_edge_num=E, max_obs_num=12)
        solo.append(MPC(car_t, [p.copy() for p in path], **kw))
        memb.append(MPC(car_t, [p.copy() for p in path], **kw))
        cars.append(car_t); scenes.append(scene)
        st = path[0].copy().reshape(3, 1)
        if dyn == "omni":
            st[2, 0] = 0.0
        states.append(st)
    fleet = Fleet(memb)
    for k in range(25):
        cur = [[o if not o.velocity.any() else (o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) if o.cone_type == "Rpositive"
                                                else o._replace(center=o.center + o.velocity * (0.1 * k))) for o in scenes[i]] for i in range(B)]
        res = fleet.control([s.copy() for s in states], 3.5, [list(c) for c in cur])
        for i in range(B):
            u, info = solo[i].control(states[i].copy(), 3.5, list(cur[i]))
            uf, inf = res[i]
            assert info["status"] == 0 and inf["status"] == 0, (k, i)
            assert np.array_equal(u, uf), (k, i, np.abs(u - uf).max())
            assert info["iters"] == inf["iters"] and info["resi_dual"] == inf["resi_dual"] and info["resi_pri"] == inf["resi_pri"]
            assert np.array_equal(np.hstack(info["opt_state_list"]), np.hstack(inf["opt_state_list"]))
            states[i] = sc.kinematic_step(states[i], u, cars[i], 0.1)
    assert fleet.batched_ticks > 0
    fleet.close()


@pytest.mark.parametrize("resort", [2, 1], ids=["fleet-resort", "member-resort"])
def test_c_fleet_closed_loop_equals_solo_closed_loops(hip, resort):
    """BASELINE config C5 as a closed loop through the C-ABI with the caller in C (tools/closed_loop_host.c closed_loop_fleet_run, what bench.py's
    `c_abi_closed_loop` fleet leg times): per fleet tick the members' scenes re-sorted (rda_fleet_scene_resort: one launch set; or rda_scene_resort member by
    member) + ONE rda_fleet_step_tracked + the members' kinematics.  Every
    member - own scene (two of them circle-heavy: norm2 rows go through the enumeration in the fleet kernels and through `warm_circle` in the solo
    kernel), own kinematics-independent path offset - against its SOLO closed loop in the same protocol (closed_loop_run: rda_tracked_begin +
    rda_scene_resort + rda_tracked_finish): bit for bit, controls and iteration counts."""
    import os
    import sys
    from rda_planner_amd.rda_solver import RDA_solver
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import closed_loop_host as clh
    host = clh.Host(hip.lib)
    B, T, N, steps = 5, 10, 16, 40
    car_t = sc.rectangle_robot(dynamics="acker")

    def make(e):
        y = 20.0 + 2.5 * e
        path = sc.line_path([4, y, 0], [44, y, 0], 0.1)
        clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
        scene = sc.scene_polygons(14 if e % 2 == 0 else 5, lo=(6, y - 6), hi=(44, y + 6), seed=70 + e, keep_clear=clear, clear_radius=2.3)
        rng = np.random.default_rng(700 + e)
        while len(scene) < 20:                       # circles (more of them in the odd members)
            c = rng.uniform((6, y - 6), (44, y + 6))
            if np.min(np.linalg.norm(clear - c, axis=1)) > 2.6:
                scene.append(sc.circle(c[0], c[1], rng.uniform(0.4, 1.0)))
        sv = RDA_solver(T, car_t, 4, N, iter_num=3, time_print=False)
        n_sc, kind, nvert, geom, vel = sv.flatten_scene(list(scene))
        kind, nvert = np.ascontiguousarray(kind, np.int32), np.ascontiguousarray(nvert, np.int32)
        geom, vel = np.ascontiguousarray(geom, float), np.ascontiguousarray(vel, float)
        P = np.ascontiguousarray(np.hstack(path)[0:3, :].T, dtype=float)
        st = np.ascontiguousarray(path[0], float).ravel()[0:3].copy()
        assert hip.upload_path(sv._be.handle, int(P.shape[0]), dptr(P)) == 0
        assert hip.upload_scene(sv._be.handle, int(n_sc), iptr(kind), iptr(nvert), dptr(geom), dptr(vel), dptr(st), 1, None) == 0
        return sv, st, int(P.shape[0])
    memb = [make(e) for e in range(B)]
    states = np.array([m[1] for m in memb])
    plen = np.array([m[2] for m in memb], np.int32)
    arr = (C.c_void_p * B)(*[m[0]._be.handle for m in memb])
    F = C.c_void_p()
    assert hip.fleet_create(arr, B, C.byref(F)) == 0
    cur, nom_u0 = np.zeros(B, np.int32), np.zeros((B, 2, T))
    u_log, t_log = np.zeros((steps, B, 2)), np.zeros(steps)
    it_log, ipm_log = np.zeros((steps, B), np.int32), np.zeros((steps, B), np.int32)
    rc = host.fleet_run(C.byref(host.fleet_api), F, arr, B, T, 0, 3.0, 0.1, 4.0, 0.1, 10, iptr(plen), resort, 0, steps, dptr(nom_u0), dptr(states), iptr(cur),
                        dptr(u_log), dptr(t_log), iptr(it_log), iptr(ipm_log))
    assert rc == 0, rc
    hip.fleet_destroy(F)
    assert np.abs(u_log[:, :, 0]).min() > 0.5 and len({tuple(np.round(u_log[-1, e], 4)) for e in range(B)}) >= 3      # the members do different things
    for e in range(B):
        sv, st, pl = make(e)
        scn = host.Scene(0, 0, 1, 0, None, None, None, None, None)
        cur_c = C.c_int32(0)
        ul, tl, il = np.zeros((steps, 2)), np.zeros(steps), np.zeros(steps, np.int32)
        rc = host.run(C.byref(host.api), sv._be.handle, C.byref(scn), T, 0, 3.0, 0.1, 4.0, 0.1, 10, pl, 0, steps, dptr(np.zeros((2, T))), dptr(st),
                      C.byref(cur_c), dptr(ul), dptr(tl), iptr(il), None, None)
        assert rc == 0, rc
        assert np.array_equal(il, it_log[:, e]), (e, il, it_log[:, e])
        assert np.array_equal(ul, u_log[:, e, :]), (e, float(np.abs(ul - u_log[:, e, :]).max()))
        assert np.allclose(st, states[e], rtol=0, atol=0)


def test_fleets_ticked_by_their_own_host_threads_equal_one_fleet(hip):
    """What bench.py's `c_abi_closed_loop.fleets_4_host_threads` times: the members split over SEVERAL fleets, each fleet's closed loop run by its own host thread
    (closed_loop_fleet_run, no interpreter lock inside) - nothing couples two fleets, the library keeps no state outside its handles, so the concurrent loops give
    the controls of ONE fleet of all members (which the test above pins to the solo loops), bit for bit."""
    import os
    import sys
    import threading
    from rda_planner_amd.rda_solver import RDA_solver
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import closed_loop_host as clh
    host = clh.Host(hip.lib)
    B, T, N, steps = 6, 10, 16, 30
    car_t = sc.rectangle_robot(dynamics="acker")

    def make(e):
        y = 20.0 + 2.5 * e
        path = sc.line_path([4, y, 0], [44, y, 0], 0.1)
        clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
        scene = sc.scene_polygons(12, lo=(6, y - 6), hi=(44, y + 6), seed=170 + e, keep_clear=clear, clear_radius=2.3)
        sv = RDA_solver(T, car_t, 4, N, iter_num=3, time_print=False)
        n_sc, kind, nvert, geom, vel = sv.flatten_scene(list(scene))
        kind, nvert = np.ascontiguousarray(kind, np.int32), np.ascontiguousarray(nvert, np.int32)
        geom, vel = np.ascontiguousarray(geom, float), np.ascontiguousarray(vel, float)
        P = np.ascontiguousarray(np.hstack(path)[0:3, :].T, dtype=float)
        st = np.ascontiguousarray(path[0], float).ravel()[0:3].copy()
        assert hip.upload_path(sv._be.handle, int(P.shape[0]), dptr(P)) == 0
        assert hip.upload_scene(sv._be.handle, int(n_sc), iptr(kind), iptr(nvert), dptr(geom), dptr(vel), dptr(st), 1, None) == 0
        return sv, st, int(P.shape[0])

    def loops(split):
        memb = [make(e) for e in range(B)]
        groups = []
        for lo, hi in split:
            n = hi - lo
            arr = (C.c_void_p * n)(*[m[0]._be.handle for m in memb[lo:hi]])
            F = C.c_void_p()
            assert hip.fleet_create(arr, n, C.byref(F)) == 0
            groups.append(dict(n=n, arr=arr, F=F, states=np.array([m[1] for m in memb[lo:hi]]), plen=np.array([m[2] for m in memb[lo:hi]], np.int32),
                               cur=np.zeros(n, np.int32), nom=np.zeros((n, 2, T)), u=np.zeros((steps, n, 2)), t=np.zeros(steps),
                               it=np.zeros((steps, n), np.int32), ipm=np.zeros((steps, n), np.int32), rc=-99))

        def run(q):
            q["rc"] = host.fleet_run(C.byref(host.fleet_api), q["F"], q["arr"], q["n"], T, 0, 3.0, 0.1, 4.0, 0.1, 10, iptr(q["plen"]), 2, 0, steps, dptr(q["nom"]),
                                     dptr(q["states"]), iptr(q["cur"]), dptr(q["u"]), dptr(q["t"]), iptr(q["it"]), iptr(q["ipm"]))
        th = [threading.Thread(target=run, args=(q,)) for q in groups]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        for q in groups:
            assert q["rc"] == 0, q["rc"]
            hip.fleet_destroy(q["F"])
        return np.concatenate([q["u"] for q in groups], axis=1), np.concatenate([q["it"] for q in groups], axis=1), np.concatenate([q["states"] for q in groups])
    u1, it1, s1 = loops([(0, B)])
    u3, it3, s3 = loops([(0, 2), (2, 4), (4, 6)])
    assert np.abs(u1[:, :, 0]).min() > 0.5
    assert np.array_equal(it1, it3)
    assert np.array_equal(u1, u3), float(np.abs(u1 - u3).max())
    assert np.array_equal(s1, s3)
