"""CPU tests of the host-side mirror classes (rda_planner_amd.RDA_solver / MPC) running on the CPU
oracle backend: API surface, reference quirks (SURVEY.md 8a Q-list), closed-loop behaviour, golden run."""
import inspect
import json
import os

import numpy as np
import pytest

from oracle.oracle_backend import oracle_backend
from rda_planner_amd import scenarios as sc
from rda_planner_amd.mpc import MPC, rdaobs
from rda_planner_amd.rda_solver import RDA_solver

GOLD = os.path.join(os.path.dirname(__file__), "golden", "path_track_diff_golden.json")


def _solver(**kw):
    car_t = sc.rectangle_robot(dynamics=kw.pop("dynamics", "acker"))
    args = dict(receding=6, car_tuple=car_t, max_edge_num=4, max_obs_num=3, iter_num=2, time_print=False, _backend=oracle_backend)
    args.update(kw)
    return RDA_solver(**args), car_t


def _obs(n, cx=8.0):
    out = []
    for i in range(n):
        A, b = sc.polygon_halfspaces(sc.box(cx + 3 * i, 2.5 + i, 2, 1, 0.3 * i).vertex)
        out.append(rdaobs(A, b, "Rpositive", None, None))
    return out


def _inputs(T):
    nom_u = np.vstack([np.full(T, 2.0), np.zeros(T)])
    nom_s = np.zeros((3, T + 1))
    for t in range(T):
        nom_s[:, t + 1] = nom_s[:, t] + 0.1 * np.array([2.0, 0, 0])
    ref = [np.array([[0.4 * t], [0.0], [0.0]]) for t in range(T + 1)]
    return nom_s, nom_u, ref


def test_signatures_match_reference():
    """argument names / defaults of the reference (rda_solver.py:18-22, mpc.py:67-84, :127)"""
    sig = inspect.signature(RDA_solver.__init__)
    assert list(sig.parameters)[1:11] == ["receding", "car_tuple", "max_edge_num", "max_obs_num", "iter_num", "step_time",
                                          "iter_threshold", "process_num", "accelerated", "time_print"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["max_edge_num"], d["max_obs_num"], d["iter_num"], d["step_time"], d["iter_threshold"], d["process_num"]) == (5, 5, 2, 0.1, 0.2, 4)
    sig = inspect.signature(MPC.__init__)
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["receding"], d["sample_time"], d["iter_num"], d["max_edge_num"], d["max_obs_num"], d["goal_index_threshold"]) == (10, 0.1, 4, 5, 5, 1)
    assert list(inspect.signature(MPC.control).parameters)[1:4] == ["state", "ref_speed", "obstacle_list"]
    for name in ("iterative_solve", "assign_adjust_parameter", "get_adjust_parameter", "reset"):
        assert hasattr(RDA_solver, name)
    for name in ("control", "update_ref_path", "update_parameter", "get_adjust_parameters", "reset", "no_ref_path",
                 "convert_rda_obstacle", "gen_inequal_global", "pre_process"):
        assert hasattr(MPC, name)


def test_info_keys_and_shapes():
    sol, _ = _solver()
    nom_s, nom_u, ref = _inputs(6)
    u, info = sol.iterative_solve(nom_s, nom_u, ref, 4.0, _obs(2))
    assert u.shape == (2, 6)
    for k in ("ref_traj_list", "opt_state_list", "iteration_time", "resi_dual", "resi_pri"):   # rda_solver.py:603-608
        assert k in info
    assert len(info["opt_state_list"]) == 7 and info["opt_state_list"][0].shape == (3, 1)
    assert 1 <= info["iters"] <= 2


def test_q3_padding_duplicates_last_and_mutates_callers_list():
    sol, _ = _solver()
    nom_s, nom_u, ref = _inputs(6)
    lst = _obs(2)
    sol.iterative_solve(nom_s, nom_u, ref, 4.0, lst)
    assert len(lst) == 3 and lst[2] is lst[1]                       # rda_solver.py:488-490
    st = sol.get_state()
    assert np.allclose(st["lam"][2], st["lam"][1]) and np.allclose(st["mu"][2], st["mu"][1])


def test_truncation_uses_first_max_obs_num():
    sol, _ = _solver()
    sol2, _ = _solver()
    nom_s, nom_u, ref = _inputs(6)
    five = _obs(5)
    u1, _ = sol.iterative_solve(nom_s, nom_u, ref, 4.0, list(five))
    u2, _ = sol2.iterative_solve(nom_s, nom_u, ref, 4.0, list(five[:3]))
    assert np.array_equal(u1, u2)


def test_zero_obstacles_q9_and_residuals():
    sol, _ = _solver()
    nom_s, nom_u, ref = _inputs(6)
    sol.iterative_solve(nom_s, nom_u, ref, 4.0, _obs(3))
    before = sol.get_state()
    u, info = sol.iterative_solve(nom_s, nom_u, ref, 4.0, [])
    after = sol.get_state()
    assert info["resi_dual"] == 0 and info["resi_pri"] == 0 and info["iters"] == 1      # rda_solver.py:614,625,594
    assert np.all(after["a_lam"][2] == 0) and np.all(after["b_lam"][2] == 0)            # only slot N-1 (Q9, :564-568)
    assert np.array_equal(after["a_lam"][:2], before["a_lam"][:2])
    for k in ("lam", "mu", "z", "xi", "zeta"):                                          # duals untouched (:625)
        assert np.array_equal(after[k], before[k])


def test_reset_q6_clears_products_not_duals():
    sol, _ = _solver()
    nom_s, nom_u, ref = _inputs(6)
    sol.iterative_solve(nom_s, nom_u, ref, 4.0, _obs(3))
    before = sol.get_state()
    sol.reset()
    after = sol.get_state()
    assert np.all(after["a_lam"] == 0) and np.all(after["b_lam"] == 0)                  # rda_solver.py:1067-1068
    for k in ("lam", "mu", "z", "xi", "zeta", "dis"):
        assert np.array_equal(after[k], before[k])


def test_adjust_parameters_roundtrip_and_effect():
    sol, _ = _solver(ro1=300, slack_gain=9)
    p = sol.get_adjust_parameter()
    assert p == {"slack_gain": 9, "max_sd": 1.0, "min_sd": 0.1, "ro1": 300, "ro2": 1, "ws": 1, "wu": 1}
    nom_s, nom_u, ref = _inputs(6)
    u1, _ = sol.iterative_solve(nom_s, nom_u, ref, 4.0, _obs(3, cx=5.0))
    sol.assign_adjust_parameter(max_sd=0.3, min_sd=0.2)
    assert sol.get_adjust_parameter()["max_sd"] == 0.3
    sol.iterative_solve(nom_s, nom_u, ref, 4.0, _obs(3, cx=5.0))
    d = sol.get_state()["dis"]
    assert (d <= 0.3 + 1e-9).all() and (d >= 0.2 - 1e-9).all()                          # rda_solver.py:944-945


def test_dynamic_obstacle_lists_and_circles():
    car_t = sc.rectangle_robot(dynamics="diff", wheelbase=0)
    mpc = MPC(car_t, sc.line_path([0, 0, 0], [20, 0, 0], 0.2), receding=6, max_edge_num=4, max_obs_num=3, iter_num=2, _backend=oracle_backend)
    obs = [sc.circle(8, 2.5, 1.0, velocity=(0.5, 0)), sc.box(12, -3, 2, 1, 0.4, velocity=(0, 0.3)), sc.circle(15, 3, 0.5)]
    rl = mpc.convert_rda_obstacle(obs, np.zeros((3, 1)), True)
    assert isinstance(rl[0].A, list) and len(rl[0].A) == 7                              # mpc.py:447-456
    u, info = mpc.control(np.zeros((3, 1)), 3.0, obs)
    assert u.shape == (2, 1) and np.isfinite(u).all()


def test_too_many_edges_raises():
    sol, _ = _solver()
    nom_s, nom_u, ref = _inputs(6)
    A, b = sc.polygon_halfspaces(sc.regular_polygon(9, 3, 6, 1.0, 0).vertex)
    with pytest.raises(ValueError):
        sol.iterative_solve(nom_s, nom_u, ref, 4.0, [rdaobs(A, b, "Rpositive", None, None)])


def test_geometry_conversion_matches_reference_rule():
    car_t = sc.rectangle_robot()
    mpc = MPC(car_t, sc.line_path([0, 0, 0], [5, 0, 0]), _backend=oracle_backend, max_obs_num=1)
    V = np.array([[31, 33, 33, 31], [24, 24, 28, 28.0]])
    A, b = mpc.gen_inequal_global(V)
    assert np.allclose(A, [[0, -2], [4, 0], [0, 2], [-4, 0]]) and np.allclose(b.ravel(), [-48, 132, 56, -124])   # mpc.py:496-508
    A2, b2 = mpc.gen_inequal_global(V[:, ::-1])                       # CW input is reversed first (mpc.py:486-487)
    assert {tuple(np.r_[r, c]) for r, c in zip(A2, b2.ravel())} == {tuple(np.r_[r, c]) for r, c in zip(A, b.ravel())}
    Ac, bc = mpc.convert_inequal_circle(np.array([[2.0], [3.0]]), 1.5)
    assert np.array_equal(Ac, [[1, 0], [0, 1], [0, 0]]) and np.allclose(bc.ravel(), [2, 3, -1.5])                 # mpc.py:444-446


def test_closed_loop_path_track_no_collision_and_golden():
    """BASELINE config C1 (example/path_track/path_track_diff.py:21-23) - tracks the path, never touches an
    obstacle, reaches the goal; the first 40 controls are pinned by a committed golden run"""
    car_d = sc.rectangle_robot(wheelbase=0, dynamics="diff")
    ref = sc.path_track_ref()
    obs = sc.scene_path_track()
    mpc = MPC(car_d, [r.copy() for r in ref], receding=10, sample_time=0.1, iter_num=2, obstacle_order=True, ro1=300,
              max_edge_num=4, max_obs_num=11, slack_gain=8, _backend=oracle_backend)
    gold = json.load(open(GOLD))
    state = np.array([[10.0], [42.0], [1.57]])      # robot state of path_track_diff.yaml:13
    minc, arrived = np.inf, False
    for i in range(500):
        u, info = mpc.control(state, 4, list(obs))
        if i < 40:
            assert np.abs(u.ravel() - np.array(gold["u"][i])).max() < 1e-6, i
        state = sc.kinematic_step(state, u, car_d, 0.1)
        minc = min(minc, sc.clearance(car_d, state, obs))
        if info["arrive"]:
            arrived = True
            break
    assert arrived and minc > 0.05, (arrived, minc)


def test_flatten_scene_layout():
    """raw obstacle objects -> the flat arrays rda_upload_scene takes (include/rda_hip.h); unknown cone types are
    skipped like the reference's convert_rda_obstacle does (mpc.py:196-203); too many vertices -> host fallback"""
    car_t = sc.rectangle_robot()
    path = sc.line_path([0, 0, 0], [10, 0, 0])
    mpc = MPC(car_t, path, receding=5, max_edge_num=4, max_obs_num=3, _backend=oracle_backend)
    assert not mpc.rda.has_scene                                   # the CPU oracle has no device pipeline
    tri = sc.regular_polygon(3.0, 1.0, 3, 0.5, 0.2, velocity=(0.5, -0.25))
    quad = sc.box(6.0, -1.0, 2.0, 1.0, 0.3)
    cir = sc.circle(8.0, 2.0, 0.7, velocity=(0.0, 0.1))
    odd = sc.Obstacle(None, None, None, "something_else", np.zeros((2, 1)))
    n, kind, nvert, geom, vel = mpc.rda.flatten_scene([tri, odd, cir, quad])
    assert n == 3 and kind.tolist() == [0, 1, 0] and nvert.tolist() == [3, 0, 4]
    assert np.array_equal(geom[0, :3], tri.vertex.T) and np.array_equal(geom[0, 3], [0, 0])
    assert np.array_equal(geom[1, 0], cir.center.ravel()) and geom[1, 1, 0] == 0.7
    assert np.array_equal(geom[2], quad.vertex.T)
    assert np.array_equal(vel, [[0.5, -0.25], [0.0, 0.1], [0.0, 0.0]])
    penta = sc.regular_polygon(0, 0, 5, 1.0, 0.0)
    assert mpc.rda.flatten_scene([penta]) is None                  # 5 vertices > max_edge_num = 4
    assert mpc.rda.flatten_scene([])[0] == 0
    # and the MPC runs through the host conversion with this backend
    u, info = mpc.control(np.zeros((3, 1)), 2.0, [tri, cir, quad])
    assert np.isfinite(u).all()


def test_headless_world_runs_the_example_loop():
    """the loop of example/path_track/path_track_diff.py:15-40 against the headless world (YAML subset of ir-sim)"""
    from collections import namedtuple
    import rda_planner_amd.world as irsim
    env = irsim.make(os.path.join(os.path.dirname(__file__), "golden", "world_path_track_diff.yaml"))
    assert env.step_time == 0.1 and len(env.get_obstacle_info_list()) == 13
    kinds = [o.cone_type for o in env.get_obstacle_info_list()]
    assert kinds.count("norm2") == 12 and kinds.count("Rpositive") == 1
    assert float(env.get_obstacle_info_list()[0].radius) == 1.5 and float(env.get_obstacle_info_list()[5].radius) == 1.0
    car = namedtuple("car", "G h cone_type wheelbase max_speed max_acce dynamics")
    info_r = env.get_robot_info()
    car_tuple = car(info_r.G, info_r.h, info_r.cone_type, info_r.shape[2], [10, 1], [10, 0.5], "diff")
    ref_path_list = sc.path_track_ref()
    mpc_opt = MPC(car_tuple, ref_path_list, receding=10, sample_time=env.step_time, process_num=4, iter_num=2, obstacle_order=True,
                  ro1=300, max_edge_num=4, max_obs_num=13, slack_gain=8, _backend=oracle_backend)
    moved = env.get_obstacle_info_list()[-1].center.copy()
    min_clear = np.inf
    for i in range(120):
        opt_vel, info = mpc_opt.control(env.robot.state[0:3], 4, env.get_obstacle_info_list())
        env.step(opt_vel)
        env.render(show_traj=True)
        min_clear = min(min_clear, env.clearance())
        if env.done() or info["arrive"]:
            break
    assert not env.collided and min_clear > 0.05
    assert np.linalg.norm(env.get_obstacle_info_list()[-1].center - moved) > 0.1      # the dynamic obstacles do move
    assert np.linalg.norm(env.robot.state[0:2] - np.array([[10.0], [42.0]])) > 15.0   # and the robot made progress


def test_headless_world_loads_the_dynamic_obs_scene():
    """the reference's dynamic scene (example/dynamic_obs/dynamic_obs.yaml:24-32) as a fixture of the headless world: 7 moving circles with the radii of the
    YAML's shape list, and the example's loop (dynamic_obs.py:22 keywords) steps on the oracle backend"""
    from collections import namedtuple
    import rda_planner_amd.world as irsim
    env = irsim.make(os.path.join(os.path.dirname(__file__), "golden", "world_dynamic_obs.yaml"))
    obs = env.get_obstacle_info_list()
    assert len(obs) == 7 and all(o.cone_type == "norm2" for o in obs)
    assert [float(o.radius) for o in obs] == [0.5, 0.6, 0.7, 1.0, 0.5, 0.5, 0.5]
    assert all(0.3 <= float(np.linalg.norm(o.velocity)) <= 1.0 for o in obs)
    assert np.allclose(env.robot.state.ravel(), [10, 40, 1.57])
    car = namedtuple("car", "G h cone_type wheelbase max_speed max_acce dynamics")
    ri = env.get_robot_info()
    mpc_opt = MPC(car(ri.G, ri.h, ri.cone_type, ri.wheelbase, [10, 1], [10, 1.0], "acker"), sc.path_track_ref(), receding=10, sample_time=env.step_time,
                  process_num=5, iter_num=2, max_edge_num=4, max_obs_num=6, min_sd=0.5, wu=0.2, obstacle_order=True, _backend=oracle_backend)
    for i in range(12):
        opt_vel, info = mpc_opt.control(env.robot.state, 6, env.get_obstacle_info_list())
        assert info["status"] == 0 if "status" in info else True
        env.step(opt_vel)
    assert not env.collided and np.linalg.norm(env.robot.state[0:2] - np.array([[10.0], [40.0]])) > 2.0


def test_corridor_example_is_traversed():
    """BASELINE config C2 (example/corridor/corridor.py:8-30, corridor.yaml:22-33): the straight reference path is blocked by
    four boxes inside a 8 m wide corridor; with the reference's default MPC parameters the robot slaloms to the goal.
    This is what tie-break T1 (central separating normal in the slack regime) is for: with max-clearance duals the
    approach is head-on and the robot pushes into the first box at step 17."""
    car_a = sc.rectangle_robot(dynamics="acker")
    path = sc.line_path([0, 20, 0], [60, 20, 0], 0.1)
    obs = sc.scene_corridor(n_extra=0)
    mpc = MPC(car_a, [p.copy() for p in path], sample_time=0.1, max_edge_num=4, max_obs_num=6, _backend=oracle_backend)
    state = np.array([[0.0], [20.0], [0.0]])
    minc, ys = np.inf, []
    for i in range(260):
        u, info = mpc.control(state, 4, list(obs))
        state = sc.kinematic_step(state, u, car_a, 0.1)
        minc = min(minc, sc.clearance(car_a, state, obs))
        ys.append(state[1, 0])
        if info["arrive"]:
            break
    assert info["arrive"] and minc > 0.2, (info["arrive"], minc)
    assert max(ys) > 21.3 and min(ys) < 19.3                       # it went above the first box and below the second


# ---- lidar front end (reference example/lidar_nav/lidar_path_track.py:20-60) ----------------------------------
def test_dbscan_equals_scikit_learn():
    """the numpy DBSCAN labels every point like sklearn.cluster.DBSCAN (cluster numbering, border points, noise)"""
    sk = pytest.importorskip("sklearn.cluster")
    from rda_planner_amd import lidar
    rng = np.random.default_rng(0)
    for trial in range(120):
        n, k = int(rng.integers(4, 150)), int(rng.integers(1, 6))
        cent = rng.uniform(-10, 10, (k, 2))
        X = cent[rng.integers(0, k, n)] + rng.normal(0, rng.uniform(0.2, 1.5), (n, 2))
        eps, ms = float(rng.choice([0.5, 1.0, 2.0])), int(rng.choice([3, 6, 10]))
        assert np.array_equal(lidar.dbscan(X, eps, ms), sk.DBSCAN(eps=eps, min_samples=ms).fit_predict(X)), trial
    assert lidar.dbscan(np.zeros((0, 2))).shape == (0,)


def test_min_area_rect_known_answers():
    from rda_planner_amd import lidar
    rng = np.random.default_rng(1)
    for trial in range(60):
        th = rng.uniform(0, np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        L, W = rng.uniform(1, 5), rng.uniform(0.3, 2)
        corners = np.array([[0, 0], [L, 0], [L, W], [0, W]])
        shift = rng.uniform(-5, 5, 2)
        pts = np.vstack([corners, rng.uniform([0, 0], [L, W], (30, 2))]) @ R.T + shift
        box = lidar.min_area_rect(pts)
        want = corners @ R.T + shift
        # the same four corners (any starting corner), counter-clockwise
        d = np.linalg.norm(box[:, None, :] - want[None, :, :], axis=2)
        assert d.min(axis=1).max() < 1e-9 and len(set(d.argmin(axis=1))) == 4
        e = np.roll(box, -1, axis=0) - box
        assert all(e[i][0] * e[(i + 1) % 4][1] - e[i][1] * e[(i + 1) % 4][0] > 0 for i in range(4))
    # degenerate clusters stay bounded obstacles: collinear points, a single point
    seg = lidar.min_area_rect(np.array([[0.0, 0.0], [1.0, 1.0], [2.0, 2.0], [0.5, 0.5]]))
    assert abs(np.linalg.norm(seg[1] - seg[0]) * np.linalg.norm(seg[2] - seg[1]) - 2 * np.sqrt(2) * 0.01) < 1e-12
    pt = lidar.min_area_rect(np.array([[3.0, 4.0]] * 5))
    assert np.allclose(pt.mean(axis=0), [3, 4]) and np.allclose(np.ptp(pt, axis=0), 0.01)


def test_lidar_scan_and_scan_box():
    """ray casting of the headless world against known geometry, then the scan_box chain on it"""
    import rda_planner_amd.world as irsim
    from rda_planner_amd import lidar
    cfg = {"world": {"step_time": 0.1},
           "robot": [{"kinematics": {"name": "acker"}, "shape": {"name": "rectangle", "length": 4.6, "width": 1.6, "wheelbase": 3},
                      "state": [0, 0, 0.5], "sensors": [{"type": "lidar2d", "range_max": 10, "angle_range": np.pi, "number": 181}]}],
           "obstacle": [{"number": 1, "distribution": {"name": "manual"}, "state": [[6 * np.cos(0.5), 6 * np.sin(0.5)]], "shape": [{"name": "circle", "radius": 1.0}]},
                        {"number": 1, "distribution": {"name": "manual"}, "state": [[0, 0, 0]],
                         "shape": [{"name": "polygon", "vertices": [[-4, 3], [-2, 3], [-2, 30], [-4, 30]]}]}]}
    env = irsim.World(cfg)
    scan = env.get_lidar_scan()
    r = np.asarray(scan["ranges"])
    assert len(r) == 181 and scan["angle_min"] == -scan["angle_max"] and scan["range_max"] == 10
    assert abs(r[90] - 5.0) < 1e-12                                  # straight ahead: centre distance 6 - radius 1
    k = 90 + 5                                                       # 5 degrees off axis: law of cosines
    a = np.deg2rad(5.0)
    assert abs(r[k] - (6 * np.cos(a) - np.sqrt(1 - (6 * np.sin(a)) ** 2))) < 1e-9
    assert r[0] == 10.0                                              # nothing on the right
    boxes = lidar.scan_box(env.robot.state, scan)
    assert len(boxes) == 2 and all(b.cone_type == "Rpositive" and b.vertex.shape == (2, 4) for b in boxes)
    cents = sorted([b.vertex.mean(axis=1) for b in boxes], key=lambda c: c[0])
    # the circle is seen as the box of its near arc: centre on the line of sight, between 5 m and 6 m away
    c = cents[1]
    assert abs(np.arctan2(c[1], c[0]) - 0.5) < 0.02 and 5.0 < np.linalg.norm(c) < 6.0
    # the wall: only its visible faces, all corners on / inside the true polygon (1 cm minimum thickness)
    wv = [b for b in boxes if b.vertex.mean(axis=1)[0] < 0][0].vertex
    assert wv[0].min() > -4.02 and wv[0].max() < -1.98 and wv[1].min() > 2.98
    assert lidar.scan_box(env.robot.state, dict(scan, ranges=np.full(181, 10.0))) == []


LIDAR_STARTS = [(0.0, 0.0, 0.0), (0.007, 0.270, -0.057), (0.269, -0.113, -0.012), (0.197, -0.054, 0.008), (-0.283, 0.152, 0.006),
                (-0.102, 0.173, -0.031), (-0.028, -0.220, -0.016), (-0.178, -0.143, 0.040)]


def lidar_closed_loop(start_offset, **backend_kw):
    """the loop of example/lidar_nav/lidar_path_track.py:64-95 against the headless world, from the yaml start state shifted by
    (dx, dy, dtheta): the planner only knows the boxes `scan_box` builds from each scan.  -> (arrived, collided, min clearance,
    most boxes seen in one scan)"""
    import rda_planner_amd.world as irsim
    from rda_planner_amd.lidar import scan_box
    env = irsim.make(os.path.join(os.path.dirname(__file__), "golden", "world_lidar_track.yaml"))
    for k in range(3):
        env.robot.state[k, 0] += start_offset[k]
    ri = env.get_robot_info()
    car_tuple = sc.car(ri.G, ri.h, ri.cone_type, ri.wheelbase, [10, 1], [10, 0.5], "acker")
    mpc_opt = MPC(car_tuple, sc.path_track_ref(), receding=10, sample_time=env.step_time, process_num=4, iter_num=2, max_edge_num=4,
                  max_obs_num=4, obstacle_order=True, wu=1.0, slack_gain=13, **backend_kw)
    min_clear, seen, arrived = np.inf, 0, False
    for i in range(500):
        obs_list = scan_box(env.robot.state, env.get_lidar_scan())
        seen = max(seen, len(obs_list))
        opt_vel, info = mpc_opt.control(env.robot.state, 4, obs_list)
        env.step(opt_vel)
        min_clear = min(min_clear, env.clearance())
        if env.done() or info["arrive"]:
            arrived = info["arrive"]
            break
    return bool(arrived), bool(env.collided), float(min_clear), seen


def test_lidar_example_reaches_the_goal_from_most_starts():
    """BASELINE config C3.  With iter_num = 2 the closed loop is CHAOTIC in this scene: a 1e-6 change of one control (a different
    but equally converged interior-point path of the su-problem, a re-ordered sum) flips an early-stop or a support decision some
    steps later and the run ends elsewhere.  Which starts succeed therefore changes with every numerical detail, the RATE does
    not: 8 of 16 perturbed starts for every su-solver variant tried (cold, warm, three end-game settings), 4 of 8 for the
    UNMODIFIED reference on an interior-point stand-in in the corridor scene.  The test asserts the rate,
    not one trajectory."""
    runs = [lidar_closed_loop(s, _backend=oracle_backend) for s in LIDAR_STARTS]
    ok = [a and not c and mc > 0.0 for a, c, mc, _ in runs]
    assert sum(ok) >= 3, runs
    assert max(r[3] for r in runs) >= 2


def test_flatten_scene_matches_object_by_object_packing():
    """`RDA_solver.flatten_scene` (one concatenate + one scatter over all obstacle objects) against packing the objects one by
    one the way the reference reads them (mpc.py:192-203: `.cone_type`, `.vertex` 2xk, `.center`, `.radius`, `.velocity`):
    polygons of mixed vertex counts, circles, skipped cone types, 1-D / nested-list attributes, the empty list, too many
    vertices"""
    from rda_planner_amd import scenarios as sc
    from rda_planner_amd.rda_solver import RDA_solver

    class Host:                       # flatten_scene only reads max_edge_num
        max_edge_num = 5

    def packed(objs, E):
        objs = [o for o in objs if o.cone_type in ("norm2", "Rpositive")]
        n = len(objs)
        kind, nvert, geom, vel = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((n, E, 2)), np.zeros((n, 2))
        for i, o in enumerate(objs):
            vel[i] = np.asarray(o.velocity, float).ravel()[0:2]
            if o.cone_type == "norm2":
                kind[i] = 1
                geom[i, 0] = np.asarray(o.center, float).ravel()[0:2]
                geom[i, 1, 0] = float(o.radius)
            else:
                V = np.asarray(o.vertex, float)[0:2]
                nvert[i] = V.shape[1]
                geom[i, :V.shape[1]] = V.T
        return n, kind, nvert, geom, vel

    rng = np.random.default_rng(11)
    poly = [sc.Obstacle(center=np.zeros((2, 1)), radius=1.0, vertex=rng.uniform(0, 9, (2, k)), cone_type="Rpositive",
                        velocity=rng.uniform(-1, 1, (2, 1))) for k in (3, 4, 5, 4, 3, 3, 5, 4)]
    circ = [sc.Obstacle(center=rng.uniform(0, 9, (2, 1)), radius=float(rng.uniform(0.3, 1)), vertex=None, cone_type="norm2",
                        velocity=rng.uniform(-1, 1, (2, 1))) for _ in range(4)]

    class Skipped:
        cone_type, vertex, velocity, center, radius = "exponential", None, np.zeros((2, 1)), np.zeros((2, 1)), 1.0

    mixed = [poly[0], circ[0], poly[1], Skipped(), poly[2], circ[1], circ[2], poly[3], poly[4], circ[3]]
    lists = [p._replace(vertex=p.vertex.tolist(), velocity=[0.25, -0.5]) for p in poly[:3]]      # nested lists, 1-D velocity
    three_rows = [p._replace(vertex=np.vstack((p.vertex, np.ones((1, p.vertex.shape[1]))))) for p in poly[:4]]   # homogeneous 3xk
    for name, objs in (("polygons", poly), ("circles", circ), ("mixed", mixed), ("lists", lists), ("three rows", three_rows), ("empty", [])):
        got, want = RDA_solver.flatten_scene(Host, objs), packed(objs, Host.max_edge_num)
        assert got[0] == want[0], name
        for g, w in zip(got[1:], want[1:]):
            assert g.dtype == w.dtype and g.shape == w.shape and np.array_equal(g, w), name
    Host.max_edge_num = 4
    assert RDA_solver.flatten_scene(Host, poly) is None            # a pentagon does not fit: the caller converts on the host
    Host.max_edge_num = 2
    assert RDA_solver.flatten_scene(Host, circ) is None            # a circle needs three rows


def test_flatten_scene_scalar_velocity_like_the_lidar_examples():
    """the reference's lidar examples build `obs(None, None, vertices, 'Rpositive', 0)` (example/lidar_nav/lidar_path_track.py:55-58):
    a SCALAR velocity, which the reference only ever feeds to np.linalg.norm and to `velocity * t` (mpc.py:447-472).  The flat
    scene must carry two numbers per obstacle for it (the (n, 1) array made rda_upload_scene read out of bounds)."""
    from collections import namedtuple

    class Host:
        max_edge_num = 4
    obs = namedtuple("obs", "center radius vertex cone_type velocity")
    V = np.array([[0.0, 1.0, 1.0, 0.0], [0.0, 0.0, 1.0, 1.0]])
    scalar = [obs(None, None, V + k, "Rpositive", 0) for k in range(3)]
    n, kind, nvert, geom, vel = RDA_solver.flatten_scene(Host, scalar)
    assert n == 3 and vel.shape == (3, 2) and vel.flags["C_CONTIGUOUS"] and not vel.any()
    mixed = [obs(None, None, V, "Rpositive", 0.5), obs(None, None, V, "Rpositive", np.array([[0.1], [0.2]])),
             obs(None, None, V, "Rpositive", [0.3, -0.4]), obs(np.array([[1.0], [2.0]]), 0.5, None, "norm2", np.float64(0.0))]
    n, kind, nvert, geom, vel = RDA_solver.flatten_scene(Host, mixed)
    assert vel.shape == (4, 2) and np.array_equal(vel, [[0.5, 0.5], [0.1, 0.2], [0.3, -0.4], [0.0, 0.0]])
    assert RDA_solver.flatten_scene(Host, [obs(None, None, V, "Rpositive", np.zeros(0))]) is None       # host conversion reports it
    assert RDA_solver.flatten_scene(Host, [obs(None, None, V, "Rpositive", np.nan)]) is None


def _degenerate_scene():
    """a healthy box, a NaN polygon, a zero-area polygon (all vertices equal), a non-convex quadrilateral, a circle of radius 0"""
    good = sc.box(12, 27, 3, 2, 0.3)
    nan = sc.box(14, 22, 2, 2, 0.0)
    nan = nan._replace(vertex=nan.vertex * np.array([[np.nan, 1, 1, 1], [1, 1, 1, 1]]))
    point = good._replace(vertex=np.tile(np.array([[18.0], [23.0]]), (1, 4)))
    dart = good._replace(vertex=np.array([[20.0, 23.0, 21.0, 21.5], [26.0, 27.0, 28.5, 27.0]]))       # reflex corner
    dot = sc.circle(16, 29, 0.0)
    return [good, nan, point, dart, dot]


def _run_degenerate(backend_kw, steps=4):
    import contextlib
    import io
    car_t = sc.rectangle_robot(dynamics="acker")
    path = sc.line_path([4, 25, 0], [40, 25, 0], 0.1)
    mpc = MPC(car_t, [p.copy() for p in path], receding=8, iter_num=3, max_edge_num=4, max_obs_num=5, obstacle_order=False,
              time_print=False, **backend_kw)
    state = path[0].copy().reshape(3, 1)
    out = []
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        for _ in range(steps):
            u, info = mpc.control(state, 4.0, _degenerate_scene())
            out.append((u.copy(), dict(info)))
            state = sc.kinematic_step(state, u, car_t, 0.1)
    return out, mpc, buf.getvalue()


def test_failure_semantics_with_degenerate_obstacles():
    """VERDICT r01 missing #3: a LamMuZ sub-problem that cannot be solved (NaN data) keeps its previous duals and makes the dual
    residual inf, which blocks the early stop - rda_solver.py:781-793 - and is counted in info['lmz_fail']; zero-area,
    non-convex and zero-radius obstacles are ordinary (solvable) inputs for the reference and stay so here"""
    from oracle.oracle_backend import oracle_backend
    out, mpc, printed = _run_degenerate({"_backend": oracle_backend})
    T = 8
    for u, info in out:
        assert np.isfinite(u).all() and np.isfinite(np.hstack(info["opt_state_list"])).all()
        assert info["lmz_fail"] == 3 * T          # the NaN slot, every stage, every one of the iter_num iterations (no early stop)
        assert info["resi_dual"] == np.inf and info["iters"] == 3 and np.isfinite(info["resi_pri"])
    assert "Update Lam Mu Fail" in printed
    st = mpc.rda.get_state()
    assert not st["lam"][1].any() and not st["mu"][1].any() and not st["z"][1].any()       # slot 1 kept its (zero) duals
    assert all(np.isfinite(st[k]).all() for k in st)
    assert st["mu"][0].any() and st["mu"][3].any()                                         # the healthy and the non-convex slot are solved


# ---------------------------------------------------------------------------------------------------------------------------
# flatten_scene: the C walk over the obstacle objects (csrc/flatten_ext.c) against the numpy implementation it accelerates
# ---------------------------------------------------------------------------------------------------------------------------
def _flat_solver(E=4):
    from rda_planner_amd.rda_solver import RDA_solver
    s = RDA_solver.__new__(RDA_solver)
    s.max_edge_num = E
    return s


def _same_scene(a, b):
    assert (a is None) == (b is None)
    if a is not None:
        assert a[0] == b[0]
        for x, y in zip(a[1:], b[1:]):
            assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y)


def test_flatten_extension_equals_numpy_flatten():
    import rda_planner_amd.rda_solver as rs
    from rda_planner_amd import _lib
    if rs._flatten is None:
        _lib.build_flatten_ext(force=True)
        import importlib
        rs._flatten = importlib.import_module("rda_planner_amd._flatten")
    sv = _flat_solver(5)
    rng = np.random.default_rng(3)
    obs = sc.scene_polygons(40, lo=(0, 0), hi=(50, 50), seed=4, moving=True)
    obs += [sc.circle(float(rng.uniform(0, 50)), float(rng.uniform(0, 50)), float(rng.uniform(0.3, 2)), (0.1 * i, -0.2)) for i in range(7)]
    obs.append(sc.Obstacle(None, None, np.array([[1.0, 3.0, 3.0, 1.0, 0.5], [1.0, 1.0, 2.0, 2.5, 1.5]]), "Rpositive", 0))          # scalar velocity (lidar examples)
    obs.append(sc.Obstacle(None, None, np.asfortranarray(np.array([[5.0, 7.0, 6.0], [5.0, 5.0, 7.0]])), "Rpositive", np.zeros((2, 1))))   # Fortran order
    obs.append(sc.Obstacle(None, None, np.array([[9.0, 8.0, 8.0, 9.0], [9.0, 9.0, 8.0, 8.0]])[:, ::-1], "Rpositive", np.array([0.3, 0.1])))  # reversed view, 1-D velocity
    obs.append(sc.Obstacle(None, None, np.array([[2.0, 4.0, 3.0], [2.0, 2.0, 4.0]]), "Rpositive", np.zeros((3, 1))))           # 3x1 velocity
    rng.shuffle(obs)
    fast = sv.flatten_scene(list(obs))
    assert rs._flatten.flatten(list(obs), 5, np.zeros(len(obs), np.int32), np.zeros(len(obs), np.int32), np.zeros((len(obs), 5, 2)), np.zeros((len(obs), 2))) == 0
    _same_scene(fast, sv._flatten_scene_numpy(list(obs)))
    assert fast[4].flags["C_CONTIGUOUS"] and fast[3].flags["C_CONTIGUOUS"]
    _same_scene(sv.flatten_scene(tuple(obs)), fast)
    # inputs the C walk declines: the numpy code decides (same answer through the public entry point)
    odd = [
        [sc.Obstacle(None, None, np.array([[1, 3, 3], [1, 1, 2]]), "Rpositive", np.zeros((2, 1)))],                   # integer vertices
        [sc.Obstacle(None, None, [[1.0, 3.0, 3.0], [1.0, 1.0, 2.0]], "Rpositive", np.zeros((2, 1)))],                 # nested lists
        [sc.Obstacle(None, None, np.zeros((2, 7)), "Rpositive", np.zeros((2, 1)))],                                    # more vertices than E
        [sc.Obstacle(None, None, np.array([[1.0, 3.0, 3.0], [1.0, 1.0, 2.0]]), "Rpositive", np.array([[np.nan], [0.0]]))],   # non-finite velocity
        [sc.circle(1, 2, 0.5), sc.Obstacle(None, None, None, "exponential", np.zeros((2, 1)))],                       # a cone type the reference skips
    ]
    for lst in odd:
        n = len(lst)
        assert rs._flatten.flatten(lst, 5, np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((n, 5, 2)), np.zeros((n, 2))) == -1
        _same_scene(sv.flatten_scene(list(lst)), sv._flatten_scene_numpy(list(lst)))
    assert _flat_solver(2).flatten_scene([sc.circle(1, 2, 0.5)]) is None                                               # circles need E >= 3
    _same_scene(sv.flatten_scene([]), sv._flatten_scene_numpy([]))


def test_env_switches_are_applied_by_the_host_package_not_by_the_library(monkeypatch):
    """round 5: librda_hip.so reads no environment variable; the RDA_* switches of the A/B tools and of tests/test_gpu_switches.py
    are applied by rda_solver.hip_options to the rda_opts it hands to rda_create_opts.  The table must name fields of the struct, fill arrays in
    order (sscanf semantics of the old C code: what is not given stays) and leave everything else alone."""
    import re
    from rda_planner_amd._capi import Opts
    from rda_planner_amd.rda_solver import _ENV_SWITCHES, _apply_env
    known = {f[0] for f in Opts._fields_}
    assert all(f in known for fields in _ENV_SWITCHES.values() for f in fields)
    o = Opts()
    o.su_warm[0], o.su_warm[1], o.su_warm_cap, o.su_cold_probe = 1e-3, 1e-3, 30, 8
    o.su_easy_max = 2
    monkeypatch.setenv("RDA_SU_WARM", "0,0,0")
    monkeypatch.setenv("RDA_LMZ_MODE", "1")
    monkeypatch.setenv("RDA_LMZ_MU", "1e-3")
    monkeypatch.setenv("RDA_SU_EASY", "1e-6,1e-6,1e-6,0.9,1e-3,3")
    monkeypatch.setenv("RDA_SU_COLD_FROM", "5")
    _apply_env(o)
    assert list(o.su_warm) == [0.0, 0.0] and o.su_warm_cap == 0 and o.lmz_mode == 1 and o.lmz_mu == 1e-3
    assert list(o.su_easy) == [1e-6, 1e-6, 1e-6, 0.9, 1e-3] and o.su_easy_max == 3
    assert o.su_cold_from == 5 and o.su_cold_probe == 8           # second value not given: stays
    #  an empty list element keeps the entry; a value that is not a number is an error that names the variable
    monkeypatch.setenv("RDA_SU_EASY", "1e-5,,1e-4")
    _apply_env(o)
    assert list(o.su_easy)[:3] == [1e-5, 1e-6, 1e-4]
    monkeypatch.setenv("RDA_LMZ_MODE", "yes")
    with pytest.raises(ValueError, match="RDA_LMZ_MODE"):
        _apply_env(o)
    monkeypatch.delenv("RDA_LMZ_MODE")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_src = "".join(open(os.path.join(root, "rda_planner_amd", "csrc", f)).read() for f in os.listdir(os.path.join(root, "rda_planner_amd", "csrc")) if f.endswith((".hip", ".h")))
    assert not re.search(r"\bgetenv\s*\(", lib_src), "librda_hip.so must not read the environment"
