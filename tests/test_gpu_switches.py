"""-m gpu : the restructurings of round 2 change WHERE and WHEN work is done, never its result.  Each one has an environment switch
(read at rda_create) that restores the plain form; a closed loop with the switch thrown must reproduce the default closed loop bit
for bit: controls, states, residuals, iteration counts.

  RDA_LMZ_DENSE_FROM=0   every LamMuZ launch in the split form (common-path kernel + work-list kernel) instead of the fused kernel
  RDA_LMZ_SPLIT=0        (with DENSE_FROM=0) every launch as the fused two-workgroups-per-CU kernel
  RDA_SU_PRE=0           the su set-up evaluates all condensed terms itself (no block sums / near masks from the LamMuZ launch)
  RDA_LMZ_TAIL=1         the early-stop verdict and the hand-over by the last-arriving LamMuZ workgroup, not by the next su launch / k_finish
  RDA_SU_LIGHT=0         convergence pass with the factorisation
  RDA_ZERO_COPY=0        result through a D2H copy + stream synchronise
  RDA_EARLY_FINISH=0     k_finish hands the result over, not the su launch that detects the early stop
  RDA_FUSE_TRACK=0       k_track and k_su as two launches
"""
import numpy as np
import pytest

from rda_planner_amd import scenarios as sc

pytestmark = pytest.mark.gpu

SWITCHES = [{"RDA_LMZ_DENSE_FROM": "0"}, {"RDA_LMZ_DENSE_FROM": "0", "RDA_LMZ_SPLIT": "0"}, {"RDA_SU_PRE": "0"}, {"RDA_SU_LIGHT": "0"}, {"RDA_LMZ_TAIL": "1"},
            {"RDA_LMZ_TAIL": "1", "RDA_LMZ_DENSE_FROM": "0"},
            {"RDA_ZERO_COPY": "0"}, {"RDA_EARLY_FINISH": "0"}, {"RDA_FUSE_TRACK": "0"},
            {"RDA_ZERO_COPY": "0", "RDA_EARLY_FINISH": "0", "RDA_FUSE_TRACK": "0", "RDA_SU_PRE": "0", "RDA_LMZ_DENSE_FROM": "0"}]


def _loop(dyn, moving, steps=45):
    from rda_planner_amd.mpc import MPC
    car_t = sc.rectangle_robot(dynamics=dyn, wheelbase=3.0 if dyn == "acker" else 0)
    path = sc.line_path([4, 25, 0], [40, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    scene = sc.scene_polygons(30, lo=(8, 14), hi=(40, 36), seed=5, keep_clear=clear, clear_radius=3.0, moving=moving)
    scene.append(sc.circle(20.0, 29.5, 0.9, (0.0, -0.1)))
    mpc = MPC(car_t, [p.copy() for p in path], receding=12, iter_num=4, max_edge_num=4, max_obs_num=32, time_print=False)
    st = path[0].copy().reshape(3, 1)
    if dyn == "omni":
        st[2, 0] = 0.0
    out = []
    for k in range(steps):
        cur = [o if not np.any(o.velocity) else (o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) if o.cone_type == "Rpositive"
                                                 else o._replace(center=o.center + o.velocity * (0.1 * k))) for o in scene]
        u, info = mpc.control(st.copy(), 4.0, cur)
        out.append((u.copy(), np.hstack(info["opt_state_list"]).copy(), info["resi_dual"], info["resi_pri"], info["iters"], info["su_ipm_iters"]))
        st = sc.kinematic_step(st, u, car_t, 0.1)
    return out


@pytest.fixture(scope="module")
def defaults():
    return {(dyn, mv): _loop(dyn, mv) for dyn, mv in (("acker", False), ("diff", True))}


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: "+".join(f"{k[4:]}={v}" for k, v in e.items()))
@pytest.mark.parametrize("case", [("acker", False), ("diff", True)], ids=["acker-static", "diff-moving"])
def test_switch_reproduces_the_default_closed_loop(defaults, monkeypatch, env, case):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    got, want = _loop(*case), defaults[case]
    for k, (g, w) in enumerate(zip(got, want)):
        assert g[4] == w[4] and g[5] == w[5], (k, g[4:], w[4:])
        assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]), (k, float(np.abs(g[0] - w[0]).max()))
        assert g[2] == w[2] and g[3] == w[3], (k, g[2:4], w[2:4])


@pytest.mark.parametrize("T,dyn,moving", [(20, "acker", False), (25, "omni", True), (30, "diff", True), (10, "acker", False)])
def test_time_split_of_the_su_newton_system_changes_nothing_but_rounding(T, dyn, moving):
    """rda_opts::su_split (round 4): the horizons with a compile-time instantiation factorise and sweep the Newton system of the su
    interior point in two halves on two waves, joined by a 5 x 5 interface system - the SAME linear system as one recursion over the
    horizon.  Two handles, su_split = 1 / 0, stepped from the same state (re-synchronised every step): same ADMM and interior-point
    iteration counts (+-1 per step where a stop test sits on its threshold), controls equal to the level at which two runs of the same
    interior-point iteration with differently rounded Newton directions stop (1e-5; the stated tolerance of the parity tests is 5e-4)."""
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd.rda_solver import hip_options
    car_t = sc.rectangle_robot(dynamics=dyn, wheelbase=3.0 if dyn == "acker" else 0)
    path = sc.line_path([4, 25, 0], [40, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    scene = sc.scene_polygons(40, lo=(8, 14), hi=(40, 36), seed=7, keep_clear=clear, clear_radius=3.0, moving=moving)
    kw = dict(receding=T, iter_num=3, max_edge_num=4, max_obs_num=40, time_print=False)
    a = MPC(car_t, [p.copy() for p in path], hip_opts=hip_options(su_split=1), **kw)
    b = MPC(car_t, [p.copy() for p in path], hip_opts=hip_options(su_split=0), **kw)
    st = path[0].copy().reshape(3, 1)
    if dyn == "omni":
        st[2, 0] = 0.0
    worst = 0.0
    for k in range(40):
        cur = [o if not np.any(o.velocity) else o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in scene]
        ua, ia = a.control(st.copy(), 4.0, list(cur))
        ub, ib = b.control(st.copy(), 4.0, list(cur))
        assert ia["iters"] == ib["iters"] and ia["status"] == ib["status"] == 0, (k, ia["iters"], ib["iters"])
        assert abs(ia["su_ipm_iters"] - ib["su_ipm_iters"]) <= 1, (k, ia["su_ipm_iters"], ib["su_ipm_iters"])
        worst = max(worst, float(np.abs(ua - ub).max()), float(np.abs(a.cur_vel_array - b.cur_vel_array).max()))
        b.rda.set_state(a.rda.get_state())
        b.cur_vel_array = a.cur_vel_array.copy(); b.cur_index = a.cur_index
        st = sc.kinematic_step(st, ua, car_t, 0.1)
    print(f"T={T} {dyn}: max |u_split - u_unsplit| over the horizon {worst:.2e}")
    assert worst <= 1e-5
