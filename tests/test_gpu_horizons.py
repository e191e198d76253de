"""-m gpu : EVERY receding horizon the interface accepts (1 .. RDA_TMAX = 64) creates a handle and steps, the horizons without a compile-time
instantiation of k_su (generic T; LDS carve-up computed at run time) against the oracle.  Round 5: rda_create failed with RDA_ERR_HIP for
T = 36 .. 40 - the su workgroup asked for more than 160 KB of LDS there (su_device.h `near_cap`) and no test created such a handle; a
random soak over horizons the examples do not use (tools/soak.py --exotic) found it.  su_device.h now asserts the fit at compile time."""
import numpy as np
import pytest

from rda_planner_amd import scenarios as sc
from tests.helpers import TOL_U

pytestmark = pytest.mark.gpu


def _scene(n, steps):
    path = sc.line_path([4, 25, 0], [4 + 0.4 * steps + 30, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    return path, sc.scene_polygons(n, lo=(8, 14), hi=(44, 36), seed=5, keep_clear=clear, clear_radius=3.0)


def test_every_horizon_creates_and_steps(hip):
    from rda_planner_amd.mpc import MPC
    car_t = sc.rectangle_robot(dynamics="diff", wheelbase=0)
    path, obstacles = _scene(12, 4)
    for T in range(1, 65):
        m = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, receding=T, iter_num=2, max_edge_num=4, max_obs_num=12)
        st = path[0].copy().reshape(3, 1)
        for _ in range(2):
            u, info = m.control(st, 4.0, list(obstacles))
            assert info["status"] == 0 and np.all(np.isfinite(u)), T
            st = sc.kinematic_step(st, u, car_t, 0.1)
    with pytest.raises(RuntimeError):
        MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, receding=65, iter_num=2, max_edge_num=4, max_obs_num=12)


@pytest.mark.parametrize("T,dyn", [(36, "acker"), (40, "diff"), (40, "omni"), (51, "acker"), (64, "diff"), (7, "omni"), (33, "acker")])
def test_generic_horizons_against_the_oracle(hip, T, dyn):
    """closed loop, re-sorted every tick (the reference's default), step by step from the same solver state"""
    from rda_planner_amd.mpc import MPC
    from oracle.oracle_backend import oracle_backend
    car_t = sc.rectangle_robot(dynamics=dyn, wheelbase=3.0 if dyn == "acker" else 0)
    steps = 16
    path, obstacles = _scene(40, steps)
    kw = dict(sample_time=0.1, time_print=False, receding=T, iter_num=3, max_edge_num=4, max_obs_num=36, ro1=200)
    gpu = MPC(car_t, [p.copy() for p in path], **kw)
    cpu = MPC(car_t, [p.copy() for p in path], _backend=oracle_backend, **kw)
    st = path[0].copy().reshape(3, 1)
    worst = 0.0
    for k in range(steps):
        ug, ig = gpu.control(st.copy(), 4.0, list(obstacles))
        uc, ic = cpu.control(st.copy(), 4.0, list(obstacles))
        assert ig["status"] == 0 and ic["status"] == 0, k
        assert ig["iters"] == ic["iters"], k
        worst = max(worst, float(np.abs(ug - uc).max()))
        cpu.rda.set_state(gpu.rda.get_state()); cpu.cur_vel_array = gpu.cur_vel_array.copy(); cpu.cur_index = gpu.cur_index
        st = sc.kinematic_step(st, ug, car_t, 0.1)
    print(f"T={T} {dyn}: max |du| {worst:.2e}")
    assert worst <= TOL_U, worst
