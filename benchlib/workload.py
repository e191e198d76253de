"""bench.py's workload (BASELINE.json north star: acker rectangle robot on a straight path through a seeded polygon field) and the Python
closed loop that records the step inputs every other leg must reproduce.  Split out of bench.py in round 5."""
import time

import numpy as np


def build_workload(seed_offset=0, n_obs=200, T=20, n_steps=110, moving=False):
    """straight reference path through a seeded field of polygons; long enough that the robot never arrives
    (an arrived robot would make every later step trivial)"""
    from rda_planner_amd import scenarios as sc
    car_t = sc.rectangle_robot(dynamics="acker")
    length = max(40.0, 0.4 * n_steps + 12.0)
    path = sc.line_path([4, 25, 0], [4 + length, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in path[::10]])
    obstacles = sc.scene_polygons(n_obs, lo=(8, 10), hi=(4 + length - 4, 40), seed=sc.SEED + seed_offset, keep_clear=clear, clear_radius=3.2,
                                  moving=moving)     # moving: velocities U[-1,1]^2 m/s, (A, b) per horizon stage (BASELINE dynamic_obs)
    kw = dict(receding=T, iter_num=4, max_edge_num=4, max_obs_num=n_obs, ro1=200, obstacle_order=True)
    return car_t, path, obstacles, kw


def record_trace(car_t, path, obstacles, kw, n_steps, backend=None, post_init=None, stage_every_step=False, moving=False):
    """closed loop with the solver in the loop; returns per-step inputs and the staged obstacle arrays (of the first step, or - for
    a scene that is re-sorted every tick - of every step: trace["staged"])"""
    from rda_planner_amd.mpc import MPC
    from rda_planner_amd import scenarios as sc
    extra = {"_backend": backend} if backend is not None else {}
    # host-side obstacle staging here: the spy below needs the staged arrays for the device-resident replay
    mpc = MPC(car_t, [p.copy() for p in path], sample_time=0.1, time_print=False, device_obstacles=False, device_track=False, **kw, **extra)
    if post_init is not None:
        post_init(mpc.rda)
    T = kw["receding"]
    state = path[0].copy().reshape(3, 1)
    tr = {"nom_s": [], "nom_u": [], "ref": [], "speed": [], "u": [], "u_solver": []}
    arrived = 0
    orig = mpc.rda.iterative_solve
    staged = {}
    per_step = []

    def spy(nom_s, nom_u, ref_states, ref_speed, obstacle_list, **k):
        tr["nom_s"].append(np.array(nom_s, float).reshape(3, T + 1))
        tr["nom_u"].append(np.array(nom_u, float).reshape(2, T))
        tr["ref"].append(np.array(np.hstack(ref_states)[0:3, :], float))
        tr["speed"].append(float(ref_speed))
        if not staged or stage_every_step:
            n, A, b, cone, per_t = mpc.rda._stage(list(obstacle_list))
            if not staged:
                staged.update(n=n, A=A, b=b, cone=cone, per_t=per_t)
            if stage_every_step:
                per_step.append((n, A.copy(), b.copy(), cone.copy(), per_t))
        u_sol, info_sol = orig(nom_s, nom_u, ref_states, ref_speed, obstacle_list, **k)
        tr["u_solver"].append(np.array(u_sol, float))
        return u_sol, info_sol

    mpc.rda.iterative_solve = spy
    t0 = time.perf_counter()
    min_clear = np.inf
    for k_ in range(n_steps):
        # static obstacles + obstacle_order=False semantics for the replay: keep slot binding fixed
        cur = obstacles if not moving else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k_)) for o in obstacles]
        u, info = mpc.control(state, 4.0, list(cur))
        tr["u"].append(u.copy())
        arrived += int(info["arrive"])
        state = sc.kinematic_step(state, u, car_t, 0.1)
    dt = time.perf_counter() - t0
    min_clear = sc.clearance(car_t, state, obstacles)
    out = {k: np.ascontiguousarray(np.array(v)) for k, v in tr.items()}
    out["closed_loop_s_per_step"] = dt / n_steps
    out["final_clearance"] = float(min_clear)
    out["arrived_steps"] = arrived
    out["staged"] = per_step
    return out, staged, mpc

