"""
`MPC` - drop-in for RDA_planner.mpc.MPC (reference mpc.py:15-569).

This is the boundary CALLER of the accelerated path: reference-path
tracking, the open-loop rollout that produces the nominal trajectory, obstacle -> (A, b, cone)
conversion and distance ordering.  It is plain host Python by design (O(T) scalar work per tick,
out of scope for acceleration); what matters is that it feeds `RDA_solver.iterative_solve`
exactly what the reference's `MPC.control` feeds it, quirks included (Q5, Q12).
Constructor / method names and argument meaning follow the reference one-to-one.
"""
from collections import namedtuple
from math import cos, inf, pi, sin, sqrt, tan

import numpy as np

from .rda_solver import RDA_solver

rdaobs = namedtuple("rdaobs", "A b cone_type center vertex")          # reference mpc.py:12


class MPC:
    def __init__(self, car_tuple, ref_path, receding: int = 10, sample_time: float = 0.1, iter_num: int = 4,
                 enable_reverse: bool = False, rda_obstacle: bool = False, obstacle_order: bool = True,
                 max_edge_num: int = 5, max_obs_num: int = 5, process_num: int = 4, accelerated: bool = True,
                 time_print: bool = False, goal_index_threshold: int = 1, **kwargs) -> None:
        # reference mpc.py:67-125
        self.car_tuple, self.ref_path = car_tuple, ref_path
        self.L, self.dynamics = car_tuple.wheelbase, car_tuple.dynamics
        self.receding, self.dt = receding, sample_time
        self.state, self.cur_index = np.zeros((3, 1)), 0
        self.cur_vel_array = kwargs.get("init_vel", np.zeros((2, receding)))
        # extension (not in the reference): device_obstacles=True lets the accelerated backend convert, predict, sort
        # and stage the raw obstacles on the GPU (rda_step_scene) instead of the host code below; same staged values
        self.device_obstacles = bool(kwargs.get("device_obstacles", True))
        # extension: device_track=True additionally runs pre_process (closest waypoint, nominal roll-out, reference
        # sampling) on the GPU (rda_step_tracked) - same values to rounding, the host code below is the fallback
        self.device_track = bool(kwargs.get("device_track", True))
        self._dev_path_key, self._dev_u, self._path_ticks = None, None, 0
        solver_kwargs = {k: v for k, v in kwargs.items() if k not in ("init_vel", "device_obstacles", "device_track")}
        self.rda = RDA_solver(receding, car_tuple, max_edge_num, max_obs_num, iter_num=iter_num,
                              step_time=sample_time, process_num=process_num, accelerated=accelerated,
                              time_print=time_print, **solver_kwargs)
        self.enable_reverse, self.rda_obstacle = enable_reverse, rda_obstacle
        self.obstacle_order, self.goal_index_threshold = obstacle_order, goal_index_threshold
        if enable_reverse:
            self.curve_list = self.split_path(self.ref_path)
            self.curve_index = 0

    # ------------------------------------------------------------------ control (mpc.py:127-187)
    def control(self, state, ref_speed=5, obstacle_list=[], **kwargs):
        if self._tracks(kwargs):
            return self._control_tracked(state, ref_speed, obstacle_list, **kwargs)
        cur_ref_path, speed, state_pre_array, ref_traj_list = self._begin(state, ref_speed, **kwargs)
        scene = None
        if not self.rda_obstacle and self.device_obstacles and self.rda.has_scene:
            scene = self.rda.flatten_scene(obstacle_list)
        if scene is not None:
            u_opt_array, info = self.rda.iterative_solve_scene(
                state_pre_array, self.cur_vel_array, ref_traj_list, speed, scene,
                np.asarray(state, float)[0:2], self.obstacle_order, **kwargs)
        else:
            if not self.rda_obstacle:
                rda_obs_list = self.convert_rda_obstacle(obstacle_list, self.state, self.obstacle_order)
            else:
                rda_obs_list = obstacle_list
            u_opt_array, info = self.rda.iterative_solve(
                state_pre_array, self.cur_vel_array, ref_traj_list, speed, rda_obs_list, **kwargs)
        return self._end(cur_ref_path, u_opt_array, info)

    # ---- extension: pre_process on the device (rda_step_tracked); same control, cur_index, path side effect ----------
    def _tracks(self, kwargs):
        return self.device_track and self.rda.has_track and set(kwargs) <= {"threshold", "ind_range"}

    def _piece(self, state):
        if np.shape(state)[0] > 3:
            state = state[0:3]
        self.state = state
        if self.enable_reverse:
            cur_ref_path = self.curve_list[self.curve_index]
            return cur_ref_path, cur_ref_path[0][-1, 0]
        return self.ref_path, 1

    def _stage_obstacles(self, obstacle_list, in_tick=False):
        """obstacles into the solver's slots without solving: raw scene through the device pipeline, else host staging.
        in_tick: a tick is open (`tracked_begin`), the scene goes behind the first su-problem without waiting."""
        scene = None
        if not self.rda_obstacle and self.device_obstacles and self.rda.has_scene:
            scene = self.rda.flatten_scene(obstacle_list)
        if scene is not None:
            upload = self.rda.upload_scene_async if in_tick else self.rda.upload_scene
            upload(scene, np.asarray(self.state, float)[0:2], self.obstacle_order)
        else:
            rda_obs = obstacle_list if self.rda_obstacle else self.convert_rda_obstacle(obstacle_list, self.state, self.obstacle_order)
            self.rda.upload_obstacles(rda_obs)

    def _sync_path(self, cur_ref_path):
        # keyed on content, not identity: the reference re-reads `ref_path` every tick (mpc.py:139-144), so a list that was
        # replaced or edited in place without update_ref_path must reach the device too (a few us for a few hundred waypoints)
        # Comparing all L waypoint arrays costs ~0.2 ms per tick for L = 800, as much as the whole device step.  Per tick: the list
        # object, its length and the CONTENT of the window the device tracker can reach in this tick (128 waypoints from the
        # current index, plus the last waypoint, which quirk Q12 rewrites) are compared with what was uploaded; every 32nd tick
        # everything is.  An in-place edit further ahead is therefore seen when the window reaches it, or within 32 ticks.
        key = self._dev_path_key
        L = len(cur_ref_path)
        self._path_ticks += 1
        if key is not None and key[0] is cur_ref_path and len(key[1]) == L and self._path_ticks % 32:
            i0 = max(0, min(self.cur_index, L) - 1)
            i1 = min(L, i0 + 128)
            if [p.tobytes() for p in cur_ref_path[i0:i1]] == key[1][i0:i1] and cur_ref_path[-1].tobytes() == key[1][-1]:
                return
        content = [p.tobytes() for p in cur_ref_path]
        if key is None or content != key[1]:
            self.rda.upload_path(cur_ref_path)
        self._dev_path_key = (cur_ref_path, content)

    def _nominal_u(self):
        """None when the device still holds cur_vel_array (the controls of the previous solve), else the array"""
        return None if self.cur_vel_array is self._dev_u else self.cur_vel_array

    def _tracked_done(self, cur_ref_path, u_opt_array, info, min_index, end_heading):
        self.cur_index = min_index
        cur_ref_path[-1][2, 0] = end_heading            # quirk Q12: the reference rewrites the last waypoint's heading in place
        if self._dev_path_key is not None and self._dev_path_key[0] is cur_ref_path:
            self._dev_path_key[1][-1] = cur_ref_path[-1].tobytes()     # ... and the device did the same to its copy
        out = self._end(cur_ref_path, u_opt_array, info)
        self._dev_u = None if info["arrive"] else self.cur_vel_array
        return out

    def _control_tracked(self, state, ref_speed, obstacle_list, **kwargs):
        cur_ref_path, gear_flag = self._piece(state)
        self._sync_path(cur_ref_path)
        if self.rda.has_pipeline:
            # pre_process and the first su-problem do not read this tick's obstacles (the first su-problem works with the
            # products of the previous step, reference quirk Q4): they run on the device while the obstacle objects are
            # flattened and staged here.  Same kernels, same order of dependent work, identical results.
            self.rda.tracked_begin(self.state, gear_flag * ref_speed, self.cur_index, self._nominal_u(), **kwargs)
            try:
                self._stage_obstacles(obstacle_list, in_tick=True)
            except BaseException:
                self.rda.tracked_finish(discard=True)           # close the tick before reporting the caller's error
                raise
            u_opt_array, info, min_index, end_heading = self.rda.tracked_finish()
        else:
            self._stage_obstacles(obstacle_list)
            u_opt_array, info, min_index, end_heading = self.rda.iterative_solve_tracked(
                self.state, gear_flag * ref_speed, self.cur_index, self._nominal_u(), **kwargs)
        return self._tracked_done(cur_ref_path, u_opt_array, info, min_index, end_heading)

    def _begin(self, state, ref_speed, **kwargs):
        """first half of `control` (mpc.py:127-147): the piece of the path in force, the signed reference speed, the
        nominal roll-out and the reference samples.  Shared with `Fleet.control`."""
        if np.shape(state)[0] > 3:
            state = state[0:3]
        self.state = state
        if self.enable_reverse:
            cur_ref_path = self.curve_list[self.curve_index]
            gear_flag = cur_ref_path[0][-1, 0]
        else:
            cur_ref_path = self.ref_path
            gear_flag = 1
        state_pre_array, ref_traj_list, self.cur_index = self.pre_process(
            state, cur_ref_path, self.cur_index, ref_speed, **kwargs)
        return cur_ref_path, gear_flag * ref_speed, state_pre_array, ref_traj_list

    def _end(self, cur_ref_path, u_opt_array, info):
        """second half of `control`, mpc.py:166-187: at the end of the (current piece of the) path the next gear piece
        starts when reverse is enabled; the last piece (or a plain path) stops the robot"""
        at_end = self.cur_index >= len(cur_ref_path) - self.goal_index_threshold
        last_piece = True
        if at_end and self.enable_reverse:
            self.curve_index, self.cur_index = self.curve_index + 1, 0
            last_piece = self.curve_index >= len(self.curve_list)
        info["arrive"] = bool(at_end and last_piece)
        if info["arrive"]:
            u_opt_array = np.zeros((2, self.receding))
        self.cur_vel_array = u_opt_array
        return u_opt_array[:, 0:1], info

    # ------------------------------------------------------------------ obstacles (mpc.py:189-218)
    def convert_rda_obstacle(self, obstacle_list, state=None, obstacle_order=False):
        out = []
        for obs in obstacle_list:
            if obs.cone_type == "norm2":
                A, b = self.convert_inequal_circle(obs.center, obs.radius, obs.velocity)
                out.append(rdaobs(A, b, obs.cone_type, obs.center, None))
            elif obs.cone_type == "Rpositive":
                A, b = self.convert_inequal_polygon(obs.vertex, obs.velocity)
                out.append(rdaobs(A, b, obs.cone_type, None, obs.vertex))
        if obstacle_order:
            out.sort(key=self.rda_obs_distance)
        return out

    def rda_obs_distance(self, rda_obs):
        if rda_obs.cone_type == "norm2":
            return MPC.distance(self.state[0:2], rda_obs.center[0:2])
        return float(np.min(np.linalg.norm(self.state[0:2] - rda_obs.vertex, axis=0)))

    def update_ref_path(self, ref_path):
        self.ref_path = ref_path
        self.cur_index = 0
        self._dev_path_key = None
        if self.enable_reverse:
            self.curve_list = self.split_path(self.ref_path)
            self.curve_index = 0

    def update_parameter(self, **kwargs):
        self.rda.assign_adjust_parameter(**kwargs)

    def split_path(self, ref_path):
        """split the path where the gear flag (last row) flips - mpc.py:232-249"""
        pieces, start, flag = [], 0, ref_path[0][-1, 0]
        for i, pt in enumerate(ref_path):
            if pt[-1, 0] != flag:
                pieces.append(ref_path[start:i])
                start, flag = i, pt[-1, 0]
        pieces.append(ref_path[start:])
        return pieces

    # ------------------------------------------------------------------ nominal rollout (mpc.py:251-291)
    def pre_process(self, state, ref_path, cur_index, ref_speed, **kwargs):
        _, min_index = self.closest_point(state, ref_path, cur_index, **kwargs)
        cur_state = state
        traj_point = ref_path[min_index]
        ref_traj_list = [traj_point]
        state_pre_list = [cur_state]
        step = {"acker": lambda s, v: self.motion_predict_model_acker(s, v, self.L, self.dt),
                "diff": lambda s, v: self.motion_predict_model_diff(s, v, self.dt),
                "omni": lambda s, v: self.motion_predict_model_omni(s, v, self.dt)}[self.dynamics]
        move_len = ref_speed * self.dt
        for i in range(self.receding):
            cur_state = step(cur_state, self.cur_vel_array[:, i:i + 1])
            state_pre_list.append(cur_state)
            traj_point, cur_index = self.inter_point(traj_point, ref_path, cur_index, move_len)
            # heading of the reference is unwrapped against the predicted heading (in place, Q12)
            traj_point[2, 0] = cur_state[2, 0] + MPC.wraptopi(traj_point[2, 0] - cur_state[2, 0])
            ref_traj_list.append(traj_point)
        return np.hstack(state_pre_list), ref_traj_list, min_index

    def motion_predict_model_acker(self, car_state, vel, wheel_base, sample_time):
        assert car_state.shape == (3, 1) and vel.shape == (2, 1)
        phi, v, psi = car_state[2, 0], vel[0, 0], vel[1, 0]
        return car_state + sample_time * np.array([[v * cos(phi)], [v * sin(phi)], [v * tan(psi) / wheel_base]])

    def motion_predict_model_diff(self, robot_state, vel, sample_time):
        assert robot_state.shape == (3, 1) and vel.shape == (2, 1)
        phi, v, w = robot_state[2, 0], vel[0, 0], vel[1, 0]
        return robot_state + sample_time * np.array([[v * cos(phi)], [v * sin(phi)], [w]])

    def motion_predict_model_omni(self, robot_state, vel, sample_time):
        assert robot_state.shape[0] >= 2 and vel.shape == (2, 1)
        return robot_state + sample_time * np.array([[vel[0, 0] * cos(vel[1, 0])], [vel[0, 0] * sin(vel[1, 0])], [0]])

    def closest_point(self, state, ref_path, start_ind, threshold=0.1, ind_range=10, **kwargs):
        min_dis, min_ind = inf, start_ind
        for i, wp in enumerate(ref_path[start_ind:start_ind + ind_range]):
            dis = MPC.distance(state[0:2], wp[0:2])
            if dis < min_dis:
                min_dis, min_ind = dis, start_ind + i
                if dis < threshold:
                    break
        return min_dis, min_ind

    def inter_point(self, traj_point, ref_path, cur_ind, length):
        """point at arc distance `length` ahead of traj_point along the polyline - mpc.py:355-383.
        At the path end the LAST WAYPOINT OBJECT itself is returned (quirk Q12)."""
        centre = np.squeeze(traj_point[0:2])
        new_point = np.copy(traj_point)
        while True:
            if cur_ind + 1 > len(ref_path) - 1:
                end_point = ref_path[-1]
                end_point[2] = MPC.wraptopi(end_point[2])
                return end_point, cur_ind
            a, b = ref_path[cur_ind], ref_path[cur_ind + 1]
            hit = self.range_cir_seg(centre, length, [np.squeeze(a[0:2]), np.squeeze(b[0:2])])
            if hit is None:
                cur_ind += 1
                continue
            half = MPC.wraptopi(b[2, 0] - a[2, 0]) / 2
            new_point[0:2, 0] = hit
            new_point[2, 0] = MPC.wraptopi(a[2, 0] + half)
            return new_point, cur_ind

    def range_cir_seg(self, circle, r, segment):
        assert circle.shape == (2,) and segment[0].shape == (2,) and segment[1].shape == (2,)
        sp, ep = segment
        d = ep - sp
        if np.linalg.norm(d) == 0:
            return None
        f = sp - circle
        qa, qb, qc = d @ d, 2 * f @ d, f @ f - r ** 2
        disc = qb ** 2 - 4 * qa * qc
        if disc < 0:
            return None
        t2 = (-qb + sqrt(disc)) / (2 * qa)
        if 0 <= t2 <= 1:
            return sp + t2 * d
        return None

    @staticmethod
    def distance(point1, point2):
        dx, dy = point1[0, 0] - point2[0, 0], point1[1, 0] - point2[1, 0]
        return sqrt(dx ** 2 + dy ** 2)

    @staticmethod
    def wraptopi(radian):
        # shift by whole turns until inside [-pi, pi]; same subtraction / addition sequence as mpc.py:425-433
        while abs(radian) > pi:
            radian = radian - 2 * pi if radian > 0 else radian + 2 * pi
        return radian

    # ------------------------------------------------------------------ geometry (mpc.py:440-549)
    def convert_inequal_circle(self, center, radius, velocity=np.zeros((2, 1))):
        eye = np.array([[1, 0], [0, 1], [0, 0]])
        tail = -radius * np.ones((1, 1))
        if np.linalg.norm(velocity) <= 0.01:
            return eye, np.vstack((center, tail))
        steps = range(self.receding + 1)                   # constant-velocity prediction over the horizon
        return [eye.copy() for _ in steps], [np.vstack((center + velocity * (t * self.dt), tail)) for t in steps]

    def convert_inequal_polygon(self, vertex, velocity=np.zeros((2, 1))):
        if np.linalg.norm(velocity) <= 0.01:
            return self.gen_inequal_global(vertex)
        pairs = [self.gen_inequal_global(vertex + velocity * (t * self.dt)) for t in range(self.receding + 1)]
        return [p[0] for p in pairs], [p[1] for p in pairs]

    def gen_inequal_global(self, vertex):
        """half-space form of a convex polygon, un-normalised edge normals (quirk Q11)"""
        convex_flag, order = self.is_convex_and_ordered(vertex)
        if not convex_flag:
            print(f"Warning: The polygon constructed by vertex is not convex. Please check the vertex: {vertex}")
        if order == "CW":
            vertex = vertex[:, ::-1]
        cur = vertex[0:2, :]
        nxt = np.roll(cur, -1, axis=1)
        edge = nxt - cur
        A = np.stack((edge[1], -edge[0]), axis=1)
        b = np.sum(A * cur.T, axis=1, keepdims=True)
        return A, b

    def cross_product(self, o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    def is_convex_and_ordered(self, points):
        """(convex?, 'CCW' | 'CW' | None): all non-zero turns of consecutive vertex triples share one sign; the sign of
        the first one names the order (all-collinear input reports 'CW', like mpc.py:527-549)"""
        n = points.shape[1]
        if n < 3:
            return False, None
        o, a, b = points[0:2], np.roll(points[0:2], -1, axis=1), np.roll(points[0:2], -2, axis=1)
        turns = (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
        turns = turns[turns != 0]
        if turns.size and not (np.all(turns > 0) or np.all(turns < 0)):
            return False, None
        return True, "CCW" if turns.size and turns[0] > 0 else "CW"

    def get_adjust_parameters(self):
        return self.rda.get_adjust_parameter()

    def no_ref_path(self):
        return len(self.ref_path) == 0

    def reset(self):
        self.cur_vel_array = np.zeros((2, self.receding))
        self.cur_index = 0
        self.curve_index = 0
        self.rda.reset()
