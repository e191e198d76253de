"""
`Fleet` - B independent `MPC` planners stepped together (BASELINE config C5, "batched multi-ego").

The reference has one `MPC` / `RDA_solver` object per robot and a multi-robot user calls `control` in a loop, one
CVXPY solve after the other (mpc.py:127-187).  Here the members are still ordinary `MPC` objects - own path, state,
obstacles, weights - but `Fleet.control` issues their ADMM iterations as ONE set of kernel launches with an ego
dimension in the grid (include/rda_hip.h, `rda_fleet_*`): the su-problems of all members run side by side on B compute
units and the LamMuZ grid is B times larger, which is what a 256-CU part needs.  Every member's result is identical to
what `member.control(...)` would have returned.
"""
import ctypes as C
import time

import numpy as np

from ._capi import Info, dptr, iptr, f64


class Fleet:
    def __init__(self, members):
        """members: `MPC` objects with equal receding / max_obs_num / max_edge_num / iter_num and robots with the same
        number of edges (anything else may differ)."""
        self.members = list(members)
        if not self.members:
            raise ValueError("a fleet needs at least one member")
        self.api = self.members[0].rda._be.api
        if not getattr(self.api, "has_fleet", False):
            raise RuntimeError("the loaded solver library has no fleet entry points")
        B = len(self.members)
        arr = (C.c_void_p * B)(*[m.rda._be.handle for m in self.members])
        self._handle = C.c_void_p()
        rc = self.api.fleet_create(arr, B, C.byref(self._handle))
        if rc != 0:
            raise RuntimeError(f"rda_fleet_create failed with code {rc} (members must agree on T, N, E, R, iter_num)")
        T = self.members[0].receding
        self._in_s, self._in_u = np.zeros((B, 3, T + 1)), np.zeros((B, 2, T))
        self._ref, self._speed = np.zeros((B, 3, T + 1)), np.zeros(B)
        self._out_u, self._out_s = np.zeros((B, 2, T)), np.zeros((B, 3, T + 1))
        self._info = (Info * B)()
        self.batched_ticks = 0          # ticks whose scenes went through the one-pass staging (diagnostics)

    def __len__(self):
        return len(self.members)

    def close(self):
        if self._handle:
            self.api.fleet_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def control(self, states, ref_speeds, obstacle_lists, **kwargs):
        """one MPC step of every member: `states[i]`, `ref_speeds[i]` (a scalar is shared) and `obstacle_lists[i]` are
        what `members[i].control` takes; returns the list of its `(u, info)` results."""
        B, start = len(self.members), time.time()
        if np.isscalar(ref_speeds):
            ref_speeds = [ref_speeds] * B
        if hasattr(self.api.lib, "rda_fleet_step_tracked") and all(m._tracks(kwargs) for m in self.members):
            return self._control_tracked(states, ref_speeds, obstacle_lists, start, **kwargs)
        begun = []
        for i, m in enumerate(self.members):
            cur_ref_path, speed, nom_s, ref_list = m._begin(states[i], ref_speeds[i], **kwargs)
            T = m.receding
            self._in_s[i] = f64(nom_s, (3, T + 1))
            self._in_u[i] = f64(m.cur_vel_array, (2, T))
            self._ref[i] = np.hstack(ref_list)[0:3, :]
            self._speed[i] = speed
            obstacles = obstacle_lists[i]
            scene = None
            if not m.rda_obstacle and m.device_obstacles and m.rda.has_scene:
                scene = m.rda.flatten_scene(obstacles)
            if scene is not None:
                m.rda.upload_scene(scene, np.asarray(m.state, float)[0:2], m.obstacle_order)
            else:
                rda_obs = obstacles if m.rda_obstacle else m.convert_rda_obstacle(obstacles, m.state, m.obstacle_order)
                m.rda.upload_obstacles(rda_obs)
            begun.append((cur_ref_path, ref_list))
        rc = self.api.fleet_step(self._handle, dptr(self._in_s), dptr(self._in_u), dptr(self._ref), dptr(self._speed),
                                 dptr(self._out_u), dptr(self._out_s), self._info)
        if rc < 0:
            raise RuntimeError(f"rda_fleet_step failed with code {rc}")
        out = []
        for i, m in enumerate(self.members):
            cur_ref_path, ref_list = begun[i]
            if self._info[i].su_status and m.rda.time_print:
                print("No update of state and control vector")        # reference rda_solver.py:699
            info = m.rda.pack_info(ref_list, self._out_s[i].copy(), self._info[i], start)
            out.append(m._end(cur_ref_path, self._out_u[i].copy(), info))
        return out

    def _stage_all(self, obstacle_lists, st):
        """all members' raw scenes flattened in ONE pass over the obstacle objects and staged by one library call
        (rda_fleet_upload_scenes, no waiting); False when some member needs the per-member route (host conversion, cone
        types the reference skips, too many vertices)"""
        ms = self.members
        if not getattr(self.api, "has_fleet_scenes", False):
            return False
        if any(m.rda_obstacle or not m.device_obstacles or not m.rda.has_scene for m in ms):
            return False
        counts = np.fromiter((len(ol) for ol in obstacle_lists), np.int32, len(ms))
        every = [o for ol in obstacle_lists for o in ol]
        scene = ms[0].rda.flatten_scene(every)
        if scene is None or scene[0] != len(every):
            return False
        _, kind, nvert, geom, vel = scene
        kind = np.ascontiguousarray(kind, np.int32); nvert = np.ascontiguousarray(nvert, np.int32)
        geom = f64(geom); vel = f64(vel)
        rob = np.ascontiguousarray(st[:, 0:2])
        order = np.fromiter((bool(m.obstacle_order) for m in ms), np.int32, len(ms))
        rc = self.api.fleet_upload_scenes(self._handle, iptr(counts), iptr(kind), iptr(nvert), dptr(geom), dptr(vel), dptr(rob), iptr(order))
        if rc < 0:
            raise RuntimeError(f"rda_fleet_upload_scenes failed with code {rc}")
        self.batched_ticks += 1
        return True

    def _control_tracked(self, states, ref_speeds, obstacle_lists, start, threshold=0.1, ind_range=10):
        """the same with every member's pre_process on the device (rda_fleet_step_tracked): per ego only the state, the
        signed speed and the path index travel"""
        B, T = len(self.members), self.members[0].receding
        st, cur = np.zeros((B, 3)), np.zeros(B, np.int32)
        pieces, resident = [], True
        for i, m in enumerate(self.members):
            cur_ref_path, gear = m._piece(states[i])
            m._sync_path(cur_ref_path)
            st[i] = np.asarray(m.state, float).ravel()[0:3]
            self._speed[i] = gear * ref_speeds[i]
            cur[i] = m.cur_index
            resident = resident and m._nominal_u() is None
            pieces.append(cur_ref_path)
        if not self._stage_all(obstacle_lists, st):
            for i, m in enumerate(self.members):
                m._stage_obstacles(obstacle_lists[i])
        nom_u = None
        if not resident:            # some member's cur_vel_array was replaced since its last solve: send them all
            for i, m in enumerate(self.members):
                self._in_u[i] = f64(m.cur_vel_array, (2, T))
            nom_u = self._in_u
        mi, eh = np.zeros(B, np.int32), np.zeros(B)
        rc = self.api.fleet_step_tracked(self._handle, dptr(st), dptr(self._speed), iptr(cur), float(threshold), int(ind_range),
                                         dptr(nom_u), dptr(self._out_u), dptr(self._out_s), self._info, dptr(self._ref),
                                         iptr(mi), dptr(eh))
        if rc < 0:
            raise RuntimeError(f"rda_fleet_step_tracked failed with code {rc}")
        out = []
        for i, m in enumerate(self.members):
            if self._info[i].su_status and m.rda.time_print:
                print("No update of state and control vector")        # reference rda_solver.py:699
            ref_own = self._ref[i].copy()
            ref_list = [ref_own[:, j:j + 1] for j in range(T + 1)]
            info = m.rda.pack_info(ref_list, self._out_s[i].copy(), self._info[i], start)
            out.append(m._tracked_done(pieces[i], self._out_u[i].copy(), info, int(mi[i]), float(eh[i])))
        return out
