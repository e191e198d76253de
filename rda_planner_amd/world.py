"""
Headless stand-in for the slice of ir-sim the reference's example scripts use: the YAML world files
under example/*/*.yaml (`world`, `robot`, `obstacle` blocks), a kinematic step, collision / arrival checks.  It lets
the example loops (e.g. example/path_track/path_track.py:24-42) run without a simulator:

    import rda_planner_amd.world as irsim          # instead of `import irsim`
    env = irsim.make("path_track_diff.yaml")
    robot_info = env.get_robot_info()              # .G .h .cone_type .wheelbase .shape
    ... mpc.control(env.robot.state, speed, env.get_obstacle_info_list()) ; env.step(vel) ; env.done()

Supported YAML subset: world.step_time; one robot with kinematics acker|diff|omni and a rectangle shape; obstacle
groups with `number`, distribution manual (`state` list) or random (`range_low/high`, seeded), shapes circle / polygon
(`vertices`, absolute) / rectangle, optional `kinematics` + `vel_max` (then the obstacle moves with a seeded constant
velocity and is reflected at the world border - a simplification of ir-sim's `dash`/`wander` behaviours).
A `lidar2d` sensor of the robot is simulated by exact ray casting (`get_lidar_scan`).
Rendering calls are accepted and ignored.
"""
from types import SimpleNamespace

import numpy as np
import yaml

from . import scenarios as sc


class _Obstacle:
    """what MPC.convert_rda_obstacle reads (reference mpc.py:192-203)"""

    def __init__(self, kind, center, radius, vertex, velocity):
        self.cone_type = "norm2" if kind == "circle" else "Rpositive"
        self.center = None if center is None else np.asarray(center, float).reshape(2, 1)
        self.radius = radius
        self.vertex = vertex
        self.velocity = np.asarray(velocity, float).reshape(2, 1)

    def advance(self, dt, lo, hi):
        if not self.velocity.any():
            return
        shift = self.velocity * dt
        ref = self.center if self.center is not None else self.vertex.mean(axis=1, keepdims=True)
        for k in range(2):                                     # reflect at the world border
            if not lo[k] <= ref[k, 0] + shift[k, 0] <= hi[k]:
                self.velocity[k, 0] = -self.velocity[k, 0]
                shift[k, 0] = -shift[k, 0]
        if self.center is not None:
            self.center = self.center + shift
        else:
            self.vertex = self.vertex + shift


class World:
    def __init__(self, cfg, seed=sc.SEED):
        w = cfg.get("world", {})
        self.step_time = float(w.get("step_time", 0.1))
        off = w.get("offset", [0, 0])
        self.lo = np.array(off[0:2], float)
        self.hi = self.lo + np.array([w.get("width", 50), w.get("height", 50)], float)
        rng = np.random.default_rng(seed)
        rb = cfg["robot"][0] if isinstance(cfg["robot"], list) else cfg["robot"]
        shape = rb.get("shape", {})
        if shape.get("name", "rectangle") != "rectangle":
            raise ValueError("only rectangle robots are supported (cone 'Rpositive')")
        self.dynamics = rb.get("kinematics", {}).get("name", "diff")
        wb = float(shape.get("wheelbase", 0) or 0)
        self._car = sc.rectangle_robot(float(shape.get("length", 4.6)), float(shape.get("width", 1.6)), wb, self.dynamics,
                                       max_speed=tuple(rb.get("vel_max", (10, 1))))
        self.robot = SimpleNamespace(state=np.asarray(rb.get("state", [0, 0, 0]), float)[0:3].reshape(3, 1).copy(),
                                     goal=np.asarray(rb.get("goal", [0, 0, 0]), float)[0:3].reshape(3, 1),
                                     goal_threshold=float(rb.get("goal_threshold", 0.3)))
        self.obstacles = []
        for grp in cfg.get("obstacle", []) or []:
            num = int(grp.get("number", 1))
            shapes = grp.get("shape", [{"name": "circle", "radius": 0.5}])
            shapes = shapes if isinstance(shapes, list) else [shapes]
            dist = grp.get("distribution", {"name": "manual"})
            states = [list(s) for s in grp.get("state", [])]
            moving = "kinematics" in grp
            vmax = float(np.abs(np.asarray(grp.get("vel_max", [1.0, 0.0]), float))[0])
            for i in range(num):
                shp = shapes[min(i, len(shapes) - 1)]
                if dist.get("name") == "random":
                    lo_, hi_ = np.asarray(dist.get("range_low", [0, 0, 0]), float), np.asarray(dist.get("range_high", [10, 10, 0]), float)
                    st = rng.uniform(lo_, hi_)
                else:
                    st = np.asarray(states[min(i, len(states) - 1)] if states else [0, 0, 0], float)
                st = np.concatenate([st, np.zeros(3)])[0:3]
                vel = np.zeros(2)
                if moving:
                    ang = rng.uniform(-np.pi, np.pi)
                    vel = rng.uniform(0.3, 1.0) * vmax * np.array([np.cos(ang), np.sin(ang)])
                name = shp.get("name", "circle")
                if name == "circle":
                    self.obstacles.append(_Obstacle("circle", st[0:2], float(shp.get("radius", 0.5)), None, vel))
                else:
                    if name == "polygon":
                        V = np.asarray(shp["vertices"], float).T
                    else:                                      # rectangle centred at the origin of its own frame
                        L, W = float(shp.get("length", 1.0)), float(shp.get("width", 1.0))
                        V = np.array([[-L / 2, L / 2, L / 2, -L / 2], [-W / 2, -W / 2, W / 2, W / 2]])
                    c, s = np.cos(st[2]), np.sin(st[2])
                    V = np.array([[c, -s], [s, c]]) @ V + st[0:2].reshape(2, 1)
                    self.obstacles.append(_Obstacle("polygon", None, None, V, vel))
        self.collided = False
        self.arrived = False
        # first lidar2d sensor of the robot (ir-sim: beams over [-angle_range/2, angle_range/2] about the heading)
        self.lidar = None
        for sen in rb.get("sensors", []) or []:
            if sen.get("type", sen.get("name")) == "lidar2d":
                half = 0.5 * float(sen.get("angle_range", np.pi))
                self.lidar = SimpleNamespace(range_min=float(sen.get("range_min", 0.0)), range_max=float(sen.get("range_max", 10.0)),
                                             angle_min=-half, angle_max=half, number=int(sen.get("number", 100)))
                break

    # ---- the calls the example scripts make ---------------------------------------------------------------------
    def get_robot_info(self):
        c = self._car
        return SimpleNamespace(G=c.G, h=c.h, cone_type=c.cone_type, wheelbase=c.wheelbase,
                               shape=[None, None, c.wheelbase, None], max_speed=c.max_speed)

    def get_obstacle_info_list(self):
        return list(self.obstacles)

    def get_lidar_scan(self):
        """ranges of the lidar beams against the current obstacles (noise-free), in the dict layout of ir-sim's scan"""
        ld = self.lidar
        if ld is None:
            raise RuntimeError("the robot of this world has no lidar2d sensor")
        ang = np.linspace(ld.angle_min, ld.angle_max, ld.number)
        th = self.robot.state[2, 0] + ang
        o = self.robot.state[0:2, 0]
        d = np.stack((np.cos(th), np.sin(th)), axis=1)                      # (beams, 2) unit directions
        rng = np.full(ld.number, ld.range_max)
        for ob in self.obstacles:
            if ob.center is not None:                                        # ray / circle
                f = o - ob.center[:, 0]
                b = d @ f
                disc = b * b - (f @ f - ob.radius ** 2)
                ok = disc >= 0
                t = np.where(ok, -b - np.sqrt(np.where(ok, disc, 0.0)), np.inf)
                t = np.where(t >= 0, t, np.inf)
                rng = np.minimum(rng, t)
            else:                                                            # ray / polygon edges
                V = ob.vertex
                for k in range(V.shape[1]):
                    p, q = V[:, k], V[:, (k + 1) % V.shape[1]]
                    e = q - p
                    den = d[:, 0] * e[1] - d[:, 1] * e[0]
                    w = p - o
                    with np.errstate(divide="ignore", invalid="ignore"):
                        t = (w[0] * e[1] - w[1] * e[0]) / den               # along the beam
                        s_ = (w[0] * d[:, 1] - w[1] * d[:, 0]) / den         # along the edge
                    ok = (np.abs(den) > 1e-12) & (t >= 0) & (s_ >= 0) & (s_ <= 1)
                    rng = np.minimum(rng, np.where(ok, t, np.inf))
        rng = np.clip(rng, ld.range_min, ld.range_max)
        return {"ranges": rng, "angle_min": ld.angle_min, "angle_max": ld.angle_max, "range_min": ld.range_min,
                "range_max": ld.range_max, "angle_increment": (ld.angle_max - ld.angle_min) / max(ld.number - 1, 1)}

    def draw_box(self, *a, **k):
        pass

    def step(self, vel):
        self.robot.state = sc.kinematic_step(self.robot.state, np.asarray(vel, float).reshape(2, 1), self._car, self.step_time)
        for o in self.obstacles:
            o.advance(self.step_time, self.lo, self.hi)
        self.collided = sc.clearance(self._car, self.robot.state, self.obstacles) <= 0.0
        self.arrived = float(np.linalg.norm(self.robot.state[0:2] - self.robot.goal[0:2])) <= self.robot.goal_threshold

    def done(self):
        return self.collided or self.arrived

    def clearance(self):
        return sc.clearance(self._car, self.robot.state, self.obstacles)

    def render(self, *a, **k):
        pass

    def draw_trajectory(self, *a, **k):
        pass

    def end(self, *a, **k):
        pass


def make(world_name, seed=sc.SEED, **_ignored):
    """`irsim.make('file.yaml', ...)`; display / animation keywords are accepted and ignored"""
    with open(world_name) as f:
        return World(yaml.safe_load(f), seed=seed)
