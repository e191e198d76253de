"""Generates tests/golden/ref_*.npz by RUNNING THE REFERENCE (unmodified /root/reference/RDA_planner/{rda_solver,mpc}.py on the
cvxpy / pathos stand-ins of oracle/refshim - see oracle/ref_harness.py).  /root/reference exists in the build container only;
the fixtures travel, so the `-m gpu` tests can compare the HIP path with reference output on the GPU box.

    python tests/golden/make_ref_golden.py [--backend refshim|cvxpy] [--only-na]

--backend refshim (default; the only one possible in the build image): the fixtures named below.
--backend cvxpy   (the day a CVXPY 1.5.2 + ECOS wheel is installable, /root/reference/setup.py:8): the SAME script on the real solver stack,
                  writing ref_*_cvxpy.npz BESIDE the shim fixtures (nothing is overwritten).  The plumbing loops then run with the reference's
                  own prob.solve() - no oracle answers injected - so everything non-unique in them (lam, mu, z individually, and whatever later
                  iterations make of them) is ECOS's; tests/test_ref_golden.py::test_cvxpy_fixtures_agree_with_the_shim_fixtures_where_the_answer_is_unique
                  compares the unique quantities across the two sets when both exist.  The requested backend must be the one that imports: asking
                  for cvxpy while only the stand-in is importable is an error, not a silent fallback.

ref_plumbing.npz  - mode "oracle": closed loops of the reference `mpc.MPC` + `RDA_solver` whose `prob.solve()` calls are answered
                    by the oracle's two argmins (cold).  Per MPC step the solver inputs (nominal, reference, the obstacle list the
                    reference's own `convert_rda_obstacle` produced, staged as dense arrays) and, after EVERY ADMM iteration, the
                    reference's parameter values: nominal s, u, `dis`, residuals in full; lam, mu, z, xi, zeta, obsA_lam, obsb_lam
                    in full for the small scenes and as 8 fixed random projections each for the large one.
ref_problems_na.npz - the same for the NON-accelerated LamMuZ cost (accelerated=False, rda_solver.py:399-402)
ref_problems.npz  - mode "ipm": LamMuZ and su problems BUILT BY THE REFERENCE's construction code and solved as they stand by the
                    generic interior-point stand-in: inputs in the layout of rda_lammuz_batch / rda_su_solve and the unique part
                    of the answers (LamMuZ: optimal value, min(Im, 0), Hm; su: s, u, d).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SUFFIX = ""          # "_cvxpy" for fixtures written on the real solver stack
INJECT_ORACLE = True  # plumbing: answer prob.solve() with the oracle's argmins (shim backend)
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from helpers import random_polygon                        # noqa: E402
from oracle import ref_harness as rh                      # noqa: E402
from oracle.oracle_backend import api as orc_api          # noqa: E402
from rda_planner_amd import scenarios as sc               # noqa: E402
from rda_planner_amd.rda_solver import RDA_solver         # noqa: E402

KEYS = ("lam", "mu", "z", "xi", "zeta", "a_lam", "b_lam")
NPROJ = 8


def projections(shape, seed):
    """fixed random weight vectors for the digest of a large array (tests regenerate them from the seed)"""
    return np.random.default_rng(seed).standard_normal((NPROJ, int(np.prod(shape))))


def plumbing(rs, mp, orc):
    out = {}
    car_d = sc.rectangle_robot(dynamics="diff", wheelbase=0)
    car_a = sc.rectangle_robot(dynamics="acker")
    car_o = sc.rectangle_robot(dynamics="omni", wheelbase=0)
    line = sc.line_path([4, 25, 0], [60, 25, 0], 0.1)
    clear = np.array([[p[0, 0], p[1, 0]] for p in line[::10]])
    moving = sc.scene_polygons(12, lo=(6, 18), hi=(30, 32), moving=True, keep_clear=clear, clear_radius=3.5)
    ns_obs = sc.scene_polygons(200, lo=(8, 10), hi=(56, 40), seed=sc.SEED, keep_clear=clear, clear_radius=3.2)
    scenes = {
        # BASELINE C1: the literal path_track scene, diff drive, T=10, iter_num=2, ro1=300, re-sorted every step, 11 slots
        "c1": dict(car=car_d, path=sc.path_track_ref(), obs=lambda k: sc.scene_path_track(), steps=10, full=True,
                   kw=dict(receding=10, iter_num=2, max_edge_num=4, max_obs_num=11, ro1=300, obstacle_order=True)),
        # padding (7 obstacles in 11 slots, Q3) and spare edge rows (max_edge_num=5)
        "pad": dict(car=car_d, path=sc.path_track_ref(), obs=lambda k: sc.scene_path_track()[4:], steps=4, full=True,
                    kw=dict(receding=10, iter_num=3, max_edge_num=5, max_obs_num=11, ro1=300, obstacle_order=True)),
        # BASELINE C2: corridor, Ackermann, T=20, 20 obstacles
        "c2": dict(car=car_a, path=sc.line_path([0, 20, 0], [60, 20, 0], 0.1), obs=lambda k: sc.scene_corridor(), steps=5, full=True,
                   kw=dict(receding=20, iter_num=3, max_edge_num=4, max_obs_num=20, obstacle_order=True)),
        # C4-shaped: moving polygons advancing every tick (per-stage A, b lists), omni, an empty list on tick 3 (Q9)
        "c4": dict(car=car_o, path=line, obs=lambda k: [] if k == 3 else [o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) for o in moving],
                   steps=6, full=True, kw=dict(receding=12, iter_num=3, max_edge_num=4, max_obs_num=12, obstacle_order=True)),
        # north-star size: T=20, N=200 static polygons - digests only
        "ns": dict(car=car_a, path=line, obs=lambda k: ns_obs, steps=3, full=False,
                   kw=dict(receding=20, iter_num=4, max_edge_num=4, max_obs_num=200, ro1=200, obstacle_order=True)),
    }
    for name, S in scenes.items():
        kw, car_t, path = S["kw"], S["car"], S["path"]
        T = kw["receding"]
        rmpc = mp.MPC(car_t, [p.copy() for p in path], process_num=1, time_print=False, **kw)
        if INJECT_ORACLE:
            rh.OracleAnswers(rmpc.rda, rs, orc)
        log = rh.record_iterations(rmpc.rda)
        stager = RDA_solver.__new__(RDA_solver)
        stager.max_obs_num, stager.max_edge_num, stager.T = kw["max_obs_num"], kw["max_edge_num"], T
        cap = {}
        orig = rmpc.rda.iterative_solve

        def spy(nom_s, nom_u, ref_states, ref_speed, obstacle_list, _cap=cap, _orig=orig, **k):
            _cap.update(nom_s=np.array(nom_s, float), nom_u=np.array(nom_u, float), ref=np.array(np.hstack(ref_states)[0:3, :], float),
                        speed=float(ref_speed), obs=list(obstacle_list))
            return _orig(nom_s, nom_u, ref_states, ref_speed, obstacle_list, **k)
        rmpc.rda.iterative_solve = spy
        state = path[0].copy().reshape(3, 1)
        out[f"{name}.steps"] = S["steps"]
        out[f"{name}.cfg"] = np.array([T, kw["max_obs_num"], kw["max_edge_num"], kw["iter_num"], kw.get("ro1", 200),
                                       {"acker": 0, "diff": 1, "omni": 2}[car_t.dynamics]], float)
        for k in range(S["steps"]):
            del log[:]
            u, info = rmpc.control(state, 4.0, list(S["obs"](k)))
            n, A, b, cone, per_t = stager._stage(list(cap["obs"]))
            pre = f"{name}.{k}"
            out[f"{pre}.nom_s"], out[f"{pre}.nom_u"], out[f"{pre}.ref"] = cap["nom_s"], cap["nom_u"], cap["ref"]
            out[f"{pre}.speed"], out[f"{pre}.n_obs"], out[f"{pre}.per_t"] = cap["speed"], n, per_t
            if n:
                out[f"{pre}.A"], out[f"{pre}.b"], out[f"{pre}.cone"] = A, b, cone
            out[f"{pre}.iters"] = len(log)
            out[f"{pre}.u_applied"] = u
            for it, snap in enumerate(log):
                q = f"{pre}.it{it}"
                out[f"{q}.s"], out[f"{q}.u"], out[f"{q}.dis"] = snap["s"], snap["u"], snap["dis"]
                out[f"{q}.resi"] = np.array([snap["resi_dual"], snap["resi_pri"]])
                for key in KEYS:
                    v = snap[key][:, 1:] if key in ("lam", "mu", "xi", "a_lam", "b_lam") else snap[key]      # column 0 is free / unused
                    out[f"{q}.{key}"] = v if S["full"] else projections(v.shape, 7) @ v.ravel()
            state = sc.kinematic_step(state, u, car_t, 0.1)
        print(f"plumbing {name}: {S['steps']} steps recorded")
    np.savez_compressed(os.path.join(HERE, "ref_plumbing" + SUFFIX + ".npz"), **out)


def problems(rs, mp):
    out = {}
    rng = np.random.default_rng(20250509)
    car_t = sc.rectangle_robot(dynamics="acker")
    G, h = np.ascontiguousarray(car_t.G, float), np.ascontiguousarray(car_t.h, float).ravel()
    # ---- LamMuZ: T stages of N obstacles, solved per obstacle by the stand-in; stored per (obstacle, stage)
    T, N, E = 5, 6, 4
    rows = {k: [] for k in ("A", "b", "cone", "p", "phi", "xi", "zeta", "dbar", "cost", "mneg", "H")}
    for accelerated in (True,):
        r = rs.RDA_solver(T, car_t, max_edge_num=E, max_obs_num=N, iter_num=2, step_time=0.1, process_num=1, time_print=False, ro2=1.0,
                          accelerated=accelerated)
        for trial in range(4):
            nom_u = np.vstack([rng.uniform(1, 4, T), rng.uniform(-0.3, 0.3, T)])
            nom_s = np.zeros((3, T + 1))
            nom_s[:, 0] = [rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-3, 3)]
            for t in range(T):
                nom_s[:, t + 1] = sc.kinematic_step(nom_s[:, t:t + 1], nom_u[:, t:t + 1], car_t, 0.1).ravel()
            dis = rng.uniform(0.1, 1.0, (1, T))
            obs = []
            for n in range(N):
                dist, th = rng.choice([1.0, 2.5, 4.0, 8.0, 20.0]), rng.uniform(0, 2 * np.pi)
                cen = nom_s[0:2, T // 2] + dist * np.array([np.cos(th), np.sin(th)])
                if rng.random() < 0.3:
                    obs.append(mp.rdaobs(np.array([[1, 0], [0, 1], [0, 0.0]]), np.array([[cen[0]], [cen[1]], [-rng.uniform(0.3, 1.5)]]), "norm2", None, None))
                else:
                    k = int(rng.integers(3, E + 1))
                    A_, b_ = random_polygon(rng, cen, k, rng.uniform(0.5, 2.0), k)
                    obs.append(mp.rdaobs(A_, b_.reshape(-1, 1), "Rpositive", None, None))
            r.assign_state_parameter(nom_s, nom_u, dis)
            r.assign_obstacle_parameter(obs)
            r.assign_combine_parameter_stateobs()
            for n in range(N):
                r.para_xi_list[n].value = np.vstack([np.zeros((1, 2)), rng.normal(0, rng.choice([0, 0.05, 0.5]), (T, 2))])
                r.para_zeta_list[n].value = rng.normal(0, rng.choice([0, 0.3, 2.0]), (1, T))
            for n in range(N):
                prob = r.prob_LamMuZ_list[n]
                prob.solve()
                if prob.status != "optimal":          # the stand-in stalled short of 1e-10 on this obstacle: not a fixture
                    continue
                Im, Hm = r.indep_Im_array_LamMuZ[n].value, r.indep_Hm_array_LamMuZ[n].value
                for t in range(T):
                    rows["A"].append(np.array(r.para_obstacle_list[n]["A"][t + 1].value, float))
                    rows["b"].append(np.array(r.para_obstacle_list[n]["b"][t + 1].value, float).ravel())
                    rows["cone"].append(int(r.para_obstacle_list[n]["cone_type"].value[1] > 0.5))
                    rows["p"].append(nom_s[0:2, t + 1].copy()); rows["phi"].append(nom_s[2, t])
                    rows["xi"].append(np.array(r.para_xi_list[n].value[t + 1], float)); rows["zeta"].append(float(r.para_zeta_list[n].value[0, t]))
                    rows["dbar"].append(float(dis[0, t]))
                    rows["cost"].append(0.5 * min(Im[t], 0.0) ** 2 + 0.5 * float(np.sum(Hm[t] ** 2)))
                    rows["mneg"].append(min(float(Im[t]), 0.0)); rows["H"].append(np.array(Hm[t], float))
    for k, v in rows.items():
        out[f"lmz.{k}"] = np.array(v)
    out["lmz.G"], out["lmz.h"] = G, h
    print("LamMuZ problems:", len(rows["cost"]), "sub-problems;", int(np.sum(np.array(rows["cost"]) > 1e-6)), "with a positive optimal value")
    # ---- su: the problem construct_su_prob builds, three kinematics
    k = 0
    for dyn in ("acker", "diff", "omni"):
        car_s = sc.rectangle_robot(dynamics=dyn, wheelbase=3.0 if dyn == "acker" else 0)
        T, N = 10, 6
        r = rs.RDA_solver(T, car_s, max_edge_num=4, max_obs_num=N, iter_num=2, step_time=0.1, process_num=1, time_print=False, ro1=200)
        for trial in range(2):
            nom_u = np.vstack([rng.uniform(1, 4, T), rng.uniform(-0.3, 0.3, T)])
            nom_s = np.zeros((3, T + 1))
            nom_s[:, 0] = [rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-3, 3)]
            for t in range(T):
                nom_s[:, t + 1] = sc.kinematic_step(nom_s[:, t:t + 1], nom_u[:, t:t + 1], car_s, 0.1).ravel()
            r.para_ref_s.value = nom_s + rng.normal(0, 0.3, (3, T + 1))
            r.para_ref_speed.value = 4.0
            r.assign_state_parameter(nom_s, nom_u, rng.uniform(0.1, 1.0, (1, T)))
            for n in range(N):
                a = rng.normal(0, 0.5, (T + 1, 2))
                a /= np.maximum(1, np.linalg.norm(a, axis=1, keepdims=True))
                r.para_obsA_lam_list[n].value = a
                r.para_obsb_lam_list[n].value = (np.einsum("tk,kt->t", a, nom_s[0:2, :]) - rng.uniform(-0.5, 1.5, T + 1)).reshape(T + 1, 1)
                r.para_mu_list[n].value = np.abs(rng.normal(0, 0.2, (4, T + 1)))
                r.para_z_list[n].value = np.abs(rng.normal(0, 0.2, (1, T)))
                r.para_zeta_list[n].value = rng.normal(0, 0.3, (1, T))
                r.para_xi_list[n].value = rng.normal(0, 0.3, (T + 1, 2))
            s_ref, u_ref, d_ref = r.su_prob_solve()
            assert r.prob_su.status == "optimal"
            inp = rh.su_inputs_from_reference(r)
            for key in ("nom_s", "nom_u", "ref", "a", "cc", "g", "d0"):
                out[f"su.{k}.{key}"] = inp[key]
            out[f"su.{k}.dyn"] = {"acker": 0, "diff": 1, "omni": 2}[dyn]
            out[f"su.{k}.s"], out[f"su.{k}.u"], out[f"su.{k}.d"] = np.array(s_ref, float), np.array(u_ref, float), np.array(d_ref, float).ravel()
            k += 1
    out["su.count"] = k
    print("su problems:", k)
    np.savez_compressed(os.path.join(HERE, "ref_problems" + SUFFIX + ".npz"), **out)



def lammuz_problems_nonaccelerated(rs, mp):
    """the NON-accelerated form of the LamMuZ cost (rda_solver.py:399-402: Im^2 instead of neg(Im)^2) as the reference's
    construction code builds it, solved by the stand-in: its own file and its own random stream, so that ref_problems.npz stays as it is.
    Here Im itself is unique (the cost is strictly convex in it), so it is stored in full."""
    out = {}
    rng = np.random.default_rng(20250510)
    car_t = sc.rectangle_robot(dynamics="acker")
    G, h = np.ascontiguousarray(car_t.G, float), np.ascontiguousarray(car_t.h, float).ravel()
    T, N, E = 5, 6, 4
    rows = {k: [] for k in ("A", "b", "cone", "p", "phi", "xi", "zeta", "dbar", "cost", "Im", "H")}
    r = rs.RDA_solver(T, car_t, max_edge_num=E, max_obs_num=N, iter_num=2, step_time=0.1, process_num=1, time_print=False, ro2=1.0,
                      accelerated=False)
    for trial in range(4):
        nom_u = np.vstack([rng.uniform(1, 4, T), rng.uniform(-0.3, 0.3, T)])
        nom_s = np.zeros((3, T + 1))
        nom_s[:, 0] = [rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-3, 3)]
        for t in range(T):
            nom_s[:, t + 1] = sc.kinematic_step(nom_s[:, t:t + 1], nom_u[:, t:t + 1], car_t, 0.1).ravel()
        dis = rng.uniform(0.1, 1.0, (1, T))
        obs = []
        for n in range(N):
            dist, th = rng.choice([1.0, 2.5, 4.0, 8.0, 20.0]), rng.uniform(0, 2 * np.pi)
            cen = nom_s[0:2, T // 2] + dist * np.array([np.cos(th), np.sin(th)])
            if rng.random() < 0.3:
                obs.append(mp.rdaobs(np.array([[1, 0], [0, 1], [0, 0.0]]), np.array([[cen[0]], [cen[1]], [-rng.uniform(0.3, 1.5)]]), "norm2", None, None))
            else:
                k = int(rng.integers(3, E + 1))
                A_, b_ = random_polygon(rng, cen, k, rng.uniform(0.5, 2.0), k)
                obs.append(mp.rdaobs(A_, b_.reshape(-1, 1), "Rpositive", None, None))
        r.assign_state_parameter(nom_s, nom_u, dis)
        r.assign_obstacle_parameter(obs)
        r.assign_combine_parameter_stateobs()
        for n in range(N):
            r.para_xi_list[n].value = np.vstack([np.zeros((1, 2)), rng.normal(0, rng.choice([0, 0.05, 0.5]), (T, 2))])
            r.para_zeta_list[n].value = rng.normal(0, rng.choice([0, 0.3, 2.0]), (1, T))
        for n in range(N):
            prob = r.prob_LamMuZ_list[n]
            prob.solve()
            if prob.status != "optimal":
                continue
            Im, Hm = r.indep_Im_array_LamMuZ[n].value, r.indep_Hm_array_LamMuZ[n].value
            for t in range(T):
                rows["A"].append(np.array(r.para_obstacle_list[n]["A"][t + 1].value, float))
                rows["b"].append(np.array(r.para_obstacle_list[n]["b"][t + 1].value, float).ravel())
                rows["cone"].append(int(r.para_obstacle_list[n]["cone_type"].value[1] > 0.5))
                rows["p"].append(nom_s[0:2, t + 1].copy()); rows["phi"].append(nom_s[2, t])
                rows["xi"].append(np.array(r.para_xi_list[n].value[t + 1], float)); rows["zeta"].append(float(r.para_zeta_list[n].value[0, t]))
                rows["dbar"].append(float(dis[0, t]))
                rows["cost"].append(0.5 * float(Im[t]) ** 2 + 0.5 * float(np.sum(Hm[t] ** 2)))
                rows["Im"].append(float(Im[t])); rows["H"].append(np.array(Hm[t], float))
    for k, v in rows.items():
        out[f"lmz.{k}"] = np.array(v)
    out["lmz.G"], out["lmz.h"] = G, h
    print("LamMuZ problems (not accelerated):", len(rows["cost"]), "sub-problems;", int(np.sum(np.array(rows["cost"]) > 1e-6)), "with a positive optimal value")
    np.savez_compressed(os.path.join(HERE, "ref_problems_na" + SUFFIX + ".npz"), **out)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=["refshim", "cvxpy"], default="refshim")
    ap.add_argument("--only-na", action="store_true", help="add the non-accelerated LamMuZ fixtures without touching the other files")
    args = ap.parse_args()
    rs, mp, backend = rh.load()
    if backend != args.backend:
        sys.exit(f"make_ref_golden: --backend {args.backend} requested, but the reference imports on {backend!r} here "
                 + ("(cvxpy / ecos are not installed: oracle/refshim answers)" if backend == "refshim" else "(the real cvxpy is importable: pass --backend cvxpy)"))
    if backend == "cvxpy":
        SUFFIX, INJECT_ORACLE = "_cvxpy", False
    orc = orc_api()
    orc.lib.orc_set_su_warm.argtypes = [C.c_double, C.c_double, C.c_int]
    orc.lib.orc_set_su_warm(0.0, 0.0, 0)
    if args.only_na:
        lammuz_problems_nonaccelerated(rs, mp)
        sys.exit(0)
    plumbing(rs, mp, orc)
    problems(rs, mp)
    lammuz_problems_nonaccelerated(rs, mp)
    for f in ("ref_plumbing", "ref_problems", "ref_problems_na"):
        print(f + SUFFIX + ".npz", os.path.getsize(os.path.join(HERE, f + SUFFIX + ".npz")) // 1024, "KiB")
