"""One-off (round 5): ADMM iteration by iteration (rda_admm_*), the packed-rows LamMuZ kernel with remembered circle supports against the one-row-per-wave
kernel (circle rows enumerated), both from the same solver state: the FIRST LamMuZ launch whose duals differ, and the rows."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import soak_lib  # noqa: E402
from rda_planner_amd import scenarios as sc  # noqa: E402
from rda_planner_amd._capi import Info, dptr, iptr  # noqa: E402
from rda_planner_amd._lib import hip_api  # noqa: E402
from rda_planner_amd.mpc import MPC  # noqa: E402
from rda_planner_amd.rda_solver import hip_options  # noqa: E402

seed, scene, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
last = int(sys.argv[4]) if len(sys.argv) > 4 else scene
TOL = float(os.environ.get("DIFF_TOL", "1e-12"))
api = hip_api()
rng = np.random.default_rng(seed)
draws = [soak_lib.draw_scene(rng, seed, s, 80, circles=True) for s in range(last + 1)]
for scene_i in range(scene, last + 1):
    d = draws[scene_i]
    kw = dict(d["kw"], device_track=False, device_obstacles=False)
    ms = [MPC(d["car"], [p.copy() for p in d["path"]], **kw, hip_opts=hip_options(**o)) for o in (dict(lmz_rows=1), (dict(lmz_rows=1) if os.environ.get("SAME_KERNEL") == "2" else dict(lmz_rows=1, lmz_warm=0)) if os.environ.get("SAME_KERNEL") else dict(lmz_rows=0))]
    if os.environ.get("CHUNK"):
        for m in ms:
            assert api.shard_config(m.rda._be.handle, 0, 1) == 0
    st = d["path"][0].copy().reshape(3, 1)
    T, K = d["kw"]["receding"], d["kw"]["iter_num"]
    np.set_printoptions(linewidth=220, precision=12)
    for k in range(steps):
        cur = [o if not np.any(o.velocity) else (o._replace(vertex=o.vertex + o.velocity * (0.1 * k)) if o.cone_type == "Rpositive"
                                                 else o._replace(center=o.center + o.velocity * (0.1 * k))) for o in d["scene"]]
        ins = []
        for m in ms:
            cur_ref_path, speed, nom_s, ref_list = m._begin(st.copy(), d["speed"])
            rl = m.convert_rda_obstacle(list(cur), m.state, m.obstacle_order)
            n, A, b, cone, per_t = m.rda._stage(list(rl))
            h = m.rda._be.handle
            assert api.upload_obstacles(h, n, dptr(A), dptr(b), iptr(cone), per_t) == 0
            refa = np.ascontiguousarray(np.hstack(ref_list)[0:3, :])
            assert api.admm_begin(h, dptr(np.ascontiguousarray(nom_s)), dptr(np.ascontiguousarray(m.cur_vel_array)), dptr(refa), float(speed)) == 0
            ins.append((cur_ref_path, cone))
        found = False
        for it in range(K):
            stop = [C.c_int(0), C.c_int(0)]
            for j, m in enumerate(ms):
                assert api.admm_su(m.rda._be.handle, it, C.byref(stop[j])) == 0
            if os.environ.get("SU_HIST"):          # the su-solves' own state after this su launch: history keys + kept multipliers
                hs = []
                for m in ms:
                    hist = (C.c_int32 * 4)(); keep = np.zeros(10 * T)
                    assert api.lib.rda_get_su_history(m.rda._be.handle, C.cast(hist, C.POINTER(C.c_int)), dptr(keep)) == 0
                    hs.append((list(hist), keep))
                if hs[0][0] != hs[1][0] or np.abs(hs[0][1] - hs[1][1]).max() > 0:
                    print(f"scene {scene_i} step {k}: su history differs after su({it}): {hs[0][0]} vs {hs[1][0]}, kept multipliers max diff {np.abs(hs[0][1] - hs[1][1]).max():.3e}")
            if stop[0].value or stop[1].value:
                break
            for m in ms:
                assert api.admm_lammuz(m.rda._be.handle) == 0
            if os.environ.get("CHUNK"):            # what the su-problem READS of this launch: (ax, ay, cb) [T][N], block sums [T][J][5], near masks [T][J]
                cs = []
                for m in ms:
                    n_ch = api.shard_chunk_doubles(m.rda._be.handle)
                    c = np.zeros(n_ch); assert api.shard_get_chunk(m.rda._be.handle, dptr(c)) == 0
                    cs.append(c)
                if not np.array_equal(cs[0].view(np.uint64), cs[1].view(np.uint64)):
                    idx = np.flatnonzero(cs[0].view(np.uint64) != cs[1].view(np.uint64))
                    N_ = ms[0].rda.max_obs_num; J_ = -(-N_ // 8)
                    print(f"scene {scene_i} step {k} iteration {it}: {len(idx)} words of the su-problem's input chunk differ (chunk {len(cs[0])} = 3*{T}*{N_} + 6*{T}*{J_}); first {idx[:8].tolist()}; values {cs[0][idx[:4]]} vs {cs[1][idx[:4]]}")
                    sa_, sb_ = ms[0].rda.get_state(), ms[1].rda.get_state()
                    for w in idx[:3]:
                        if w < 3 * T * N_:
                            arr, rem = divmod(int(w), T * N_); t_, n_ = divmod(rem, N_)
                            print(f"   word {w}: array {arr} (0 ax, 1 ay, 2 cb) stage {t_} slot {n_} cone {ins[0][1][n_]}: {cs[0][w]!r} vs {cs[1][w]!r}")
                            for key in ("lam", "mu", "xi"):
                                print(f"     {key} col {t_ + 1}: {sa_[key][n_][t_ + 1].tolist()} | {sb_[key][n_][t_ + 1].tolist()}")
                            print(f"     z {sa_['z'][n_][t_]!r} | {sb_['z'][n_][t_]!r}   zeta {sa_['zeta'][n_][t_]!r} | {sb_['zeta'][n_][t_]!r}   dis {sa_['dis'][t_]!r} | {sb_['dis'][t_]!r}")
            sa, sb = ms[0].rda.get_state(), ms[1].rda.get_state()
            dl = np.abs(sa["lam"] - sb["lam"])
        if os.environ.get("ALL_DUALS"):        # lam [N][T+1][E]: fold the other duals' differences into the same (slot, column) grid
            dm_ = np.abs(sa["mu"] - sb["mu"]).max(axis=2); dx_ = np.abs(sa["xi"] - sb["xi"]).reshape(dl.shape[0], dl.shape[1], -1).max(axis=2)
            dz_ = np.zeros_like(dm_); dz_[:, 1:] = np.maximum(np.abs(sa["z"] - sb["z"]), np.abs(sa["zeta"] - sb["zeta"]))
            dl = np.maximum(dl, np.maximum(np.maximum(dm_, dx_), dz_)[:, :, None])
            if dl.max() > TOL:
                per = dl.reshape(dl.shape[0], dl.shape[1], -1).max(axis=2)
                rows = np.argwhere(per > TOL)
                cones = np.asarray(ins[0][1])[rows[:, 0]]
                print(f"   dis differs: {np.abs(sa['dis'] - sb['dis']).max():.3e}; a_lam {np.abs(sa['a_lam'] - sb['a_lam']).max():.3e}; rows by cone: circle {int((cones == 1).sum())} polygon {int((cones == 0).sum())}; columns {sorted(set(rows[:, 1].tolist()))[:8]}")
                print(f"scene {scene_i} step {k} iteration {it}: {len(rows)} rows differ, max {dl.max():.3e}")
                for n_, t_ in rows[:6]:
                    print(f"  slot {n_} (cone {ins[0][1][n_]}) column {t_}: lam rows {sa['lam'][n_][t_].tolist()} enum {sb['lam'][n_][t_].tolist()}  |a| {np.hypot(*sa['lam'][n_][t_][:2]):.12f} / {np.hypot(*sb['lam'][n_][t_][:2]):.12f}")
                    print(f"      xi rows {sa['xi'][n_][t_] if sa['xi'].ndim == 3 else sa['xi'].reshape(dl.shape[0], dl.shape[1], -1)[n_][t_]} enum {sb['xi'][n_][t_] if sb['xi'].ndim == 3 else sb['xi'].reshape(dl.shape[0], dl.shape[1], -1)[n_][t_]}")
                print(f"      mu rows {sa['mu'][n_][t_].tolist()} enum {sb['mu'][n_][t_].tolist()}   z {sa['z'][n_][t_ - 1] if t_ else None} / {sb['z'][n_][t_ - 1] if t_ else None}")
                found = True
                break
        if found:
            break
        outs = []
        for j, m in enumerate(ms):
            u = np.zeros((2, T)); so = np.zeros((3, T + 1)); info = Info()
            assert api.admm_finish(m.rda._be.handle, dptr(u), dptr(so), C.byref(info)) == 0
            outs.append(u)
            m._end(ins[j][0], u.copy(), {})
        ms[1].rda.set_state(ms[0].rda.get_state()); ms[1].cur_vel_array = ms[0].cur_vel_array.copy(); ms[1].cur_index = ms[0].cur_index
        st = sc.kinematic_step(st, outs[0][:, 0:1], d["car"], 0.1)
    else:
        print("scene", scene_i, "no difference in", steps, "steps")
