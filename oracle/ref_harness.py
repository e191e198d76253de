"""ORACLE - test infrastructure only.

Runs the UNMODIFIED reference classes (`RDA_planner.rda_solver.RDA_solver`, `RDA_planner.mpc.MPC`, imported
from /root/reference by oracle/ref_loader.py) and offers two ways of answering their `prob.solve(...)` calls
(rda_solver.py:693,768,800 - CVXPY -> ECOS in the original, neither installable here):

  mode "ipm"    - the problems the reference's own construction code builds are solved as they stand by the
                  generic interior-point method of oracle/refshim (nothing of this repo's oracle involved):
                  pins the oracle's two argmins on the reference formulation itself.
  mode "oracle" - `Problem.solve` of the reference's problem objects is answered by the oracle's pure argmin
                  functions (`orc_su_solve`, `orc_lammuz_one`) fed ONLY from the reference's parameter objects:
                  everything else - parameter staging, padding, the ADMM order, residuals, early stop, dual
                  updates, quirks Q1..Q12 - is reference code executing, so `orc_step` / `rda_step` can be
                  compared with it iteration by iteration.

Nothing here is reachable from the product path (rda_planner_amd/).
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from rda_planner_amd._capi import Cfg, DYNAMICS, dptr, f64

from . import ref_loader

car = namedtuple("car", "G h cone_type wheelbase max_speed max_acce dynamics")


def make_cfg_from_reference(ref_solver, delta=1e-6, eps_u=1e-8):
    """orc_cfg / rda_cfg filled from the attributes of a constructed reference RDA_solver"""
    r = ref_solver
    c = Cfg()
    c.T, c.N, c.E, c.R = r.T, r.max_obs_num, r.max_edge_num, r.car_tuple.G.shape[0]
    c.dynamics = DYNAMICS[r.dynamics]
    c.accelerated = int(bool(r.accelerated))
    c.iter_num = r.iter_num
    c.robot_norm2 = int(r.car_tuple.cone_type == "norm2")
    c.dt, c.L = r.dt, float(r.L) if r.L else 0.0
    c.max_speed[0], c.max_speed[1] = float(r.max_speed[0, 0]), float(r.max_speed[1, 0])
    c.acce_bound[0], c.acce_bound[1] = float(r.acce_bound[0, 0]), float(r.acce_bound[1, 0])
    c.iter_threshold = r.iter_threshold
    c.ws, c.wu = float(r.ws), float(r.wu)
    c.slack_gain, c.max_sd, c.min_sd = float(r.para_slack_gain.value), float(r.para_max_sd.value), float(r.para_min_sd.value)
    c.ro1, c.ro2 = float(r.ro1.value), float(r.ro2.value)
    c.delta, c.eps_u = delta, eps_u
    return c


def reference_state(r):
    """every persistent parameter of a reference solver, in the layouts of orc_get_state / rda_get_state"""
    T, N = r.T, r.max_obs_num
    st = {
        "lam": np.stack([p.value.T for p in r.para_lam_list]),                 # [N][T+1][E]
        "mu": np.stack([p.value.T for p in r.para_mu_list]),                   # [N][T+1][R]
        "z": np.stack([p.value[0] for p in r.para_z_list]),                    # [N][T]
        "xi": np.stack([p.value for p in r.para_xi_list]),                     # [N][T+1][2]
        "zeta": np.stack([p.value[0] for p in r.para_zeta_list]),              # [N][T]
        "dis": np.array(r.para_dis.value, float).reshape(T),
        "a_lam": np.stack([p.value for p in r.para_obsA_lam_list]),            # [N][T+1][2]
        "b_lam": np.stack([p.value[:, 0] for p in r.para_obsb_lam_list]),      # [N][T+1]
        "s": np.array(r.para_s.value, float),
        "u": np.array(r.para_u.value, float),
    }
    assert st["lam"].shape[0] == N
    return {k: np.array(v, float, copy=True) for k, v in st.items()}


def su_inputs_from_reference(r):
    """condensed su-problem data read from the reference's parameter objects - including the stale
    `obsA_lam` / `obsb_lam` products of quirk Q4"""
    T, N = r.T, r.max_obs_num
    G, h = np.asarray(r.car_tuple.G, float), np.asarray(r.car_tuple.h, float).reshape(-1)
    a = np.zeros((N, T, 2))
    cc = np.zeros((N, T))
    g = np.zeros((N, T, 2))
    for n in range(N):
        mu = r.para_mu_list[n].value
        a[n] = r.para_obsA_lam_list[n].value[1:, :]
        cc[n] = r.para_obsb_lam_list[n].value[1:, 0] + mu[:, 1:].T @ h + r.para_z_list[n].value[0] - r.para_zeta_list[n].value[0]
        g[n] = mu[:, 1:].T @ G + r.para_xi_list[n].value[1:, :]
    return dict(nom_s=f64(r.para_s.value), nom_u=f64(r.para_u.value), ref=f64(r.para_ref_s.value),
                vref=float(r.para_ref_speed.value), a=f64(a), cc=f64(cc), g=f64(g), d0=f64(r.para_dis.value).reshape(T))


def check_linearisation(r, tol=0.0):
    """the reference's A_t, B_t, C_t, R_t lists vs. what the nominal they were built from implies (sanity of the
    harness itself: `orc_su_solve` re-linearises about (nom_s, nom_u), the reference reads these lists)"""
    s, u = r.para_s.value, r.para_u.value
    for t in range(r.T):
        phi = s[2, t]
        Rm = np.array([[np.cos(phi), -np.sin(phi)], [np.sin(phi), np.cos(phi)]])
        assert np.max(np.abs(r.para_rot_list[t].value - Rm)) <= tol
    return True



def _set_status(prob, status):
    """`Problem.status` after a solve answered from outside: a plain attribute on the stand-in, a read-only property over `_status`
    in cvxpy proper - so that this harness (and tests/golden/make_ref_golden.py) also runs on a machine that has the real cvxpy"""
    try:
        prob.status = status
    except AttributeError:
        prob._status = status


class OracleAnswers:
    """mode "oracle": answer Problem.solve of a reference solver's problem objects with the oracle's argmins"""

    def __init__(self, ref_solver, rs_module, orc_api, tie_centre=True, checks=True):
        self.r, self.rs, self.api = ref_solver, rs_module, orc_api
        self.cfg = make_cfg_from_reference(ref_solver)
        self.G = f64(ref_solver.car_tuple.G)
        self.h = f64(ref_solver.car_tuple.h).ravel()
        self.checks = checks
        self.su_iters = []
        r = ref_solver
        r.prob_su.solve = self._su_solve
        probs = r.prob_LamMuZ_list if r.process_num == 1 else rs_module.prob_LamMuZ_list
        for n, prob in enumerate(probs):
            prob.solve = (lambda n_: (lambda solver=None, verbose=False, **kw: self._lmz_solve(n_, probs[n_])))(n)
        self.probs = probs

    # ---- su-problem (rda_solver.py:692-700) --------------------------------------------------------
    def _su_solve(self, solver=None, verbose=False, **kw):
        r = self.r
        T = r.T
        inp = su_inputs_from_reference(r)
        if self.checks:
            check_linearisation(r)
        s, u, d = np.zeros((3, T + 1)), np.zeros((2, T)), np.zeros(T)
        it = C.c_int(0)
        self.cfg.slack_gain, self.cfg.max_sd, self.cfg.min_sd = float(r.para_slack_gain.value), float(r.para_max_sd.value), float(r.para_min_sd.value)
        self.cfg.ro1, self.cfg.ro2 = float(r.ro1.value), float(r.ro2.value)
        st = self.api.lib.orc_su_solve(C.byref(self.cfg), dptr(inp["nom_s"]), dptr(inp["nom_u"]), dptr(inp["ref"]), inp["vref"],
                                       dptr(inp["a"]), dptr(inp["cc"]), dptr(inp["g"]), dptr(inp["d0"]), dptr(s), dptr(u), dptr(d),
                                       C.byref(it))
        self.su_iters.append(it.value)
        if st == 0:
            r.indep_s.value, r.indep_u.value, r.indep_dis.value = s, u, d.reshape(1, T)      # (public cvxpy API: Leaf.value setter)
            _set_status(r.prob_su, "optimal")
        else:
            _set_status(r.prob_su, "solver_error")
        return None

    # ---- one obstacle's LamMuZ problem (rda_solver.py:743-826) -------------------------------------
    def _lmz_solve(self, n, prob):
        r = self.r
        T, E, R = r.T, r.max_edge_num, self.G.shape[0]
        rs = self.rs
        if r.process_num == 1:
            para_s, para_dis, para_xi, para_zeta = r.para_s, r.para_dis, r.para_xi_list[n], r.para_zeta_list[n]
            para_obs, obsA_rot, obsA_trans = r.para_obstacle_list[n], r.para_obsA_rot_list[n], r.para_obsA_trans_list[n]
        else:           # the worker globals of the pool branch (rda_solver.py:268)
            para_s, para_dis, para_xi, para_zeta = rs.para_s, rs.para_dis, rs.para_xi_list[n], rs.para_zeta_list[n]
            para_obs, obsA_rot, obsA_trans = rs.para_obstacle_list[n], rs.para_obsA_rot_list[n], rs.para_obsA_trans_list[n]
        s = para_s.value
        cone = int(np.asarray(para_obs["cone_type"].value)[1] > 0.5)
        lam_prev, mu_prev = r.para_lam_list[n].value, r.para_mu_list[n].value
        lam = np.array(lam_prev, float, copy=True)          # column 0 is not determined by the problem: it is kept
        mu = np.array(mu_prev, float, copy=True)
        z = np.zeros((1, T))
        for t in range(T):
            A = f64(para_obs["A"][t + 1].value)
            b = f64(para_obs["b"][t + 1].value).ravel()
            p = f64(s[0:2, t + 1])
            phi = float(s[2, t])                            # quirk Q1: heading of column t, position of column t+1
            if self.checks:
                Rm = np.array([[np.cos(phi), -np.sin(phi)], [np.sin(phi), np.cos(phi)]])
                assert np.array_equal(obsA_rot[t + 1].value, A @ Rm), "obsA_rot is not A_{t+1} R(phi_t)"
                assert np.array_equal(obsA_trans[t + 1].value, A @ p.reshape(2, 1)), "obsA_trans is not A_{t+1} p_{t+1}"
            lo, mo, zo = np.zeros(E), np.zeros(R), C.c_double(0)
            self.api.lib.orc_lammuz_one(E, R, dptr(A), dptr(b), cone, dptr(p), phi, dptr(self.G), dptr(self.h),
                                        dptr(f64(para_xi.value[t + 1])), float(para_zeta.value[0, t]), float(para_dis.value[0, t]),
                                        float(r.ro2.value), self.cfg.delta, int(bool(r.accelerated)), dptr(lo), dptr(mo),
                                        C.cast(C.byref(zo), C.POINTER(C.c_double)), None)
            lam[:, t + 1], mu[:, t + 1], z[0, t] = lo, mo, zo.value
        r.indep_lam_list[n].value, r.indep_mu_list[n].value, r.indep_z_list[n].value = lam, mu, z
        _set_status(prob, "optimal")
        return None


def record_iterations(ref_solver):
    """wrap `rda_solver()` (one ADMM iteration, rda_solver.py:612-637) so that the full parameter state is
    snapshotted after every iteration; returns the list the snapshots are appended to"""
    log = []
    orig = ref_solver.rda_solver

    def wrapped():
        out = orig()
        snap = reference_state(ref_solver)
        snap["resi_dual"], snap["resi_pri"] = float(out[2]), float(out[3])
        log.append(snap)
        return out
    ref_solver.rda_solver = wrapped
    return log


def load():
    return ref_loader.load()
