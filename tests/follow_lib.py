"""Shared by tests/test_oracle_follow.py (CPU) and tests/test_gpu_follow.py: the oracle driven piecewise with a hook between the first
su-problem and the first LamMuZ pass of a tick, and the re-arrangement of the dual rows that rda_opts::duals_follow performs on the
device.  Test infrastructure (drives oracle/)."""
import ctypes as C

import numpy as np


class PiecewiseOracle:
    """the oracle's api with `step` driven through orc_admm_*: `hook()` runs between the first su-problem and the first LamMuZ pass"""

    def __init__(self, base, iter_num):
        self._b, self.iter_num, self.hook = base, iter_num, None

    def __getattr__(self, k):
        return getattr(self._b, k)

    def step(self, h, nom_s, nom_u, ref, speed, n_obs, A, b, cone, per_t, out_u, out_s, info):
        B = self._b
        assert B.upload_obstacles(h, n_obs, A, b, cone, per_t) == 0
        assert B.admm_begin(h, nom_s, nom_u, ref, speed) == 0
        stopped = C.c_int(0)
        for it in range(self.iter_num):
            assert B.admm_su(h, it, C.byref(stopped)) == 0
            if stopped.value:
                break
            if it == 0 and self.hook is not None:
                self.hook()
            assert B.admm_lammuz(h) == 0
        return B.admm_finish(h, out_u, out_s, info)


def follow(state_dict, prev, now):
    """rows of lam, mu, z, xi, zeta of the slots `now` (slot -> obstacle id) taken from where those obstacles sat in `prev`; zeros for newcomers"""
    where = {int(s): i for i, s in enumerate(prev)}
    new = {key: np.zeros_like(state_dict[key]) for key in ("lam", "mu", "z", "xi", "zeta")}
    for i, s in enumerate(now):
        j = where.get(int(s), -1)
        if j >= 0:
            for key in new:
                new[key][i] = state_dict[key][j]
    return dict(new, dis=None, a_lam=None, b_lam=None)
