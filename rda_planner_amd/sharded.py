"""
Obstacle-sharded RDA solve across the GPUs of one node (include/rda_hip.h `rda_shard_*`).

The reference fans the N per-obstacle LamMuZ problems out over a `pathos` process pool and gathers
`(lam, mu, z)` back through pipes once per ADMM iteration (rda_solver.py:706-725).  Here rank r owns the
obstacle slots [r*ceil(N/P), (r+1)*ceil(N/P)): their duals never leave that GPU; what is exchanged per iteration is what the
su-problem reads - 24 bytes per (slot, stage) (a, the hinge offset) plus 48 bytes per (stage, 8-slot block) of reduced sums,
residual partials and the near mask - as ONE all-gather.  Every rank
then solves the identical su-problem, so no broadcast is needed and all ranks take the same early-stop
decision.

Two ways to run the exchange:
  * `enable_rccl(solver, rank, world, broadcast_bytes)` - in-library `ncclAllGather` over xGMI on the handle's
    stream; afterwards `solver.iterative_solve` / `rda_enqueue_step` are used unchanged (no host involvement).
  * `ShardedRDA(solver, rank, world, all_gather)` - the ADMM loop is driven from the host and `all_gather`
    is any callable (torch.distributed with gloo/nccl, MPI, an in-process emulation for tests).
"""
import ctypes as C

import numpy as np

from ._capi import Info, dptr, iptr, f64


def enable_rccl(solver, rank, world, broadcast_bytes):
    """`broadcast_bytes(buf: bytes | None) -> bytes` ships rank 0's 128-byte RCCL unique id to every rank."""
    api, h = solver._be.api, solver._be.handle
    rc = api.shard_config(h, rank, world)
    if rc != 0:
        raise RuntimeError(f"shard_config failed ({rc})")
    uid = C.create_string_buffer(128)
    if rank == 0:
        rc = api.lib.rda_shard_unique_id(h, uid)
        if rc != 0:
            raise RuntimeError(f"rda_shard_unique_id failed ({rc})")
    raw = broadcast_bytes(uid.raw if rank == 0 else None)
    uid = C.create_string_buffer(raw, 128)
    rc = api.lib.rda_shard_comm_init(h, uid)
    if rc != 0:
        raise RuntimeError(f"rda_shard_comm_init failed ({rc})")


class ShardedRDA:
    """host-driven sharded ADMM loop on top of an `RDA_solver` (any backend exposing the C-ABI pieces)"""

    def __init__(self, solver, rank, world, all_gather):
        self.solver, self.rank, self.world, self.all_gather = solver, rank, world, all_gather
        self.api, self.h = solver._be.api, solver._be.handle
        rc = self.api.shard_config(self.h, rank, world)
        if rc != 0:
            raise RuntimeError(f"shard_config failed ({rc})")
        self.chunk = self.api.shard_chunk_doubles(self.h)

    def iterative_solve(self, nom_s, nom_u, ref_states, ref_speed, obstacle_list):
        s = self.solver
        T = s.T
        ref = f64(np.hstack(ref_states)[0:3, :], (3, T + 1))
        nom_s, nom_u = f64(nom_s, (3, T + 1)), f64(nom_u, (2, T))
        n_obs, A, b, cone, per_t = s._stage(obstacle_list)
        api, h = self.api, self.h

        def ok(rc, what):           # never `assert` a call with side effects: python -O would drop the call itself
            if rc != 0:
                raise RuntimeError(f"{api.prefix}_{what} failed with code {rc}")
        ok(api.upload_obstacles(h, n_obs, dptr(A), dptr(b), iptr(cone), per_t), "upload_obstacles")
        ok(api.admm_begin(h, dptr(nom_s), dptr(nom_u), dptr(ref), float(ref_speed)), "admm_begin")
        stopped = C.c_int(0)
        mine = np.zeros(self.chunk)
        for it in range(s.iter_num):
            ok(api.admm_su(h, it, C.byref(stopped)), "admm_su")
            if stopped.value:
                break
            ok(api.admm_lammuz(h), "admm_lammuz")
            ok(api.shard_get_chunk(h, dptr(mine)), "shard_get_chunk")
            everyone = np.ascontiguousarray(self.all_gather(mine), dtype=np.float64)
            if everyone.size != self.chunk * self.world:
                raise RuntimeError(f"all_gather returned {everyone.size} doubles, expected {self.chunk * self.world}")
            ok(api.shard_set_chunks(h, dptr(everyone)), "shard_set_chunks")
        u, so, info = np.zeros((2, T)), np.zeros((3, T + 1)), Info()
        ok(api.admm_finish(h, dptr(u), dptr(so), C.byref(info)), "admm_finish")
        return u, {"opt_state_list": [so[:, i:i + 1].copy() for i in range(T + 1)], "ref_traj_list": ref_states,
                   "resi_dual": info.resi_dual, "resi_pri": info.resi_pri, "iters": info.iters, "status": info.su_status}
