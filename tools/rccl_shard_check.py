"""torchrun --nproc-per-node P tools/rccl_shard_check.py : obstacle-sharded solve with the in-library
ncclAllGather (xGMI) against the un-sharded solve on rank 0.  Prints RCCL_SHARD_OK on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("gloo")                       # side channel for the 128-byte unique id only
    from rda_planner_amd._lib import hip_api
    from rda_planner_amd.rda_solver import RDA_solver
    from rda_planner_amd.sharded import enable_rccl
    from rda_planner_amd import scenarios as sc
    from rda_planner_amd.mpc import MPC
    hip_api().lib.rda_set_device(local)
    car_t = sc.rectangle_robot(dynamics="acker")
    T, N = 12, 8 * world + int(os.environ.get("RDA_SHARD_EXTRA", "1"))      # default: N % world != 0 (padded last shard)
    obstacles = sc.scene_polygons(N, lo=(4, -8), hi=(24, 8), seed=3)
    conv = MPC.__new__(MPC)
    conv.receding, conv.dt, conv.state = T, 0.1, np.zeros((3, 1))
    rl = MPC.convert_rda_obstacle(conv, obstacles, np.zeros((3, 1)), False)

    def bcast(buf):
        t = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            t = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
        dist.broadcast(t, 0)
        return bytes(t.numpy().tobytes())
    sharded = RDA_solver(T, car_t, 4, N, iter_num=3, time_print=False)
    enable_rccl(sharded, rank, world, bcast)
    single = RDA_solver(T, car_t, 4, N, iter_num=3, time_print=False) if rank == 0 else None
    rng = np.random.default_rng(1)
    ok = True
    for k in range(5):
        nom_u = np.vstack([np.full(T, 3.0), rng.uniform(-0.1, 0.1, T)])
        nom_s = np.zeros((3, T + 1))
        for t in range(T):
            nom_s[:, t + 1] = nom_s[:, t] + 0.1 * np.array([3.0 * np.cos(nom_s[2, t]), 3.0 * np.sin(nom_s[2, t]), np.tan(nom_u[1, t])])
        ref = [np.array([[0.4 * t], [0.05 * k], [0.0]]) for t in range(T + 1)]
        u, info = sharded.iterative_solve(nom_s, nom_u, ref, 4.0, list(rl))
        if rank == 0:
            u1, i1 = single.iterative_solve(nom_s, nom_u, ref, 4.0, list(rl))
            ok &= bool(np.abs(u - u1).max() < 1e-8 and info["iters"] == i1["iters"])
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    if rank == 0:
        print("RCCL_SHARD_OK" if ok else "RCCL_SHARD_MISMATCH")
    sys.exit(0 if flag.item() else 1)


if __name__ == "__main__":
    main()
