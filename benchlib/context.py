"""What every leg of bench.py shares: the parsed arguments, the process group, the C-ABI, the workload and its recorded traces."""
import os
import sys
import time
from dataclasses import dataclass, field
from typing import Any

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@dataclass
class Ctx:
    args: Any
    api: Any                     # rda_planner_amd._lib.hip_api()
    rank: int = 0
    world: int = 1
    dist: Any = None             # torch.distributed when world > 1
    tdev: str = "cpu"            # device of the tensors the reductions over ranks travel in
    shard: bool = False          # --mode shard: ONE ego, obstacles sharded over the ranks
    oversub: bool = False        # fewer GPUs than ranks (plumbing run)
    ndev: int = 1
    car_t: Any = None
    path: Any = None
    obstacles: Any = None
    kw: dict = field(default_factory=dict)        # MPC keyword arguments of the workload (obstacle_order=True: the reference's default)
    kw_rec: dict = field(default_factory=dict)    # ... with obstacle_order=False (slots bound once: what the replay legs need)
    T: int = 0
    N: int = 0
    K: int = 0
    W: int = 0
    path_length: float = 0.0
    trace: Any = None            # recorded Python closed loop, fixed slot binding
    staged: Any = None           # its staged obstacle arrays
    trace_o: Any = None          # recorded Python closed loop, re-sorted every tick (the headline workload), staged slots of every step
    u_ord: Any = None            # its controls [W+K][2] (static scenes)
    make_sharded: Any = None     # hook run on every new solver (obstacle shards + RCCL)

    # ---- over the ranks ---------------------------------------------------------------------------------------------------
    def barrier_all(self):
        if self.dist is not None:
            import torch
            self.dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        import torch
        tt = torch.tensor([x], dtype=torch.float64, device=self.tdev)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def gather_over_ranks(self, x):
        """one float per rank, on every rank (the per-rank values of the driver line)"""
        if self.dist is None:
            return [x]
        import torch
        mine = torch.tensor([x], dtype=torch.float64, device=self.tdev)
        everyone = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(everyone, mine)
        return [float(t.item()) for t in everyone]

    # ---- solvers -----------------------------------------------------------------------------------------------------------
    def new_solver(self, car=None, **extra):
        from rda_planner_amd.rda_solver import RDA_solver
        kw = self.kw
        sv = RDA_solver(self.T, car or self.car_t, kw["max_edge_num"], self.N, iter_num=kw["iter_num"], step_time=0.1, time_print=False, ro1=kw["ro1"], **extra)
        if self.make_sharded is not None:
            self.make_sharded(sv)
        return sv

    def closed_loop_host(self):
        """tools/libclosed_loop_host.so: the caller's loop in C, entry points of librda_hip.so handed over (tools/closed_loop_host.py)"""
        tools = os.path.join(ROOT, "tools")
        if tools not in sys.path:
            sys.path.insert(0, tools)
        import closed_loop_host as clh
        return clh.Host(self.api.lib)


def stats(K, elapsed, times, iters, **more):
    """the common block of a closed-loop leg"""
    import numpy as np
    out = {"steps_per_s": round(K / elapsed, 2), "median_ms_per_step": round(float(np.median(times) * 1e3), 5), "mean_admm_iters": round(float(np.mean(iters)), 3)}
    out.update(more)
    return out


now = time.perf_counter
