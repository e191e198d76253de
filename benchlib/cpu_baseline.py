"""cpu_baseline leg of bench.py (task contract, section 4): the ORACLE (oracle/rda_oracle.c, kind "port") timed on the GPU box's host cores, rank 0
at N=1 only, on a bounded sample of the same workload.  This is the one place outside tests/ and smoke() that may touch oracle/ - as a
reported baseline, never as the thing measured or shipped."""
import ctypes as C
import os
import time

import numpy as np


def run(ctx):
    from oracle.oracle_backend import oracle_backend, api as orc_api
    from rda_planner_amd._capi import Info, dptr, iptr
    from rda_planner_amd.rda_solver import RDA_solver
    args, kw, T, N, K, W, trace, trace_o, staged = ctx.args, ctx.kw, ctx.T, ctx.N, ctx.K, ctx.W, ctx.trace, ctx.trace_o, ctx.staged
    ncore = os.cpu_count() or 1
    cpu = RDA_solver(T, ctx.car_t, kw["max_edge_num"], N, iter_num=kw["iter_num"], step_time=0.1, time_print=False, ro1=kw["ro1"], _backend=oracle_backend)
    info_c, ou, os_ = Info(), np.zeros((2, T)), np.zeros((3, T + 1))
    sweep, err, n_total = {}, 0.0, 0
    k = 0
    counts = sorted({t for t in (1, 8, 16, 32, 64, ncore) if t <= ncore}) if not args.cpu_threads else [min(args.cpu_threads, ncore)]
    min_steps = 20 if args.cpu_threads else 3            # a `sizes` leg is ONE thread count: at least 20 steps of it (N = 2000: ~16 s)
    for nthr in counts:
        orc_api().lib.orc_set_threads(nthr)
        n_cpu, t_cpu = 0, 0.0
        while t_cpu < 2.5 or n_cpu < min_steps:          # consecutive steps of ONE closed loop (the duals stay warm) ...
            kk = k % (W + K)                             # ... wrapping around the recorded trace when it is used up
            tr_c = trace_o if trace_o is not None else trace     # the headline workload: the scene re-sorted on every tick
            n_c, A_c, b_c, cone_c, pt_c = tr_c["staged"][kk] if tr_c.get("staged") else (staged["n"], staged["A"], staged["b"], staged["cone"], staged["per_t"])
            t1 = time.perf_counter()
            cpu._be.api.step(cpu._be.handle, dptr(tr_c["nom_s"][kk]), dptr(tr_c["nom_u"][kk]), dptr(tr_c["ref"][kk]), float(tr_c["speed"][kk]),
                             n_c, dptr(A_c), dptr(b_c), iptr(cone_c), pt_c, dptr(ou), dptr(os_), C.byref(info_c))
            t_cpu += time.perf_counter() - t1
            if k < W + K:                                # first pass only: the same state history as the GPU run
                err = max(err, float(np.abs(ou - tr_c["u_solver"][kk]).max()))
            n_cpu += 1
            k += 1
        sweep[nthr] = round(n_cpu / t_cpu, 3)
        n_total += n_cpu
    best = max(sweep, key=sweep.get)
    return {"value": sweep[best], "unit": "steps/s", "cores": best, "kind": "port",
            "single_thread": sweep.get(1), "thread_sweep": sweep, "host_cores": ncore, "steps": n_total,
            "sample": f"{n_total} steps of the headline closed loop (obstacle_order=True: the staged slots of every tick as the GPU run had them; consecutive, "
                      f"wrapping around), >= 2.5 s and >= {min_steps} steps per thread count (oracle/rda_oracle.c: OpenMP over obstacles, OMP_PROC_BIND=close, su-problem serial, the "
                      "kernel's start rules mirrored); best thread count reported",
            "note": "a restatement of the ADMM in C, NOT the reference's CVXPY+ECOS+pathos path (not installable here): the "
                    "north-star '>=100x the reference CPU path' cannot be measured against this number",
            "max_du_vs_gpu": err}
