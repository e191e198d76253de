"""The ONE line the driver parses.  bench.py collects a large dictionary (every leg with its `what` strings, thread sweeps, per-size rooflines);
round 5 printed all of it as the last stdout line - 21.5 KB - and the driver, whose tail holds 8 018 characters, recorded `parsed: null`.
Now: the full dictionary goes to gpurun_out/bench_detail.json (tools/profile_round.sh copies it to profiles/); the ONLY stdout line of a
top-level run is `compact(full)`: <= MAX_CHARS characters, every key the contract names (metric, value, unit, n_gpus, steps, warmup,
ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config, roofline, cpu_baseline) and one number per leg.
tests/test_benchlib.py builds the line from a worst-case dictionary and asserts the bound and the json round trip."""
import json
import os

from .context import ROOT

MAX_CHARS = 6000
PROTOCOL_CHARS = 200

_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches", "skipped_launches",
              "algorithmic_bytes_per_launch", "ipc_per_wave", "fp64_frac", "lds_conflict_ratio")
_RESI_KEYS = ("iter_threshold", "median_resi_dual", "max_resi_dual", "steps_below_threshold", "steps", "su_interior_point_iters_per_step")


def _num(x, nd=4):
    """numbers short: 6 significant figures are what a reader of the line uses"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        xf = float(x)
    except (TypeError, ValueError):
        return str(x)[:40]
    if xf != xf or xf in (float("inf"), float("-inf")):      # no NaN / Infinity tokens in the line (not JSON)
        return None
    return float(f"{xf:.6g}")


def _pick(d, keys):
    return {k: _num(d.get(k)) for k in keys if isinstance(d, dict) and k in d}


def _roof(r):
    return _pick(r, _ROOF_KEYS) if isinstance(r, dict) else None


def _sps(leg):
    """one number of a closed-loop leg"""
    if not isinstance(leg, dict):
        return None
    if "error" in leg:
        return "error"
    return _num(leg.get("steps_per_s", leg.get("aggregate_steps_per_s")))


def _window(w):
    return _pick(w, ("steps", "steps_per_s", "median_steps_per_s", "mean_admm_iters")) if isinstance(w, dict) else None


def _cpu(c):
    if not isinstance(c, dict):
        return None
    out = _pick(c, ("value", "unit", "cores", "kind", "host_cores", "single_thread", "steps", "max_du_vs_gpu"))
    if "sample" in c:
        out["sample"] = str(c["sample"])[:160]
    return out


def _size(e):
    if not isinstance(e, dict):
        return None
    if "error" in e or "skipped" in e:
        return {k: str(e[k])[:80] for k in ("error", "skipped") if k in e}
    out = _pick(e, ("value", "ms_per_step", "mean_admm_iters", "steps"))
    r, r2, c = e.get("roofline") or {}, e.get("roofline_secondary") or {}, e.get("cpu_baseline") or {}
    out["roofline"] = _pick(r, ("kernel", "frac", "avg_launch_us", "traffic"))
    out["roofline_secondary"] = _pick(r2, ("kernel", "frac", "avg_launch_us", "traffic"))
    out["cpu_baseline"] = _pick(c, ("value", "cores", "steps"))
    for k in ("fixed_slot_binding_steps_per_s", "gpu_over_cpu_port"):
        if e.get(k) is not None:
            out[k] = _num(e[k])
    fl = e.get("multi_ego_fleet")
    if isinstance(fl, dict):
        out["fleet"] = _pick(fl, ("egos", "aggregate_steps_per_s"))
        cl = fl.get("c_abi_closed_loop")
        if isinstance(cl, dict):
            out["fleet"]["closed_loop_ego_steps_per_s"] = _num(cl.get("ego_steps_per_s"))
            if isinstance(cl.get("fleets_4_host_threads"), dict):
                out["fleet"]["closed_loop_4_fleets_ego_steps_per_s"] = _num(cl["fleets_4_host_threads"].get("ego_steps_per_s"))
    return out


def compact(full):
    """full bench dictionary -> the driver's line (a dict; `emit` serialises it)"""
    out = {k: _num(full.get(k)) if k not in ("metric", "unit", "dtype", "data", "scaling") else full.get(k)
           for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    out["config"] = {"workload": str(cfg.get("workload", ""))[:160], "parallelism": str(cfg.get("parallelism", ""))[:100],
                     "protocol": str(cfg.get("protocol", ""))[:PROTOCOL_CHARS]}
    for k in ("median_ms_per_step", "mean_admm_iters", "max_du_vs_python_closed_loop"):
        if full.get(k) is not None:
            out[k] = _num(full[k])
    if full.get("residuals"):
        out["residuals"] = _pick(full["residuals"], _RESI_KEYS)
    if full.get("second_window"):
        out["second_window"] = _window(full["second_window"])
    out["roofline"] = _roof(full.get("roofline"))
    out["roofline_secondary"] = _roof(full.get("roofline_secondary"))
    if full.get("cpu_baseline"):
        out["cpu_baseline"] = _cpu(full["cpu_baseline"])
    # one number per leg (steps/s); everything else of a leg is in the detail file
    legs = {}
    for k in ("fixed_slot_binding", "su_hard_warm_off", "su_tol_early", "duals_follow_obstacles", "pcie_inclusive", "python_caller_closed_loop"):
        if full.get(k) is not None:
            legs[k] = _sps(full[k])
    if isinstance(full.get("device_resident_replay"), dict):
        legs["device_resident_replay"] = _num(full["device_resident_replay"].get("steps_per_s"))
    pa = full.get("python_api_closed_loop")
    if isinstance(pa, dict):
        legs["python_mpc_control"] = {"host_staging": _num(pa.get("host_obstacle_staging_steps_per_s")), "device_scene": _sps(pa.get("device_obstacles")),
                                      "device_scene_and_tracking": _sps(pa.get("device_obstacles_and_tracking"))}
    if isinstance(full.get("multi_ego_one_gpu"), dict):
        legs["multi_ego_one_gpu"] = _pick(full["multi_ego_one_gpu"], ("egos", "aggregate_steps_per_s"))
    fl = full.get("multi_ego_fleet")
    if isinstance(fl, dict):
        legs["multi_ego_fleet"] = _pick(fl, ("egos", "aggregate_steps_per_s"))
        if isinstance(fl.get("python_api_closed_loop"), dict):
            legs["multi_ego_fleet"]["python_fleet_control"] = _sps(fl["python_api_closed_loop"])
        if isinstance(fl.get("c_abi_closed_loop"), dict):
            legs["multi_ego_fleet"]["c_abi_closed_loop_ego_steps_per_s"] = _num(fl["c_abi_closed_loop"].get("ego_steps_per_s"))
            if isinstance(fl["c_abi_closed_loop"].get("fleets_4_host_threads"), dict):
                legs["multi_ego_fleet"]["c_abi_closed_loop_4_fleets_ego_steps_per_s"] = _num(fl["c_abi_closed_loop"]["fleets_4_host_threads"].get("ego_steps_per_s"))
    ip = full.get("lammuz_interior_point_closed_loops")
    if isinstance(ip, dict):
        e = {}
        for name, ent in ip.items():
            if isinstance(ent, dict):
                e[name] = {p: {"steps_per_s": _sps(v), "mean_admm_iters": _num(v.get("mean_admm_iters")),
                               "su_ip_iters_per_step": _num((v.get("residuals") or {}).get("su_interior_point_iters_per_step"))}
                           for p, v in ent.items() if isinstance(v, dict)}
            else:
                e[name] = str(ent)[:80]
        legs["lammuz_interior_point"] = e
    if legs:
        out["legs"] = legs
    if isinstance(full.get("sizes"), dict):
        out["sizes"] = {k: _size(v) for k, v in full["sizes"].items()}
    sh = full.get("obstacle_shard_leg")
    if isinstance(sh, dict):
        if "error" in sh:
            out["obstacle_shard_leg"] = {"error": str(sh["error"])[:120]}
        else:
            e = _pick(sh, ("steps_per_s", "ms_per_step", "unsharded_one_gpu_steps_per_s", "speedup_vs_one_gpu", "gather_us_per_iteration", "gathers", "chunk_bytes_per_rank"))
            am = sh.get("amdahl") or {}
            e["amdahl_bound"] = _num(am.get("bound_speedup_with_measured_gather", am.get("bound_speedup_without_exchange")))
            e["workload"] = str(sh.get("workload", ""))[:100]
            out["obstacle_shard_leg"] = e
    if full.get("per_rank_steps_per_s") is not None:
        out["per_rank_steps_per_s"] = [_num(v) for v in full["per_rank_steps_per_s"]][:16]
    if full.get("only_headline"):
        out["only_headline"] = True
    out["detail"] = "gpurun_out/bench_detail.json"
    return out


def line(full):
    """the serialised compact line; sheds optional blocks (never the contract keys) if a pathological input pushes it over the bound"""
    c = compact(full)
    s = json.dumps(c, separators=(",", ":"))
    for drop in ("legs", "obstacle_shard_leg", "sizes", "second_window", "roofline_secondary"):
        if len(s) <= MAX_CHARS:
            break
        c.pop(drop, None)
        c["dropped_for_length"] = c.get("dropped_for_length", []) + [drop]
        s = json.dumps(c, separators=(",", ":"))
    assert len(s) <= MAX_CHARS, len(s)
    return s


def emit(full, detail_name="bench_detail.json", detail_to_stdout=False):
    """full dictionary -> gpurun_out/<detail_name>; the compact line is the ONLY thing a top-level run writes to stdout (the driver may keep
    the head or the tail of stdout: keep all of it small).  detail_to_stdout: the `sizes` sub-runs hand their full dictionary to the parent
    on a stdout line that starts with DETAIL."""
    txt = json.dumps(full)
    try:
        if detail_name:
            d = os.path.join(ROOT, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, detail_name), "w") as f:
                f.write(txt + "\n")
    except OSError:
        pass
    if detail_to_stdout:
        print("DETAIL " + txt, flush=True)
    print(line(full), flush=True)
