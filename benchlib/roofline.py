"""Roofline objects of bench.py's JSON line (task contract, section 4): algorithmic bytes per EXECUTED launch / hipEvent time per executed launch
against the HBM peak, PMC traffic from the committed profile of the same workload, and the issue figures that say what really bounds these
kernels (instruction issue, not bytes).  Split out of bench.py in round 5."""
import json
import os

import numpy as np

PEAK_HBM_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~ 8 TB/s
PEAK_FP64_VECTOR = 78.6e12      # vector fp64 FLOP/s


def unit_bytes(E, R):
    """bytes one (obstacle, stage) sub-problem has to move - 288 B at E = R = 4"""
    return 8 * (5 * E + 2 * R + 8)


def su_bytes(T, n_loc, ranks=1):
    """k_su has no pass over the N terms: its set-up reads the reduced form the LamMuZ launch leaves behind - per (stage, 8-slot block) three sums
    and a near mask (32 of the 48 bytes of a block record) - plus the nominal / reference / kept multipliers; per interior-point pass it visits
    the NEAR terms only (24 B each, data dependent: not counted, so the fraction is a lower bound)"""
    J = -(-n_loc // 8)
    return 32 * T * J * ranks + 8 * (8 * (T + 1) + 5 * T + 10 * T + 4 * T)


def roof(name, ms, bytes_per_launch, n_exec):
    """per EXECUTED launch: launches queued behind the device early-stop flag return at once (no bytes, ~3 us) and are separated from the executed
    ones by their count (sum of rda_info.iters) - the n_exec longest launches are the executed ones"""
    ms = np.sort(np.asarray(ms, float))
    n_noop = max(ms.size - n_exec, 0)
    ex, noop = ms[n_noop:], ms[:n_noop]
    avg_s = float(ex.mean()) * 1e-3 if ex.size else 0.0
    ach = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
    return {"kernel": name, "bound": "hbm", "achieved": round(ach, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(ach / PEAK_HBM_GBS, 6), "traffic": None, "avg_launch_us": round(avg_s * 1e6, 2),
            "launches": int(ex.size), "total_ms": round(float(ex.sum()), 3), "algorithmic_bytes_per_launch": bytes_per_launch,
            "skipped_launches": int(noop.size), "skipped_avg_us": round(float(noop.mean()) * 1e3, 2) if noop.size else None,
            "avg_us_over_all_launches": round(float(ms.mean()) * 1e3, 2) if ms.size else None}


def attach_traffic(root, r_su, r_lm, N, T, moving, lm_kernel):
    """PMC bytes per executed launch of THIS workload only (profiles/traffic.json, written by tools/profile_collect.py from the --pmc passes)"""
    path = os.path.join(root, "profiles", "traffic.json")
    if not os.path.exists(path):
        return
    try:
        tj = json.load(open(path))
        mode_now = 1 if "k_lammuz_ip" in lm_kernel or "k_lammuz_cp" in lm_kernel else 0
        for wl in tj.get("workloads", {}).values():
            if (wl["n_obs"], wl["horizon"], bool(wl["moving"]), wl.get("lmz_mode", 0)) == (N, T, bool(moving), mode_now):
                r_lm["traffic"], r_su["traffic"] = wl.get("k_lammuz"), wl.get("k_su")
                r_su["traffic_source"] = r_lm["traffic_source"] = tj.get("source")
    except Exception:
        pass


def attach_issue(root, r_su, r_lm, N, T, moving):
    """What actually bounds these kernels: instruction issue, not bytes.  From the SQ counters of the committed profile of this
    workload (profiles/issue.json, written by tools/profile_collect.py from the --pmc passes of tools/profile_round.sh; since round 5 per EXECUTED
    dispatch where the profile says so - `per`: "executed" - else averaged over executed and skipped launches alike, in which case every figure
    is a RATIO of two counters of the same pass):
      ipc_per_wave  = (VALU + SALU + LDS + VMEM wave-instructions) / (4 SQ_WAVE_CYCLES)   (SQ_WAVE_CYCLES counts quad-cycles summed over the waves;
                      1 = a wave issuing every cycle it is resident)
      fp64_frac     = fp64 FLOP (2 FMA + MUL + ADD, x 64 lanes = upper bound) per launch / measured launch time / 78.6 TFLOP/s (vector fp64)
      serial_cycles = k_su only: instructions of ONE wave x 6.4 cycles (measured issue interval of a lone wave, tools/latency_micro.cpp)
                      = the length of the dependent chain the launch walks; serial_frac = that / the measured launch time at 2.4 GHz
      lds_conflict_ratio = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS (cycles; VERDICT r04 #2a)"""
    path = os.path.join(root, "profiles", "issue.json")
    if not os.path.exists(path):
        return
    try:
        ij = json.load(open(path))
        for wl in ij.get("workloads", {}).values():
            if (wl["n_obs"], wl["horizon"], bool(wl["moving"])) != (N, T, bool(moving)):
                continue
            executed = wl.get("per") == "executed"
            for r in (r_su, r_lm):
                c = wl["kernels"].get(r["kernel"].split("+")[0])
                if not c:
                    continue
                insts = c.get("SQ_INSTS_VALU", 0) + c.get("SQ_INSTS_SALU", 0) + c.get("SQ_INSTS_LDS", 0) + c.get("SQ_INSTS_VMEM_RD", 0)
                if c.get("SQ_WAVE_CYCLES"):
                    r["ipc_per_wave"] = round(insts / (4.0 * c["SQ_WAVE_CYCLES"]), 4)
                flop = 64.0 * (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0))
                t_ref = ((r["avg_launch_us"] if executed else r["avg_us_over_all_launches"]) or 0) * 1e-6
                if flop and t_ref:
                    r["fp64_gflops"] = round(flop / t_ref / 1e9, 2)
                    r["fp64_frac"] = round(flop / t_ref / PEAK_FP64_VECTOR, 6)
                if r is r_su and c.get("SQ_WAVES"):
                    per_wave = insts / c["SQ_WAVES"]
                    r["serial_cycles"] = round(per_wave * 6.4)
                    if t_ref:
                        r["serial_frac"] = round(per_wave * 6.4 / (t_ref * 2.4e9), 4)
                if c.get("SQ_ACTIVE_INST_LDS"):
                    r["lds_conflict_ratio"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_ACTIVE_INST_LDS"], 3)
                r["issue_source"] = ij.get("source")
                r["issue_per"] = "executed dispatch" if executed else "dispatch (executed and skipped alike)"
    except Exception:
        pass
