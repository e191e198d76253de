"""Same-box A/B of builds of librda_hip.so on the headline loop (the reference's default protocol, re-sorted every tick):
    python tools/ab_headline.py [--rounds 2] [--steps 60] [--extra "--n-obs 2000"] base new ...
Every tag is tools/_bin/librda_hip_<tag>.so (`cur` = the in-tree build), selected with RDA_HIP_SO; one `bench.py --only-headline` process per
tag and round, interleaved.  Prints value, k_su / LamMuZ microseconds per executed launch and the interior-point iterations per step."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tags", nargs="+")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--extra", default="")
    args = ap.parse_args()
    for rnd in range(args.rounds):
        for tag in args.tags:
            env = dict(os.environ)
            if tag != "cur":
                env["RDA_HIP_SO"] = os.path.join(ROOT, "tools", "_bin", f"librda_hip_{tag}.so")
            pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--only-headline", "--steps", str(args.steps), "--warmup", str(args.warmup)] + args.extra.split(),
                                capture_output=True, text=True, env=env)
            lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
            if not lines:
                print(f"{tag} round {rnd}: FAILED rc={pr.returncode} {pr.stderr[-300:]!r}", flush=True)
                continue
            j = json.loads(lines[-1])
            r, r2 = j["roofline"], j["roofline_secondary"]
            su, lm = (r, r2) if r["kernel"].startswith("k_su") else (r2, r)
            print(f"{tag:10s} round {rnd}: value {j['value']:9.2f}  median/s {1e3 / j['median_ms_per_step']:9.2f}  su {su['avg_launch_us']:7.2f} us  lmz {lm['avg_launch_us']:6.2f} us  "
                  f"ipm/step {j['residuals']['su_interior_point_iters_per_step']:6.2f}  admm {j['mean_admm_iters']:.2f}  2nd window {((j.get('second_window') or {}).get('steps_per_s'))}", flush=True)


if __name__ == "__main__":
    main()
