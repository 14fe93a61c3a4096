#!/usr/bin/env python3
"""A/B of route switches on ONE box: wall-clock ms/step (bench.py, hipGraph) and per-kernel time per step (rocprofv3
--kernel-trace --stats) for several environment settings.

    python tools/abenv.py base= wring=MI3D_CONV_WRING=1 nobn=MI3D_NO_BN_ONEPASS=1 [--steps 40] [--rounds 2] [--noprof]
Each argument is name=ENV=VAL[,ENV=VAL...] (empty = default build).  The first setting is the reference of the kernel diff."""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def main():
    args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 40
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2
    extra = sys.argv[sys.argv.index("--bench-args") + 1].split() if "--bench-args" in sys.argv else []
    cfgs = []
    for a in args:
        name, _, envs = a.partition("=")
        env = dict(e.split("=", 1) for e in envs.split(",") if e)
        cfgs.append((name, env))
    wall = {n: [] for n, _ in cfgs}
    clk = {}
    for _ in range(rounds):
        for n, env in cfgs:
            e = dict(os.environ, **env)
            out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-roofline", "--sclk", "--steps", str(steps)] + extra,
                                 cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
            try:
                line = json.loads(out.stdout.strip().splitlines()[-1])
                wall[n].append(line["ms_per_step"])
                clk.setdefault(n, []).append((line.get("sclk_mhz") or {}).get("mean"))
            except Exception:      # noqa: BLE001
                print(n, "bench failed:", out.stderr[-400:])
                wall[n].append(float("nan"))
    for n, _ in cfgs:
        print(f"wall {n:12s} " + "  ".join(f"{v:.4f}" for v in wall[n]) + " ms/step" +
              "   sclk " + " ".join(f"{c:.0f}" if c else "-" for c in clk.get(n, [])) + " MHz", flush=True)
    if "--noprof" in sys.argv:
        return
    per = {}
    nsteps = 20
    for n, env in cfgs:
        d = os.path.join(ROOT, "gpurun_out", "abenv_" + n)
        shutil.rmtree(d, ignore_errors=True)
        e = dict(os.environ, **env)
        e["TMPDIR"] = "/tmp"
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", sys.executable,
                        os.path.join(ROOT, "bench.py"), "--steps", str(nsteps), "--warmup", "3", "--no-cpu-baseline", "--no-roofline"] + extra,
                       cwd="/tmp", env=e, capture_output=True, text=True, timeout=600)
        f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        if not f:
            print(n, "no kernel stats")
            continue
        per[n] = {}
        for r in csv.DictReader(open(f[0])):
            k = short(r["Name"])
            c, tot = per[n].get(k, (0, 0.0))
            per[n][k] = (c + int(r["Calls"]), tot + float(r["TotalDurationNs"]) / 1e3)
        shutil.rmtree(d, ignore_errors=True)
    if not per:
        return
    ref = cfgs[0][0]
    div = nsteps + 3 + 1        # timed + warm-up + the capture's eager warm-up pass
    names = sorted(set().union(*[set(v) for v in per.values()]), key=lambda k: -per.get(ref, {}).get(k, (0, 0))[1])
    print(f"\nper-kernel us/step (calls/step), reference = {ref}")
    tot = {n: 0.0 for n in per}
    for k in names:
        row = []
        for n, _ in cfgs:
            c, t = per.get(n, {}).get(k, (0, 0.0))
            tot[n] = tot.get(n, 0.0) + t / div
            row.append(f"{t / div:8.1f} ({c / div:4.1f})")
        base = per.get(ref, {}).get(k, (0, 0.0))[1] / div
        if any(abs(per.get(n, {}).get(k, (0, 0.0))[1] / div - base) > 0.8 for n, _ in cfgs):
            print(f"{k:72s} " + "  ".join(row))
    print(f"{'TOTAL kernel time':72s} " + "  ".join(f"{tot[n]:8.1f}       " for n, _ in cfgs))


if __name__ == "__main__":
    main()
