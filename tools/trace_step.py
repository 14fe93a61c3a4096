#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: every kernel node of the last complete step
(steps are delimited by conv3_c1_fwd_mfma_kernel, the first conv of a forward: once per step) with its duration and the idle gap before it.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 6 --warmup 3 ...
    python tools/trace_step.py gpurun_out/trace [--out profiles/rNN_step_timeline.txt]
"""
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)I", name)
    if m:
        name = m.group(1)
    return name.split("(")[0][:60]


def main():
    d = sys.argv[1]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "conv3_c1_fwd_mfma_kernel" in r["Kernel_Name"]]
    if len(starts) < 3:
        print("not enough steps in trace", len(starts))
        return
    a, b = starts[-2], starts[-1]
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    lines = []
    tot_d = tot_g = 0.0
    prev_end = None
    agg = {}
    for i, r in enumerate(step):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        dur = (e - s) / 1e3
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        prev_end = max(prev_end or e, e)
        nm = short(r["Kernel_Name"])
        g = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        w = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
        lines.append(f"{i:4d} {(s - t0) / 1e3:9.1f} {dur:8.1f} {gap:6.1f}  {nm:60s} grid {g} wg {w}")
        tot_d += dur
        tot_g += max(gap, 0.0)
        k = agg.setdefault(nm, [0, 0.0, 0.0])
        k[0] += 1; k[1] += dur; k[2] += max(gap, 0.0)
    span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
    hdr = [f"# {f}", f"# kernels {len(step)}  span {span:.1f} us  sum(dur) {tot_d:.1f} us  sum(gaps) {tot_g:.1f} us",
           "#  idx   start_us   dur_us gap_us  kernel"]
    summ = ["", "# by kernel: count, total dur us, total gap-before us"]
    for nm, (c, dsum, gsum) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        summ.append(f"# {nm:60s} {c:4d} {dsum:9.1f} {gsum:8.1f}")
    text = "\n".join(hdr + lines + summ)
    if out:
        open(out, "w").write(text + "\n")
    print("\n".join(hdr[:2] + summ[:40]))


if __name__ == "__main__":
    main()
