#!/usr/bin/env python3
"""Two-stream view of ONE training step from a rocprofv3 --kernel-trace CSV: like tools/trace_step.py, with the hardware
queue / stream of every launch, the end time, and how many other kernels were running when it started.

    python tools/trace_streams.py gpurun_out/trace [--out file]
"""
import csv
import glob
import sys

from trace_step import short


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
    qs = sorted({r.get("Queue_Id", "?") for r in step})
    lines = [f"# {f}", f"# kernels {len(step)} span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us; queues {qs}",
             "#  idx  start_us    end_us   dur_us  q  stream  running  kernel"]
    busy = {q: 0.0 for q in qs}
    for i, r in enumerate(step):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        running = sum(1 for o in step if int(o["Start_Timestamp"]) < s < int(o["End_Timestamp"]))
        q = r.get("Queue_Id", "?")
        busy[q] += (e - s) / 1e3
        lines.append(f"{i:4d} {(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {qs.index(q)}  {r.get('Stream_Id', '?'):>5s}  {running:3d}  "
                     f"{short(r['Kernel_Name']):60s} grid {r.get('Grid_Size_X', '?')} wg {r.get('Workgroup_Size_X', '?')}")
    lines.append("# busy us per queue: " + ", ".join(f"{qs.index(q)}: {v:.1f}" for q, v in busy.items()))
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    else:
        print(text)


if __name__ == "__main__":
    main()
