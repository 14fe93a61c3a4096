#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately: they do not fit one pass on
gfx950) into per-kernel HBM traffic, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
  * counters are in KiB (bytes = value * 1024);
  * on gfx950 FETCH_SIZE reports exactly half the bytes of wide (16 B/lane) coalesced streaming reads -> x2 for the
    kernels whose reads are all 16-B vector loads (every kernel here stages/streams with dwordx4 loads);
  * WRITE_SIZE is exact for 16-B-per-lane stores.
Usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [--step <step_out.json>]
  --step: additionally sum the dispatches of the LAST complete training step (delimited by conv3_c1_fwd_mfma_kernel, the first
  conv of a forward) and write per-dispatch + whole-step HBM bytes (to compare with the 4.84 GB algorithmic model)."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return agg


def per_dispatch(d, counter):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            e = agg.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0, int(r["Grid_Size"])])
            e[1] += float(r["Counter_Value"])
    return agg


def step_summary(fd, wd, out):
    fa, wa = per_dispatch(fd, "FETCH_SIZE"), per_dispatch(wd, "WRITE_SIZE")

    def last_step(agg):
        ids = sorted(agg)
        starts = [i for i in ids if "conv3_c1_fwd_mfma" in agg[i][0]]
        return [i for i in ids if starts[-2] <= i < starts[-1]]

    fi, wi = last_step(fa), last_step(wa)
    assert len(fi) == len(wi), (len(fi), len(wi))
    rows, tf, tw = [], 0.0, 0.0
    for a, b in zip(fi, wi):
        assert fa[a][0] == wa[b][0]
        f, w = 2 * fa[a][1] * 1024, wa[b][1] * 1024
        tf += f; tw += w
        rows.append({"kernel": fa[a][0][:120], "grid_threads": fa[a][2], "fetch_bytes_corrected": f, "write_bytes": w})
    res = {"note": "last complete step of `bench.py --no-graph` under rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in separate "
                   "passes); FETCH_SIZE x2 (gfx950 wide-read under-count) x 1024 (KiB); Infinity-Cache hits are counted",
           "dispatches": len(rows), "fetch_bytes_corrected": tf, "write_bytes": tw, "hbm_bytes": tf + tw,
           "algorithmic_bytes": 4.842e9, "ratio_to_algorithmic": (tf + tw) / 4.842e9, "per_dispatch": rows}
    json.dump(res, open(out, "w"), indent=1)
    print(f"step: {len(rows)} dispatches, fetch {tf / 1e9:.3f} GB + write {tw / 1e9:.3f} GB = {(tf + tw) / 1e9:.3f} GB "
          f"= {(tf + tw) / 4.842e9:.2f} x algorithmic")


def main():
    if "--step" in sys.argv:
        step_summary(sys.argv[1], sys.argv[2], sys.argv[sys.argv.index("--step") + 1])
    fd, wd, out = sys.argv[1:4]
    fa, wa = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = []
    for k in sorted(fa, key=lambda k: -sum(fa[k])):
        fetch = sum(fa[k]) / len(fa[k]) * 1024
        write = sum(wa.get(k, [0.0])) / max(1, len(wa.get(k, [0.0]))) * 1024
        res.append({"kernel": k[0], "grid_threads": int(k[1]), "launches": len(fa[k]),
                    "fetch_bytes_raw": fetch, "fetch_bytes_corrected": 2 * fetch, "write_bytes": write,
                    "hbm_bytes": 2 * fetch + write})
    json.dump({"note": "per-launch averages; FETCH_SIZE x2 (gfx950 wide-read under-count), KiB units", "kernels": res},
              open(out, "w"), indent=1)
    for r in res[:12]:
        print(f"{r['kernel'][:80]:80s} grid {r['grid_threads']:8d} n={r['launches']:3d} hbm {r['hbm_bytes']/1e6:8.1f} MB")


if __name__ == "__main__":
    main()
