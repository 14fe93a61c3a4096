#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately: they do not fit one pass on
gfx950) into per-kernel HBM traffic, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
  * counters are in KiB (bytes = value * 1024);
  * on gfx950 FETCH_SIZE reports exactly half the bytes of wide (16 B/lane) coalesced streaming reads -> x2 for the
    kernels whose reads are all 16-B vector loads (every kernel here stages/streams with dwordx4 loads);
  * WRITE_SIZE is exact for 16-B-per-lane stores.
Usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
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


def main():
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
