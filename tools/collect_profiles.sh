#!/bin/bash
# Collect the per-round evidence for profiles/ on the GPU box: tools/collect_profiles.sh rNN
#   default mode (eager launches, weight gradients + optimizer tail on the aux stream):
#     bench line, rocprofv3 kernel stats of the same command, one-step two-stream timeline, whole-step PMC traffic (two passes),
#     the roofline candidates' traffic records that bench.py reports (profiles/roofline_kernel_traffic.json)
#   single-stream mode (--graph: one hipGraph, the round-3 route):
#     bench line, one-step timeline, per-launch PMC traffic (eager single stream: --no-aux-wgrad) and the per-launch roofline table
set -e
r=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/final
rm -rf $o && mkdir -p $o
python bench.py > $o/bench_stdout.txt 2> $o/bench_stderr.txt
tail -1 $o/bench_stdout.txt > $o/${r}_bench.json
python bench.py --graph --no-cpu-baseline > $o/bench_graph_stdout.txt 2>> $o/bench_stderr.txt
tail -1 $o/bench_graph_stdout.txt > $o/${r}_bench_graph.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python bench.py --no-cpu-baseline > $o/stats_stdout.txt 2>&1
cp $(ls $o/stats/*/*_kernel_stats.csv | head -1) $o/${r}_bench_kernel_stats.csv
echo "stats done"
rocprofv3 --kernel-trace --output-format csv -d $o/trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python tools/trace_streams.py $o/trace --out $o/${r}_step_streams.txt > /dev/null
rocprofv3 --kernel-trace --output-format csv -d $o/trace_g -- python bench.py --graph --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python tools/trace_step.py $o/trace_g --out $o/${r}_step_timeline.txt > /dev/null
echo "trace done"
# per-launch traffic of the single-stream route (dispatch order = the timeline's order)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python bench.py --steps 3 --warmup 2 --no-aux-wgrad --no-cpu-baseline --no-roofline > /dev/null 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python bench.py --steps 3 --warmup 2 --no-aux-wgrad --no-cpu-baseline --no-roofline > /dev/null 2>&1
echo "pmc write done"
python tools/pmc_traffic.py $o/pmc_fetch $o/pmc_write $o/${r}_pmc_traffic_all_kernels.json --step $o/${r}_pmc_step_traffic.json > $o/pmc_stdout.txt 2>&1
python tools/kernel_roofline.py $o/${r}_step_timeline.txt $o/${r}_pmc_step_traffic.json --out $o/${r}_kernel_roofline.csv > $o/${r}_kernel_roofline_levels.txt
# the default route's own dispatches (aux-stream weight gradients, stand-alone input gradients): whole step + the roofline candidates
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch_d -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write_d -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python tools/pmc_traffic.py $o/pmc_fetch_d $o/pmc_write_d $o/${r}_pmc_traffic_all_kernels_default.json --step $o/${r}_pmc_step_traffic_default.json >> $o/pmc_stdout.txt 2>&1
echo "pmc default done"
python - "$o" "$r" <<'P'
import json, sys
sys.path.insert(0, ".")
import bench
o, r = sys.argv[1], sys.argv[2]
step = json.load(open(f"{o}/{r}_pmc_step_traffic_default.json"))["per_dispatch"]
live = {}
try:
    line = json.load(open(f"{o}/{r}_bench.json"))
    rf = line.get("roofline") or {}
    for k in [rf] + rf.get("others", []):
        if k.get("key"):
            live[k["key"]] = k["ms_per_launch"]
except Exception as e:      # noqa: BLE001
    print("no bench line:", e)
recs = {}
for cd in bench.roofline_candidates(96, 2):
    hits = [d for d in step if cd["kernel"].replace(" ", "") in d["kernel"].replace(" ", "")]
    if len(hits) <= cd["ordinal"]:
        continue
    d = hits[cd["ordinal"]]
    recs[cd["key"]] = {"kernel": cd["kernel"], "layer": cd["layer"], "dispatch_ordinal_in_step": cd["ordinal"],
                       "hbm_bytes_per_launch": d["fetch_bytes_corrected"] + d["write_bytes"],
                       "algorithmic_bytes_per_launch": cd["bytes"], "ms_per_launch_when_measured": live.get(cd["key"])}
rec = {"records": recs,
       "method": f"{r}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py` (default mode: eager, aux-stream weight "
                 "gradients; counter collection serialises the dispatches), the dispatch of each kernel in the last complete step; FETCH_SIZE x2 "
                 "(gfx950 wide-read correction), KiB units (tools/pmc_traffic.py); ms_per_launch_when_measured = the live HIP-event time of the same "
                 f"launch in {r}_bench.json (inside training steps, neighbours running)"}
json.dump(rec, open(f"{o}/roofline_kernel_traffic.json", "w"), indent=1)
for k, v in recs.items():
    print(k, round(v["hbm_bytes_per_launch"] / 1e6, 1), "MB counter /", round(v["algorithmic_bytes_per_launch"] / 1e6, 1), "MB algorithmic", v["ms_per_launch_when_measured"])
P
rm -rf $o/stats $o/trace $o/trace_g $o/pmc_fetch $o/pmc_write $o/pmc_fetch_d $o/pmc_write_d
cat $o/${r}_bench.json | cut -c1-900
cat $o/${r}_bench_graph.json | cut -c1-300
head -3 $o/${r}_step_timeline.txt
tail -4 $o/pmc_stdout.txt
cat $o/${r}_kernel_roofline_levels.txt
