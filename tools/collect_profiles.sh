#!/bin/bash
# Collect the per-round evidence for profiles/ on the GPU box: tools/collect_profiles.sh rNN
# (bench line, rocprofv3 kernel stats of the same command, one-step timeline, whole-step PMC traffic in two passes,
#  the per-launch roofline table, the dominant kernel's traffic record that bench.py reports)
set -e
r=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/final
rm -rf $o && mkdir -p $o
python bench.py > $o/bench_stdout.txt 2> $o/bench_stderr.txt
tail -1 $o/bench_stdout.txt > $o/${r}_bench.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python bench.py --no-cpu-baseline > $o/stats_stdout.txt 2>&1
cp $(ls $o/stats/*/*_kernel_stats.csv | head -1) $o/${r}_bench_kernel_stats.csv
echo "stats done"
rocprofv3 --kernel-trace --output-format csv -d $o/trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python tools/trace_step.py $o/trace --out $o/${r}_step_timeline.txt > /dev/null
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > /dev/null 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > /dev/null 2>&1
echo "pmc write done"
python tools/pmc_traffic.py $o/pmc_fetch $o/pmc_write $o/${r}_pmc_traffic_all_kernels.json --step $o/${r}_pmc_step_traffic.json > $o/pmc_stdout.txt 2>&1
python tools/kernel_roofline.py $o/${r}_step_timeline.txt $o/${r}_pmc_step_traffic.json --out $o/${r}_kernel_roofline.csv > $o/${r}_kernel_roofline_levels.txt
python - "$o" "$r" <<'P'
import csv, json, sys
o, r = sys.argv[1], sys.argv[2]
name = "conv3_bwd_fused_persist_kernel<2, 1>"
step = json.load(open(f"{o}/{r}_pmc_step_traffic.json"))["per_dispatch"]
hb = [d["fetch_bytes_corrected"] + d["write_bytes"] for d in step if name in d["kernel"]]
us = [float(x["AverageNs"]) / 1e3 for x in csv.DictReader(open(f"{o}/{r}_bench_kernel_stats.csv")) if name in x["Name"]]
rec = {"kernel": name, "hbm_bytes_per_launch": sum(hb) / max(len(hb), 1), "ms_per_launch_when_measured": (us[0] / 1e3) if us else None,
       "method": f"{r}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --no-graph`, the dispatch of this kernel in the "
                 "last complete step; FETCH_SIZE x2 (gfx950 wide-read correction), KiB units (tools/pmc_traffic.py); time = its average in "
                 f"{r}_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py`)"}
json.dump(rec, open(f"{o}/roofline_kernel_traffic.json", "w"), indent=1)
print(rec)
P
rm -rf $o/stats $o/trace $o/pmc_fetch $o/pmc_write
cat $o/${r}_bench.json | cut -c1-600
head -3 $o/${r}_step_timeline.txt
tail -3 $o/pmc_stdout.txt
cat $o/${r}_kernel_roofline_levels.txt
