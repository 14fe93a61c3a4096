#!/bin/bash
# Collect the per-round evidence for profiles/ on the GPU box: tools/collect_profiles.sh rNN
# (bench line, rocprofv3 kernel stats of the same command, one-step timeline, whole-step PMC traffic in two passes)
set -e
r=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/final
rm -rf $o && mkdir -p $o
python bench.py > $o/bench_stdout.txt 2> $o/bench_stderr.txt
tail -1 $o/bench_stdout.txt > $o/${r}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python bench.py --no-cpu-baseline > $o/stats_stdout.txt 2>&1
cp $(ls $o/stats/*/*_kernel_stats.csv | head -1) $o/${r}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $o/trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python tools/trace_step.py $o/trace --out $o/${r}_step_timeline.txt > /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline > /dev/null 2>&1
python tools/pmc_traffic.py $o/pmc_fetch $o/pmc_write $o/${r}_pmc_traffic_all_kernels.json --step $o/${r}_pmc_step_traffic.json > $o/pmc_stdout.txt 2>&1
rm -rf $o/stats $o/trace $o/pmc_fetch $o/pmc_write
cat $o/${r}_bench.json | cut -c1-400
head -3 $o/${r}_step_timeline.txt
tail -3 $o/pmc_stdout.txt
