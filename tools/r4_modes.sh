set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
for mode in "" "--no-graph"; do
  for aux in "" "--no-aux-wgrad"; do
    echo "mode=[$mode] aux=[$aux]" >> $o/r4_modes.log
    python bench.py --no-cpu-baseline --no-roofline --steps 40 $mode $aux 2>/dev/null | tail -1 | cut -c1-200 >> $o/r4_modes.log
  done
done
cat $o/r4_modes.log
rocprofv3 --kernel-trace --output-format csv -d $o/tr_g -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python tools/trace_streams.py $o/tr_g --out $o/r4_defer_graph_streams.txt
rocprofv3 --kernel-trace --output-format csv -d $o/tr_e -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > /dev/null 2>&1
python tools/trace_streams.py $o/tr_e --out $o/r4_defer_eager_streams.txt
rm -rf $o/tr_g $o/tr_e
