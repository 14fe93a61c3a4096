#!/bin/bash
# round 4: where does the two-stream backward gain / lose?  eager launches (the hipGraph executor spreads a forked graph
# over three queues and runs it at half speed: profiles/r04_defer_graph_streams.txt), wall clock only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/abenv.py chain=MI3D_NO_DEFER_WGRAD=1 \
  lazy3= nolazy=MI3D_NO_LAZY_AUX=1 lazy2=MI3D_AUX_DRAIN=2 lazy1=MI3D_AUX_DRAIN=1 lazy5=MI3D_AUX_DRAIN=5 \
  lazy3_low=MI3D_AUX_PRIO=low \
  --rounds 3 --noprof --bench-args "--no-graph" 2>&1 | tee gpurun_out/r4_aux_scan3.log
python tools/abenv.py graphchain=MI3D_NO_DEFER_WGRAD=1 --rounds 3 --noprof 2>&1 | tee -a gpurun_out/r4_aux_scan3.log
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_e -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > /dev/null 2>&1
python tools/trace_streams.py gpurun_out/tr_e --out gpurun_out/r4_defer_eager_streams_lazy.txt
rm -rf gpurun_out/tr_e
