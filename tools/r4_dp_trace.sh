#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/abenv.py one= extra_high=MI3D_BENCH_EXTRA_STREAM=high extra_normal=MI3D_BENCH_EXTRA_STREAM=normal extra_low=MI3D_BENCH_EXTRA_STREAM=low --rounds 3 --noprof --bench-args "--no-aux-wgrad" 2>&1 | tee gpurun_out/r4_dp_ab7.log
python tools/abenv.py graph_one= graph_extra=MI3D_BENCH_EXTRA_STREAM=high --rounds 2 --noprof --bench-args "--graph" 2>&1 | tee -a gpurun_out/r4_dp_ab7.log
