#!/bin/bash
# round 4: apply-on-load at the deep levels -- tests, then A/B (eager + aux weight gradients; one graph, single stream)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round4.py -x -q > gpurun_out/r4_aol_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_aol_tests.log
tail -25 gpurun_out/r4_aol_tests.log
python tools/abenv.py aol= noaol=MI3D_NO_APPLY_ON_LOAD=1 --rounds 3 --bench-args "--no-graph" 2>&1 | tee gpurun_out/r4_aol_ab_eager.log
python tools/abenv.py aol=MI3D_NO_DEFER_WGRAD=1 noaol=MI3D_NO_DEFER_WGRAD=1,MI3D_NO_APPLY_ON_LOAD=1 --rounds 3 --noprof 2>&1 | tee gpurun_out/r4_aol_ab_graph.log
