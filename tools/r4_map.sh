#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q > gpurun_out/r4_map_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_map_tests.log
tail -6 gpurun_out/r4_map_tests.log
python tools/abenv.py new= oldmap=MI3D_NO_FUSEDP_MAP=1 oldorder=MI3D_NO_WGRAD_TORDER=1 old=MI3D_NO_FUSEDP_MAP=1,MI3D_NO_WGRAD_TORDER=1 --rounds 3 --bench-args "--graph" 2>&1 | tee gpurun_out/r4_map_ab_graph.log
python tools/abenv.py new= old=MI3D_NO_FUSEDP_MAP=1,MI3D_NO_WGRAD_TORDER=1 --rounds 3 --noprof 2>&1 | tee gpurun_out/r4_map_ab_eager.log
