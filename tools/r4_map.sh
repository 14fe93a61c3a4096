#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -x -q > gpurun_out/r4_map_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_map_tests.log
tail -4 gpurun_out/r4_map_tests.log
python tools/abenv.py new= old=MI3D_NO_WIDE_STORE=1 --rounds 3 2>&1 | tee gpurun_out/r4_widestore_ab.log
