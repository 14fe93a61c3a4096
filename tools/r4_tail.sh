#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q > gpurun_out/r4_tail_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_tail_tests.log
tail -15 gpurun_out/r4_tail_tests.log
python tools/abenv.py tail= notail=MI3D_NO_OPT_TAIL=1 fe2=MI3D_DEFER_FORK_EACH=2 fe2_notail=MI3D_DEFER_FORK_EACH=2,MI3D_NO_OPT_TAIL=1 \
   late=MI3D_G1_FORK_LATE=1 late_fe2=MI3D_G1_FORK_LATE=1,MI3D_DEFER_FORK_EACH=2 chain=MI3D_NO_DEFER_WGRAD=1 --rounds 4 --noprof 2>&1 | tee gpurun_out/r4_tail_ab2.log
MI3D_DEFER_FORK_EACH=2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_e -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python tools/trace_streams.py gpurun_out/tr_e --out gpurun_out/r4_tail_fe2_streams.txt
rm -rf gpurun_out/tr_e
