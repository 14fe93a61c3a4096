#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/abenv.py overlap= serial=MI3D_COMM_SERIAL=1 --rounds 3 --noprof --bench-args "--force-comm" 2>&1 | tee gpurun_out/r4_comm_serial.log
python tools/abenv.py nocomm= --rounds 3 --noprof --bench-args "--no-aux-wgrad" 2>&1 | tee -a gpurun_out/r4_comm_serial.log
python -m pytest tests/test_gpu_dp.py -x -q 2>&1 | tail -2
