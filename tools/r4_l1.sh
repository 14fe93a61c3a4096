#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/abenv.py base= cob1=MI3D_L1_COB1=1 --rounds 3 2>&1 | tee gpurun_out/r4_l1_ab.log
