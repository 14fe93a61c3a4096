#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r4_full_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_full_tests.log
tail -4 gpurun_out/r4_full_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r4_smoke.log; tail -3 gpurun_out/r4_smoke.log
python tools/abenv.py new= --rounds 2 2>&1 | grep -E "wall|head_loss" | tee gpurun_out/r4_headwide.log
