#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r4_full_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_full_tests.log
tail -3 gpurun_out/r4_full_tests.log
bash tools/collect_profiles.sh r04 > gpurun_out/r4_collect.log 2>&1; tail -20 gpurun_out/r4_collect.log | cut -c1-300
bash tools/r4_shapes.sh > /dev/null 2>&1
cat gpurun_out/r4_shapes.log
