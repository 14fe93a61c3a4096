#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r04 > gpurun_out/r4_collect.log 2>&1; tail -22 gpurun_out/r4_collect.log | cut -c1-400
