#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in "X=1" "GPU_MAX_HW_QUEUES=1" "GPU_MAX_HW_QUEUES=8" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_USE_STDMUTEX_IN_AMD_MONITOR=1" "HSA_ENABLE_SDMA=0" "AMD_DIRECT_DISPATCH=0"; do
  echo "== $e" | tee -a gpurun_out/r4_poke_env.log
  env $e python tools/poke_step.py 2>&1 | grep "plain" | tee -a gpurun_out/r4_poke_env.log
done
