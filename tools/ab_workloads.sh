#!/bin/bash
# the DANN / distillation / eval steps with a route switched through the environment: tools/ab_workloads.sh MI3D_WIDE_BN=0 [rounds]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sw=$1; rounds=${2:-3}
run() { python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"; }
for wl in dann distill; do
  for r in $(seq $rounds); do
    a=$(run --steps 30 --workload $wl); b=$(env $sw python bench.py --steps 30 --workload $wl --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")
    echo "$wl default $a   $sw $b"
  done
done
