#!/bin/bash
# other workloads / shapes / launch modes of the same build (one line each): tools/shapes.sh
run() { python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*'.ljust(44), round(d['ms_per_step'],3), 'ms', round(d['value'],1), d['unit'], 'graph', d['config'].get('hipgraph'), 'comm', d['config'].get('comm_path'))"; }
run --steps 40
run --steps 40 --no-graph
run --steps 40 --force-comm --graph-segments
run --steps 40 --force-comm
run --steps 20 --batch 4
run --steps 20 --batch 8
run --steps 20 --size 128
run --steps 10 --size 192 --batch 1
run --steps 40 --dropout 0.1
run --steps 20 --workload distill
run --steps 20 --workload dann
run --steps 40 --workload eval
run --steps 40 --workload eval --size 128
