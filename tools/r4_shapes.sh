#!/bin/bash
# round 4 (verdict item 6): other shapes on the final build, and the three launch-shape constants scanned at each of them
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*'.ljust(44), round(d['ms_per_step'],3), 'ms', round(d['value'],1), d['unit'], 'frac', round(d['step_hbm_frac'],3), 'graph', d['config'].get('hipgraph'), 'comm', d['config'].get('comm_path'))"; }
{
echo "# shapes, default mode (eager + aux-stream weight gradients + optimizer tail) and --graph (one hipGraph, one stream)"
for g in "" "--graph"; do
run --steps 40 $g
run --steps 20 --batch 4 $g
run --steps 20 --batch 8 $g
run --steps 20 --size 128 $g
run --steps 10 --size 192 --batch 1 $g
run --steps 40 --dropout 0.1 $g
done
run --steps 20 --workload distill
run --steps 20 --workload dann
run --steps 40 --workload eval
run --steps 40 --workload eval --size 128
run --steps 40 --force-comm
run --steps 40 --force-comm --graph-segments
run --steps 40 --dtype fp32 --steps 10
} 2>&1 | tee gpurun_out/r4_shapes.log
[ -n "$SKIP_CONSTANTS" ] && exit 0
{
for shp in "--size 96 --batch 4" "--size 128 --batch 2" "--size 192 --batch 1"; do
  echo "# constants at: $shp (default 128 / 128 / 288)"
  python tools/abenv.py base= ks64=MI3D_KS_TARGET=64 ks256=MI3D_KS_TARGET=256 ksb64=MI3D_KS_TARGET_BWD=64 ksb256=MI3D_KS_TARGET_BWD=256 \
      wg192=MI3D_FUSED_WG_TARGET=192 wg384=MI3D_FUSED_WG_TARGET=384 --rounds 2 --steps 20 --noprof --bench-args "$shp"
done
} 2>&1 | tee gpurun_out/r4_shape_constants.log
