#!/bin/bash
# eight-wave weight gradient: exactness through the stand-alone route, then per-kernel time against the four-wave kernel.
# Needs profiles/r03_exp_wgrad8.patch applied (the experiment was not kept; MI3D_WGRAD8 exists only in that patch).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/wg8; rm -rf $o; mkdir -p $o
MI3D_NO_FUSED_BWD=1 MI3D_WGRAD8=1 timeout -k 10 300 python -m pytest tests/test_gpu_round2.py -q -k "backward_kernels_of_the_step_exact or fused_persist_16to32" > $o/tests.txt 2>&1 || { tail -30 $o/tests.txt; exit 1; }
tail -3 $o/tests.txt
shapes="32 32 48 2  64 32 48 2  64 64 24 2  128 64 24 2  128 128 12 2  256 128 12 2  256 256 6 2"
for v in 4 8; do
  if [ $v = 8 ]; then export MI3D_WGRAD8=1; else unset MI3D_WGRAD8; fi
  MI3D_NO_FUSED_BWD=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/p$v -- python tools/time_conv3.py --bwd $shapes > $o/time$v.txt 2>&1
  cat $o/time$v.txt | grep conv3
done
python - <<'P'
import csv,glob
for v in (4,8):
    f=glob.glob(f'gpurun_out/wg8/p{v}/*/*_kernel_stats.csv')[0]
    print("waves",v)
    for r in csv.DictReader(open(f)):
        if 'wgrad' in r['Name'] or 'slab' in r['Name']:
            print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.2f} us")
P
