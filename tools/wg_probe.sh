#!/bin/bash
# stand-alone backward conv kernels: exactness tests, then per-shape kernel medians (rocprofv3 kernel trace of tools/time_conv3.py --bwd)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/wg; rm -rf $o; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -q -k "conv3" > $o/tests.txt 2>&1 || { tail -30 $o/tests.txt; exit 1; }
MI3D_NO_FUSED_BWD=1 timeout -k 10 300 python -m pytest tests/test_gpu_round2.py -q -k "backward_kernels_of_the_step_exact or fused_persist_16to32" >> $o/tests.txt 2>&1 || { tail -30 $o/tests.txt; exit 1; }
grep -E "passed|failed" $o/tests.txt
shapes="32 16 96 2  16 16 96 2  32 32 48 2  64 32 48 2  64 64 24 2  128 64 24 2  128 128 12 2  256 128 12 2  256 256 6 2"
export MI3D_NO_FUSED_BWD=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/p -- python tools/time_conv3.py --bwd $shapes > $o/time.txt 2>&1
grep conv3 $o/time.txt
python - <<'P'
import csv,glob,collections
f=glob.glob('gpurun_out/wg/p/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
agg=collections.OrderedDict()
for r in rows:
    k=(r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:46], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['Workgroup_Size_X'])
    agg.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,d in agg.items():
    if 'wgrad' in k[0] or 'conv3' in k[0]:
        d=sorted(d); print(f"  {k[0]:46s} grid {k[1]:>7},{k[2]:>3},{k[3]:>3} wg {k[4]:>4}  n {len(d):3d} median {d[len(d)//2]:7.2f}")
P
