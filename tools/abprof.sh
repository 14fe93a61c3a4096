#!/bin/bash
# per-kernel A/B of two builds under rocprofv3 on the same box: tools/abprof.sh <old.so> ; prints kernels whose time/step moved
old=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ab_new gpurun_out/ab_old
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_new -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
MI3D_LIB_PATH=$old rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_old -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python - <<'P'
import csv,glob
def load(d):
    f=glob.glob(f'gpurun_out/{d}/*/*_kernel_stats.csv')[0]
    return {r['Name']:(int(r['Calls']),float(r['AverageNs'])/1e3) for r in csv.DictReader(open(f))}
o,n=load('ab_old'),load('ab_new')
tot=0
for k in sorted(set(o)|set(n), key=lambda k:-(o.get(k,(0,0))[0]*o.get(k,(0,0))[1])):
    a=o.get(k,(0,0)); b=n.get(k,(0,0))
    d=(b[0]*b[1]-a[0]*a[1])/24
    tot+=d
    if abs(d)>1.0: print(f"{k[:100]:100s} old {a[1]:7.2f} new {b[1]:7.2f} d/step {d:7.1f}")
print("total d/step", round(tot,1))
P
