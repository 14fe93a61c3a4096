#!/bin/bash
# SQ / LDS / L2 counters of the stand-alone conv kernels (one pass per counter set): tools/pmc_kernel.sh "<time_conv3 args>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/pmc; rm -rf $o; mkdir -p $o
rocprofv3 -L > $o/counters.txt 2>&1
args=${1:---bwd 64 32 48 2 64 64 24 2}
i=0
for pass in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" \
  "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $o/p$i -- python tools/time_conv3.py $args > $o/log$i.txt 2>&1 || echo "pass $i failed: $(tail -2 $o/log$i.txt)"
done
python - <<'P'
import csv,glob,collections
agg=collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/pmc/p*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:40], r['Grid_Size'])
        agg.setdefault(k,collections.OrderedDict()).setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for k,d in agg.items():
    if not any(s in k[0] for s in ('wgrad','conv3_mfma','fused')): continue
    print(k)
    for c,v in d.items():
        v=sorted(v); print(f"    {c:32s} n {len(v):3d} median {v[len(v)//2]:16.0f}")
P
