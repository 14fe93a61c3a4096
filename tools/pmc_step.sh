#!/bin/bash
# SQ / LDS / L2 counters for every kernel of the training step (eager launch, one pass per counter set):
#   tools/pmc_step.sh [out-file]      -> per (kernel, grid): median of each counter over the dispatches of 3 steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/pmcstep; rm -rf $o; mkdir -p $o
out=${1:-gpurun_out/pmc_step.txt}
i=0
for pass in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL" \
  "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $o/p$i -- python bench.py --no-graph --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $o/log$i.txt 2>&1 || echo "pass $i failed: $(tail -2 $o/log$i.txt)"
done
python - "$out" <<'P'
import csv,glob,collections,sys
agg=collections.OrderedDict(); dur=collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/pmcstep/p*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:60], int(r['Grid_Size']))
        agg.setdefault(k,collections.OrderedDict()).setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for f in sorted(glob.glob('gpurun_out/pmcstep/p1/*/*kernel_trace.csv')):
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:60], int(r['Grid_Size_X'])*int(r['Grid_Size_Y'])*int(r['Grid_Size_Z']))
        dur.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
med=lambda v: sorted(v)[len(v)//2]
w=open(sys.argv[1],'w')
def P(*a):
    s=' '.join(str(x) for x in a); print(s); w.write(s+'\n')
P("# per (kernel, grid threads): median over dispatches; SQ_*_CYCLES / WAIT / ACTIVE in quad-cycles summed over waves; us = under the counter pass")
for k,d in agg.items():
    g=lambda c: med(d[c]) if c in d else float('nan')
    wc=g('SQ_WAVE_CYCLES')
    if not wc or wc!=wc: continue
    us=med(dur.get(k,[float('nan')]))
    P(f"{k[0]:60s} grid {k[1]:>8d} n {len(d['SQ_WAVE_CYCLES']):3d} us {us:7.1f} | waves {g('SQ_WAVES'):7.0f} active {g('SQ_ACTIVE_INST_ANY')/wc:5.2f} wait_any {g('SQ_WAIT_ANY')/wc:5.2f} wait_inst {g('SQ_WAIT_INST_ANY')/wc:5.2f} (lds {g('SQ_WAIT_INST_LDS')/wc:4.2f}) | per wave: valu {g('SQ_INSTS_VALU')/g('SQ_WAVES'):7.0f} salu {g('SQ_INSTS_SALU')/g('SQ_WAVES'):6.0f} lds {g('SQ_INSTS_LDS')/g('SQ_WAVES'):6.0f} mfma {g('SQ_INSTS_MFMA')/g('SQ_WAVES'):6.0f} vmem {(g('SQ_INSTS_VMEM_RD')+g('SQ_INSTS_VMEM_WR'))/g('SQ_WAVES'):5.0f} | lds conflict/active {g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1):4.2f} mfma_busy_cyc {g('SQ_VALU_MFMA_BUSY_CYCLES'):11.0f} | L2 req {g('TCP_TCC_READ_REQ_sum'):9.0f} hit {g('TCC_HIT_sum'):9.0f} miss {g('TCC_MISS_sum'):9.0f} ea_rd {g('TCC_EA0_RDREQ_sum'):9.0f}")
P
