cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_ds6
mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python $R/tools/ds6_probe.py 8192 > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python $R/tools/ds6_probe.py 8192 > $O/b.log 2>&1
cd $R
python - <<'P'
import csv,glob
from collections import defaultdict
for d in ('a','b'):
    f=glob.glob('gpurun_out/pmc_ds6/%s/*/*counter_collection.csv'%d)
    if not f: print('no file',d); continue
    rows=defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if not k.startswith('k_fm_dec'): continue
        rows[(k,r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
    seen=defaultdict(int)
    for (k,d_),c in sorted(rows.items(), key=lambda x:int(x[0][1])):
        seen[k]+=1
        if seen[k]%6!=1: continue
        g=c.get('GRBM_GUI_ACTIVE',0)/8
        print(k[:44], 'us %.0f'%(g/2.4e3), {n:'%.3g'%v for n,v in c.items() if n!='GRBM_GUI_ACTIVE'})
P
