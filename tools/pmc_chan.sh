# tools/pmc_chan.sh TAG -- the channeliser's kernels: trace durations, issue counters, LDS counters (separate passes), printed per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_chan}
mkdir -p $O
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
LDS="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
CMD="python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --workload chan --variants none --no-parity"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1
i=0
for set in "$SQ" "$LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $CMD > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python - $O <<'P'
import csv,glob,sys,collections
O=sys.argv[1]
for f in glob.glob(O+'/trace/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_ch' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(O+'/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'k_ch' not in k: continue
        k=k[:40]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
for k in acc:
    print(k, {c: round(v/n[k][c]) for c,v in acc[k].items()})
P
