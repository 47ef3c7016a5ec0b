cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_fifth
mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload rx_fm --no-parity --blocks 8192 > $O/log$i 2>&1
done
cd $R
python - $O <<'P'
import sys, glob
sys.path.insert(0, 'tools')
from collect_profiles import fold
for f in sorted(glob.glob(sys.argv[1] + '/p*/*/*_counter_collection.csv')):
    d = fold(f)
    for k in ('k_fm_fifth_fused<3, true, false, false>', 'k_fm_decimate<false, true, true, true, true>', 'k_fm_decimate_small<true, 6, 5>'):
        if k in d:
            print(k[:44], {c: round(v / 1e6, 2) for c, v in d[k].items() if not c.startswith('_')})
P
tail -3 $O/log3
