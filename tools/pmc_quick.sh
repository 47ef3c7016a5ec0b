# tools/pmc_quick.sh TAG [bench args] -- one PMC pass (VALU/issue counters) over a short rx_fm bench; prints per-kernel figures
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcq_${1:-x}
mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload rx_fm --no-parity --blocks 8192 > $O/log 2>&1
cd $R
python - $O <<'P'
import sys, glob
sys.path.insert(0, 'tools')
from collect_profiles import fold
f = glob.glob(sys.argv[1] + '/pmc/*/*_counter_collection.csv')[0]
for k, v in sorted(fold(f).items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0)):
    if v.get('SQ_INSTS_VALU', 0) < 5e6: continue
    cyc = v['GRBM_GUI_ACTIVE'] / 8.0    # summed over 8 XCDs
    print(k[:50].ljust(50), 'grid', v['_grid_size'], 'valu %.1fM' % (v['SQ_INSTS_VALU'] / 1e6), 'cyc/xcd %.3fM (%.0f us)' % (cyc / 1e6, cyc / 2400.0),
          'ipc/simd %.3f' % (v['SQ_INSTS_VALU'] / (cyc * 1024.0)), 'lds %.1fM confl %.2f' % (v.get('SQ_INSTS_LDS', 0) / 1e6, v.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, v.get('SQ_LDS_IDX_ACTIVE', 1))))
P
