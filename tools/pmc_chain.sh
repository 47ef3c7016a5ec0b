# tools/pmc_chain.sh TAG BLOCKS DS KERNEL_SUBSTRING -- rocprofv3 counter passes over tools/chain_once.py, per-kernel averages printed and kept in gpurun_out/TAG/pmc.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python $R/tools/chain_once.py $2 $3 2 > $O/log$i 2>&1 || echo "pass $i failed: $(tail -2 $O/log$i)"
done
cd $R
python - $O "$4" <<'P'
import sys, glob, json
sys.path.insert(0, 'tools')
from pmc_summary import fold
out = {}
for f in sorted(glob.glob(sys.argv[1] + '/p*/*/*_counter_collection.csv')):
    for k, d in fold(f).items():
        if sys.argv[2] in k or not sys.argv[2]:
            out.setdefault(k, {}).update(d)
for k, d in out.items():
    if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_INSTS_VALU"):
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0
        d["valu_wave_instr_per_simd_cycle"] = d["SQ_INSTS_VALU"] / (cyc * 1024.0)
        d["wait_fraction_of_wave_cycles"] = d.get("SQ_WAIT_ANY", 0) / max(1.0, d.get("SQ_WAVE_CYCLES", 1))
    if "FETCH_SIZE" in d:
        d["hbm_bytes"] = (2.0 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)) * 1024.0
    print(k, json.dumps({c: (round(v / 1e6, 3) if v > 1e4 else round(v, 4)) for c, v in d.items()}))
json.dump(out, open(sys.argv[1] + '/pmc.json', 'w'), indent=1)
P
