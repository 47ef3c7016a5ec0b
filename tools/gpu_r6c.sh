cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c
mkdir -p $O
cd $R
timeout 1500 python -u -m pytest tests/test_gpu_power.py tests/test_gpu_chan.py -m gpu -q -p no:cacheprovider --timeout 900 -x > $O/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/pytest.log)"
grep -E "^(FAILED|ERROR)" $O/pytest.log | head
python bench.py --workload chan --no-parity --cpu-seconds 0 --variants none --full-out $O/chan_full.json > $O/chan.json 2> $O/chan.err; python - $O/chan_full.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d.get("channeliser", d)
print("chan GS/s", c["value"]/1e3, "launch ms", c["roofline"]["avg_launch_ms"], "valu frac", c["roofline"]["valu"]["frac"])
PY
python bench.py --workload rx_power --no-parity --cpu-seconds 0 --variants none --full-out $O/pw_full.json > $O/pw.json 2> $O/pw.err; python - $O/pw_full.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d.get("rx_power", d)
print("rx_power Gbins/s", c["value"]/1e3, "launch ms", c["roofline"]["avg_launch_ms"], "valu frac", c["roofline"]["frac"])
PY
LDS="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
rm -rf $O/pmc_lds; timeout 300 rocprofv3 --pmc $LDS --output-format csv -d $O/pmc_lds -- python tools/chan_once.py > $O/pmc_lds.log 2>&1
python - $O <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1]+"/pmc_lds/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if k.startswith("k_ch"): acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,d in acc.items(): print(k, "conflict fraction", round(d["SQ_LDS_BANK_CONFLICT"]/max(d["SQ_LDS_IDX_ACTIVE"],1),3), {c:int(v) for c,v in d.items()})
PY
