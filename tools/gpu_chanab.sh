# tools/gpu_chanab.sh TAG -- channeliser tests, then the channeliser bench leg with RXGPU_CH_GPW = default, 1, 2, 8 (one process each)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-chanab}
mkdir -p $O
cd $R
timeout 900 python -u -m pytest tests/test_gpu_chan.py -m gpu -q -p no:cacheprovider --timeout 600 > $O/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/pytest.log)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
for g in "" 1 2 8; do
  RXGPU_CH_GPW=$g timeout 600 python bench.py --workload chan --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_gpw${g:-def}.json 2> $O/bench_gpw${g:-def}.err
  python - $O/bench_gpw${g:-def}.json "$g" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d.get('channeliser',d)
print('GPW',sys.argv[2] or 'default', round(c['value']/1e3,1),'GS/s', c['ms_per_step'], c['roofline'].get('avg_launch_ms'), c['parity'].get('parity_ok'), c.get('nco_mode',{}).get('value'))
P
done
