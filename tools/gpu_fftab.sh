# tools/gpu_fftab.sh TAG -- rx_power / channeliser GPU tests, then their bench legs without the CPU checkers (timing only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-fftab}
mkdir -p $O
cd $R
timeout 900 python -u -m pytest tests/test_gpu_power.py tests/test_gpu_chan.py tests/test_gpu_golden.py -m gpu -q -p no:cacheprovider -k "not multi_rank" > $O/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/pytest.log)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
timeout 600 python bench.py --workload rx_power --steps 10 --warmup 3 --cpu-seconds 0 --no-parity > $O/pw.json 2> $O/pw.err; echo "pw rc=$?"
timeout 600 python bench.py --workload chan --steps 20 --warmup 5 --cpu-seconds 0 --no-parity > $O/ch.json 2> $O/ch.err; echo "ch rc=$?"
python - "$O" <<'P'
import json, sys
d = json.load(open(sys.argv[1] + '/pw.json'))
print('rx_power', round(d['value'] / 1e3, 1), 'Gbins/s', 'launch ms', round(d['roofline']['avg_launch_ms'], 3), 'dropin', d.get('dropin_scan_us'))
for k, v in d['other_geometries'].items():
    print('  %-60s %.1f Gbins/s  %.3f ms' % (k[:60], v['Mbins/s'] / 1e3, v['ms']))
c = json.load(open(sys.argv[1] + '/ch.json'))['channeliser']
print('chan', round(c['value'] / 1e3, 1), 'GS/s', 'launch ms', round(c['roofline']['avg_launch_ms'], 3), 'ms/step', round(c['ms_per_step'], 3))
P
