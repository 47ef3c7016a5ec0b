# tools/gpu_quick.sh TAG -- the rx_fm / rx_power GPU tests, the small-decimation probe and a short rx_fm bench (a 30-second gpurun call)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-quick}
mkdir -p $O
cd $R
for f in test_gpu_fm test_gpu_power test_gpu_chan test_gpu_dropin; do
  timeout 900 python -u -m pytest tests/$f.py -m gpu -q -x -p no:cacheprovider --timeout 600 > $O/$f.log 2>&1
  echo "$f rc=$? $(tail -1 $O/$f.log)"
done
timeout 300 python tools/ds6_probe.py 8192 > $O/ds6_probe.log 2>&1; grep tiled $O/ds6_probe.log
timeout 600 python bench.py --steps 20 --warmup 5 --workload rx_fm --cpu-seconds 0 --no-parity > $O/bench_fm.json 2> $O/bench_fm.err
echo bench rc=$?
python - "$O" <<'P'
import json, sys
d = json.load(open(sys.argv[1] + '/bench_fm.json'))
print('headline', round(d['value'] / 1e6, 3), 'TS/s dec frac', round(d['roofline']['frac'], 3), 'ms', round(d['roofline']['avg_launch_ms'], 3))
for k, v in d['rx_fm_variants'].items():
    print(k[:40], round(v['value'] / 1e6, 3), v['stage_us_per_step'])
P
