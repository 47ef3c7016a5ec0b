# tools/gpu_fmab.sh TAG [pytest -k expr] -- rx_fm GPU tests (optional filter), then the rx_fm bench legs without the CPU checkers (timing only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-fmab}
mkdir -p $O
cd $R
timeout 900 python -u -m pytest tests/test_gpu_fm.py tests/test_gpu_golden.py -m gpu -q -x -p no:cacheprovider ${2:+-k "$2"} > $O/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/pytest.log)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
timeout 600 python bench.py --steps 20 --warmup 5 --workload rx_fm --cpu-seconds 0 --no-parity > $O/bench_fm.json 2> $O/bench_fm.err
echo bench rc=$?
python - "$O" <<'P'
import json, sys
d = json.load(open(sys.argv[1] + '/bench_fm.json'))
print('headline', round(d['value'] / 1e6, 3), 'TS/s dec frac', round(d['roofline']['frac'], 3), 'ms', round(d['roofline']['avg_launch_ms'], 3), 'box', {k: round(v) for k, v in d['roofline']['box_ceilings'].items() if k != 'how'})
for k, v in d['rx_fm_variants'].items():
    print(k[:40].ljust(40), round(v['value'] / 1e6, 3), round(v['frac_of_hbm_peak'], 3), v['stage_us_per_step'])
P
