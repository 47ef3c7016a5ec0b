cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g
mkdir -p $O
cd $R
timeout 900 python -u -m pytest tests/test_gpu_fm.py tests/test_gpu_power.py tests/test_gpu_dropin.py tests/test_gpu_golden.py tests/test_dropin_e2e.py -m gpu -q -x -p no:cacheprovider --timeout 600 -k "fifth or power or scan or dropin or golden" > $O/tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/tests.log)"
timeout 300 python tools/pw_probe.py > $O/pw_probe.log 2>&1; grep -v amdgpu.ids $O/pw_probe.log | tail -8
timeout 600 python bench.py --steps 10 --warmup 3 --workload rx_fm --cpu-seconds 0 --no-parity > $O/bench_fm.json 2> $O/bench_fm.err
echo bench rc=$?
python - <<'P'
import json
d=json.load(open('gpurun_out/r2g/bench_fm.json'))
print('headline', round(d['value']/1e6,3), 'TS/s dec frac', round(d['roofline']['frac'],3), 'ms', round(d['roofline']['avg_launch_ms'],3))
for k,v in d['rx_fm_variants'].items(): print(k[:40], round(v['value']/1e6,3), v['stage_us_per_step'])
P
