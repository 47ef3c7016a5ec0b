# tools/gpu_bench.sh TAG [pytest -k expression] -- selected GPU tests, then the driver's bench command; everything under gpurun_out/TAG/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-bench}
mkdir -p $O
cd $R
if [ -n "$2" ]; then
  timeout 1500 python -u -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -k "$2" > $O/pytest.log 2>&1
  echo "pytest rc=$? $(tail -1 $O/pytest.log)"; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
fi
timeout 1200 python bench.py --steps 20 --warmup 5 --full-out $O/bench_full.json > $O/bench.json 2> $O/bench.err
echo "bench rc=$? $(grep -v "^BENCH_FULL" $O/bench.err | tail -3)"
echo "stdout: $(wc -l < $O/bench.json) line(s), $(wc -c < $O/bench.json) bytes"; cat $O/bench.json
python tools/bench_show.py $O/bench_full.json 2>&1 | tail -40
