# tools/gpu_full.sh TAG -- the whole GPU test-suite (one pytest process per file: a crash in one cannot hide the others) and the
# default bench line, written to gpurun_out/TAG/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-full}
mkdir -p $O
cd $R
for f in tests/test_*.py; do
  b=$(basename $f .py)
  timeout 1200 python -u -m pytest $f -m gpu -q -p no:cacheprovider --timeout 900 > $O/$b.log 2>&1
  echo "$b rc=$? $(tail -1 $O/$b.log)"
done
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log)"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench rc=$? $(cut -c1-300 $O/bench.json)"
