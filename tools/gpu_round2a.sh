cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2a
timeout 1500 python -m pytest $R/tests -m gpu -q -p no:cacheprovider --timeout 600 > $R/gpurun_out/r2a/tests.log 2>&1
tail -15 $R/gpurun_out/r2a/tests.log
cd $R && timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
echo bench rc=$?
tail -c 1500 gpurun_out/r2a/bench.err
