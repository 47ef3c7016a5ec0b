cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2e
mkdir -p $O
cd $R
for f in test_gpu_power; do
  timeout 900 python -u -m pytest tests/$f.py -m gpu -q -x -p no:cacheprovider --timeout 600 > $O/$f.log 2>&1
  echo "$f rc=$? $(tail -1 $O/$f.log)"
done
timeout 300 python tools/pw_probe.py > $O/pw_probe.log 2>&1; grep -v amdgpu.ids $O/pw_probe.log | tail -12
