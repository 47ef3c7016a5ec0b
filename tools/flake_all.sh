cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  timeout 900 python -u -m pytest tests -m gpu -q -x -p no:cacheprovider -s > gpurun_out/flake_$i.log 2>&1
  echo "run $i rc=$? $(tail -n 1 gpurun_out/flake_$i.log)"
done
