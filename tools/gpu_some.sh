# tools/gpu_some.sh TAG "pytest args" -- selected GPU tests into gpurun_out/TAG/ (one log), tail printed
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-some}
mkdir -p $O
cd $R
shift
timeout 1500 python -u -m pytest "$@" -m gpu -q -p no:cacheprovider --timeout 900 > $O/pytest.log 2>&1
echo "rc=$? $(tail -1 $O/pytest.log)"
grep -E "^(FAILED|ERROR)" $O/pytest.log | head -40
