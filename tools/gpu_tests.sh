# tools/gpu_tests.sh TAG [pytest args] -- the whole -m gpu suite in one pytest process, the way the driver runs it, + smoke(); log under gpurun_out/TAG/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-tests}
shift
mkdir -p $O
cd $R
timeout 1500 python -u -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 "$@" > $O/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/pytest.log)"
grep -E "^(FAILED|ERROR)" $O/pytest.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log)"
