# tools/gpu_some.sh TAG pytest-args... -- selected GPU tests into gpurun_out/TAG/pytest.log (no test path among the arguments: all of tests/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-some}
shift
mkdir -p $O
cd $R
WHERE=tests
for a in "$@"; do case "$a" in tests/*) WHERE="";; esac; done
timeout 1500 python -u -m pytest $WHERE -m gpu -q -p no:cacheprovider --timeout 900 "$@" > $O/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $O/pytest.log)"
grep -E "^(FAILED|ERROR)" $O/pytest.log | head -20
