# tools/prof_cmd.sh TAG CMD... -- rocprofv3 --kernel-trace --stats of one command; the per-kernel table (calls, mean us, total ms) under gpurun_out/TAG/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof}
shift
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- "$@" > $O/cmd.out 2> $O/cmd.err
python - "$O" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(sys.argv[1] + "/kernel_stats.txt", "w") as out:
    for r in rows[:40]:
        line = "%-90s calls %6s  mean %10.1f us  total %9.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6)
        print(line); out.write(line + "\n")
PY
