# tools/timeline.sh KERNEL_SUBSTRING CMD... -- rocprofv3 kernel trace of CMD; prints the gaps between consecutive launches of the
# named kernel and everything that ran between two of them late in the run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift
O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -- "$@" > $O/out 2> $O/log
cd $R
python - "$K" <<'P'
import csv, glob, statistics as st, sys
rows = list(csv.DictReader(open(glob.glob('gpurun_out/tl/t/*/*_kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if sys.argv[1] in r['Kernel_Name']]
n = len(idx)
sel = idx[n // 2: n - 2]
gaps = [(int(rows[b]['Start_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3 for a, b in zip(sel, sel[1:])]
durs = [(int(rows[a]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 for a in sel]
print('%d launches; duration median %.1f us, gap median %.1f mean %.1f max %.1f' % (n, st.median(durs), st.median(gaps), st.mean(gaps), max(gaps)))
i0 = idx[n - 4]; t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:idx[n - 3] + 1]:
    print('  ', r['Kernel_Name'].split('(')[0].replace('void ', '')[:40].ljust(40), 'q', r['Queue_Id'], 'start %8.1f end %8.1f' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3))
P
