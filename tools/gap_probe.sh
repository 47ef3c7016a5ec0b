# tools/gap_probe.sh -- gaps between consecutive headline decimator launches and the length of the audio chain between them (rocprofv3 trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gap; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python $R/bench.py --steps 40 --warmup 5 --cpu-seconds 0 --workload rx_fm --variants none --no-parity > $O/bench.json 2> $O/log
cd $R
python - <<'P'
import csv, glob, statistics as st, json
rows = list(csv.DictReader(open(glob.glob('gpurun_out/gap/t/*/*_kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_fm_decimate<' in r['Kernel_Name']]
gaps = [(int(rows[b]['Start_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3 for a, b in zip(idx[8:40], idx[9:41])]
durs = [(int(rows[a]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 for a in idx[8:40]]
print('decimator median %.1f us, gap median %.1f mean %.1f max %.1f' % (st.median(durs), st.median(gaps), st.mean(gaps), max(gaps)))
i0 = idx[20]; t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:idx[21] + 1]:
    print('  ', r['Kernel_Name'].split('(')[0].replace('void ', '')[:34].ljust(34), 'start %8.1f end %8.1f' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3))
d = json.load(open('gpurun_out/gap/bench.json'))
print('bench under the profiler: value %.3f TS/s, step %.1f us, kernel %.1f us' % (d['value'] / 1e6, d['ms_per_step'] * 1e3, d['roofline']['avg_launch_ms'] * 1e3))
P
