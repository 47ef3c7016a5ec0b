"""tools/kernel_timeline.py TRACE_DIR [MARKER] -- the kernels of the LAST launch in a `rocprofv3 --kernel-trace --output-format csv -d TRACE_DIR` run, in
start order with duration and the idle gap in front of each: where a launch of many kernels spends its time (the small ones included).  A launch is what
lies between the last two kernels whose name contains MARKER (default k_pwm_reduce: the last kernel of a large-N rx_power launch)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else 'k_pwm_reduce'
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%-46s start %8.1f  dur %8.1f  gap %6.1f" % (r['Kernel_Name'][:46], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = max(prev_end, e)
print("total %.1f us" % ((prev_end - t0) / 1e3))
