# tools/trace_quick.sh TAG CMD... -- rocprofv3 kernel trace + stats of CMD, prints the k_* rows of the stats table
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trq_$1; shift
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- "$@" > $O/log 2>&1
cd $R
python - $O <<'P'
import sys, glob, csv
f = glob.glob(sys.argv[1] + '/t/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].split('(')[0].replace('void ', '')
    if n.startswith('k_'):
        print(n[:46].ljust(46), 'calls', r['Calls'].rjust(5), 'avg_us %9.1f' % (float(r['AverageNs']) / 1e3), 'min %8.1f' % (float(r['MinNs']) / 1e3), 'max %8.1f' % (float(r['MaxNs']) / 1e3))
P
