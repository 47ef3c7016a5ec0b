cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_power.py tests/test_gpu_golden.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do timeout 600 python bench.py --steps 10 --warmup 3 --workload rx_power --cpu-seconds 0 --variants none 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); d=d.get('rx_power', d)
print('rx_power', round(d['value']/1e3,1), 'Gbins/s ms', round(d['roofline']['avg_launch_ms'],3))
"; done
