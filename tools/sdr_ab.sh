cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_sdr.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 600 python bench.py --steps 3 --warmup 1 --workload sdr --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); d=d.get('sdr_convert', d)
for k,v in d['legs'].items(): print(k, round(v['GB/s']), round(v['frac_of_hbm_peak'],3))
"
