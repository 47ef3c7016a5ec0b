# tools/gpu_blocks.sh TAG BLOCKS... -- the rx_fm bench legs (timing only) at several run sizes: does a smaller run (its pcm inside the 256 MB Infinity Cache) go faster per sample?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-blocks}; shift
mkdir -p $O
cd $R
for b in "$@"; do
  timeout 300 python bench.py --steps 40 --warmup 5 --workload rx_fm --cpu-seconds 0 --no-parity --blocks $b > $O/b$b.json 2> $O/b$b.err
  python - "$O/b$b.json" $b <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print('blocks', sys.argv[2], 'headline', round(d['value'] / 1e6, 3), '|', ' '.join('%.3f' % (v['value'] / 1e6) for v in d['rx_fm_variants'].values()))
P
done
