#!/bin/bash
# tools/profile_round3.sh [TAG] -- the rocprofv3 evidence of round 3, written to gpurun_out/prof_TAG/ (run through gpurun);
# tools/collect_round3.py folds it into profiles/ (see profiles/README.md).  Counter passes are separate runs without tracing.
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
# 1. the default bench line (what the driver runs), every leg checked against the reference at size
python $REPO/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
# 2. kernel trace + stats of the headline command (--variants none: only headline launches of k_fm_decimate), rx_power and channeliser included
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py --variants none --no-parity > $OUT/trace_bench.json 2> $OUT/trace.log
# 3. kernel trace of the variants
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_variants -- python $REPO/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --workload rx_fm --no-parity > $OUT/trace_variants.json 2> $OUT/trace_variants.log
# 4. every rx_fm chain alone (tools/chain_once.py: 2 pipelined runs of 4 GiB): issue counters, fetched bytes, written bytes
for ds in 118 6 5 -7 -39; do
  i=0
  for set in "$SQ" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/chain_${ds}_p$i -- python $REPO/tools/chain_once.py 8192 $ds 2 > $OUT/chain_${ds}_p$i.log 2>&1 || echo "chain $ds pass $i failed"
  done
done
# 5. rx_power (configs[2] launch) and the channeliser
for wl in rx_power chan; do
  i=0
  for set in "$SQ" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/${wl}_p$i -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload $wl --variants none --no-parity > $OUT/${wl}_p$i.log 2>&1 || echo "$wl pass $i failed"
  done
done
# 6. the VALU issue ceiling per opcode
timeout 200 $REPO/tools/valu_issue $OUT/valu_issue.json > $OUT/valu_issue.txt 2>&1
# 7. the HBM ceiling for read streams with writes mixed in, on this box
timeout 120 $REPO/tools/rwmix $OUT/rwmix.json > $OUT/rwmix.txt 2>&1
cd $REPO
cut -c1-400 $OUT/bench_n1.json
ls $OUT | head -60
