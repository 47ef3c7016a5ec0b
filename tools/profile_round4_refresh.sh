#!/bin/bash
# tools/profile_round4_refresh.sh -- the parts of tools/profile_round4.sh that the second half of round 4 changed (the bench line, the kernel trace of the
# headline command, the channeliser's counter passes and its A/B, the drop-in latency table), into the same gpurun_out/prof_r04/;
# tools/collect_round4.py then folds everything into profiles/ again
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
LDS="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
python $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py --steps 20 --warmup 5 --variants none --no-parity > $OUT/trace_bench.json 2> $OUT/trace.log
i=0
for set in "$SQ" "FETCH_SIZE" "WRITE_SIZE" "$LDS"; do
  i=$((i+1))
  rm -rf $OUT/chan_p$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/chan_p$i -- python $REPO/tools/chan_once.py > $OUT/chan_p$i.log 2>&1 || echo "chan pass $i failed"
done
python $REPO/tools/ab_chan.py RXGPU_FFT_TW=global RXGPU_CH_GPW=1 RXGPU_CH_WPG=8,RXGPU_CH_GPW=8 > $OUT/ab_chan_tw.txt 2>&1
python $REPO/tools/dropin_latency.py > $OUT/dropin_latency.txt 2>&1
cd $REPO
python tools/bench_show.py $OUT/bench_n1.json 2>&1 | tail -30
tail -5 $OUT/ab_chan_tw.txt; tail -4 $OUT/dropin_latency.txt
