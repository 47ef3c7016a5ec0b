#!/bin/bash
# tools/profile_round4.sh [TAG] -- the rocprofv3 evidence of round 4, written to gpurun_out/prof_TAG/ (run through gpurun);
# tools/collect_round4.py folds it into profiles/ (see profiles/README.md).  Counter passes are separate runs without tracing.
set -u
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
LDS="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
# 1. the driver's bench command, every leg checked against the reference at size
python $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
# 2. kernel trace + stats of the headline command (--variants none: only headline launches of k_fm_decimate), rx_power and channeliser included
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py --steps 20 --warmup 5 --variants none --no-parity > $OUT/trace_bench.json 2> $OUT/trace.log
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
# 5. rx_power (configs[2] launch) and the channeliser, LDS counters included; the configs[2] launch also with the twiddles through the vector cache (A/B)
for wl in rx_power chan; do
  i=0
  for set in "$SQ" "FETCH_SIZE" "WRITE_SIZE" "$LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/${wl}_p$i -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload $wl --variants none --no-parity > $OUT/${wl}_p$i.log 2>&1 || echo "$wl pass $i failed"
  done
done
i=0
for set in "$SQ" "$LDS"; do
  i=$((i+1))
  RXGPU_FFT_TW=global timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/rx_power_twglobal_p$i -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload rx_power --variants none --no-parity > $OUT/rx_power_twglobal_p$i.log 2>&1 || echo "rx_power tw=global pass $i failed"
done
# 6. the other rx_power geometries of the bench line, one launch shape each (tools/pw_big_once.py): issue counters, fetched and written bytes
leg() { # name range passes boxcar fir
  i=0
  for set in "$SQ" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    PW_BOXCAR=$4 PW_FIR=$5 timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/leg_$1_p$i -- python $REPO/tools/pw_big_once.py $2 $3 2 > $OUT/leg_$1_p$i.log 2>&1 || echo "leg $1 pass $i failed"
  done
}
leg n14_fir9 100M:100.1M:10 4096 0 9
leg n14_box28 100M:100.1M:10 4096 1 0
leg n15_box14 100M:100.2M:10 2048 1 0
leg n18 100M:102.8M:20 256 1 0
# 7. A/B inside one process: twiddles from LDS or through the vector cache (rx_power configs[2], the channeliser)
python $REPO/tools/ab_power.py RXGPU_FFT_TW=global > $OUT/ab_power_tw.txt 2>&1
python $REPO/tools/ab_chan.py RXGPU_FFT_TW=global > $OUT/ab_chan_tw.txt 2>&1
python $REPO/tools/ab_power_big.py > $OUT/ab_power_big.txt 2>&1
# 8. the VALU issue ceiling per opcode, the cndmask probe, the mixed-traffic ceilings of this box
timeout 200 $REPO/tools/valu_issue $OUT/valu_issue.json > $OUT/valu_issue.txt 2>&1
timeout 100 $REPO/tools/cndmask_probe > $OUT/cndmask_probe.txt 2>&1
timeout 120 $REPO/tools/rwmix $OUT/rwmix.json > $OUT/rwmix.txt 2>&1
cd $REPO
cut -c1-400 $OUT/bench_n1.json
ls $OUT | wc -l
