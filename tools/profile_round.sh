#!/bin/bash
# tools/profile_round.sh TAG -- the rocprofv3 evidence of one round, written to gpurun_out/prof_TAG/ (run through gpurun)
# then copy what is to be judged into profiles/ (see profiles/README.md).
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
# the SAME command as the bench line above (defaults: 100 steps, 10 warm-up, CPU baselines included)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py > $OUT/trace_bench.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload rx_fm --blocks 8192 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload rx_fm --blocks 8192 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/valu -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload rx_power > $OUT/valu.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/valu_fm -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload rx_fm --blocks 8192 > $OUT/valu_fm.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -30
cat $OUT/bench_n1.json | cut -c1-1500
