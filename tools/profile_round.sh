#!/bin/bash
# tools/profile_round.sh TAG -- the rocprofv3 evidence of one round, written to gpurun_out/prof_TAG/ (run through gpurun),
# then copy what is to be judged into profiles/ (tools/collect_profiles.py; see profiles/README.md).
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the default bench line (what the driver runs)
python $REPO/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
# 2. kernel trace + stats of the headline command.  --variants none: the ds=6 / ds=5 chains launch the SAME decimator kernel as the
#    headline, so the per-kernel averages of this run describe the headline launches only; everything else is the default command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py --variants none > $OUT/trace_bench.json 2> $OUT/trace.log
# 3. HBM traffic of the rx_fm kernels: separate --pmc passes, no tracing; 4 GiB launches
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload rx_fm --variants none --no-parity --blocks 8192 > $OUT/pmc_$c.log 2>&1
done
# 4. what binds the rx_power transform: VALU issue and LDS conflicts
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu_power -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload rx_power --variants none > $OUT/pmc_valu_power.log 2>&1
# 5. instruction counts of every rx_fm kernel incl. the small-decimation and -F chains (one launch shape each)
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu_fm -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload rx_fm --no-parity --blocks 8192 > $OUT/pmc_valu_fm.log 2>&1
# 6. kernel trace of the variants (which kernels the ds=6 / -F chains launch, and for how long)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_variants -- python $REPO/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --workload rx_fm --no-parity > $OUT/trace_variants.json 2> $OUT/trace_variants.log
cd $REPO
find $OUT -name "*.csv" | head -40
cut -c1-600 $OUT/bench_n1.json
