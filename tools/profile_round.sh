#!/bin/bash
# tools/profile_round.sh TAG [SECTION...] -- the rocprofv3 evidence of a round, written to gpurun_out/prof_TAG/ (run through gpurun);
# `python tools/collect_profiles.py TAG gpurun_out/prof_TAG` then folds it into profiles/ (profiles/README.md says what each file is).
# Counter passes are separate runs with --pmc only (never combined with a trace).  SECTIONs (default: all but `probes`):
#   bench     the driver's bench command (compact line + full record), every leg checked against the reference at size
#   trace     rocprofv3 --kernel-trace --stats of the headline command and of the rx_fm variants
#   chains    every rx_fm chain alone (tools/chain_once.py): issue counters, FETCH_SIZE, WRITE_SIZE
#   power     rx_power configs[2] launches: issue, FETCH_SIZE, WRITE_SIZE, LDS counters (--settle-ms 0: the per-launch figures divide by steps + warmup)
#   chan      the channeliser's bench shape (tools/chan_once.py): the same four passes; its other modes (-A std, audio stages, NCO): the issue
#             counters behind their bound; the per-channel audio stages' kernel trace
#   legs      the other rx_power geometries of the bench line (tools/pw_big_once.py): issue, FETCH_SIZE, WRITE_SIZE
#   dropin    per-block latency of rxgpu_callback + rxgpu_full_demod (tools/dropin_latency.py)
#   probes    box ceilings that do not change with the code: VALU issue per opcode (tools/valu_issue.hip), mixed read/write traffic (tools/rwmix.hip)
set -u
TAG=${1:-r05}
shift || true
SECTIONS=${*:-bench trace chains power chan legs dropin}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
LDS="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
has() { case " $SECTIONS " in *" $1 "*) return 0;; esac; return 1; }
pmc() { # DIR-PREFIX "counter sets..." -- CMD...   one rocprofv3 --pmc run per set
  local prefix=$1 sets=$2; shift 2
  local i=0
  IFS='|' read -ra SETS <<< "$sets"
  for set in "${SETS[@]}"; do
    i=$((i+1))
    rm -rf $OUT/${prefix}_p$i
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/${prefix}_p$i -- "$@" > $OUT/${prefix}_p$i.log 2>&1 || echo "$prefix pass $i failed"
  done
}
if has bench; then
  python $REPO/bench.py --steps 20 --warmup 5 --full-out $OUT/bench_full.json > $OUT/bench_n1.json 2> $OUT/bench_n1.err
  echo "bench rc=$? stdout $(wc -l < $OUT/bench_n1.json) line(s) $(wc -c < $OUT/bench_n1.json) bytes"
fi
if has trace; then
  rm -rf $OUT/trace $OUT/trace_variants
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py --steps 20 --warmup 5 --variants none --no-parity --full-out $OUT/trace_bench_full.json > $OUT/trace_bench.json 2> $OUT/trace.log
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_variants -- python $REPO/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --workload rx_fm --no-parity --full-out $OUT/trace_variants_full.json > $OUT/trace_variants.json 2> $OUT/trace_variants.log
fi
if has chains; then
  for ds in 118 6 5 -7 -39; do
    pmc chain_${ds} "$SQ|FETCH_SIZE|WRITE_SIZE" python $REPO/tools/chain_once.py 8192 $ds 2
  done
fi
if has power; then
  pmc rx_power "$SQ|FETCH_SIZE|WRITE_SIZE|$LDS" python $REPO/bench.py --steps 2 --warmup 1 --settle-ms 0 --cpu-seconds 0 --workload rx_power --variants none --no-parity --full-out $OUT/pmc_power_full.json
fi
if has chan; then
  pmc chan "$SQ|FETCH_SIZE|WRITE_SIZE|$LDS" python $REPO/tools/chan_once.py
  for mode in std audio nco; do
    pmc chanmode_$mode "$SQ" python $REPO/tools/chan_once.py $mode
  done
  rm -rf $OUT/trace_chan_audio
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_chan_audio -- python $REPO/tools/chan_audio_time.py > $OUT/chan_audio.txt 2> $OUT/chan_audio.err
fi
if has legs; then
  leg() { PW_BOXCAR=$4 PW_FIR=$5 pmc leg_$1 "$SQ|FETCH_SIZE|WRITE_SIZE" python $REPO/tools/pw_big_once.py $2 $3 2; }
  leg n14_fir9 100M:100.1M:10 4096 0 9
  leg n14_box28 100M:100.1M:10 4096 1 0
  leg n15_box14 100M:100.2M:10 2048 1 0
  leg n18 100M:102.8M:20 256 1 0
fi
if has dropin; then
  python $REPO/tools/dropin_latency.py > $OUT/dropin_latency.txt 2>&1
fi
if has probes; then
  (cd $REPO/tools && for t in valu_issue rwmix; do [ -x $t ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $t $t.hip; done)
  timeout 200 $REPO/tools/valu_issue $OUT/valu_issue.json > $OUT/valu_issue.txt 2>&1
  timeout 120 $REPO/tools/rwmix $OUT/rwmix.json > $OUT/rwmix.txt 2>&1
fi
cd $REPO
[ -f $OUT/bench_n1.json ] && cut -c1-600 $OUT/bench_n1.json
ls $OUT | wc -l
