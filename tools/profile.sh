#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py, then the HBM
# PMC counters in their own passes (never combined with tracing; FETCH_SIZE and WRITE_SIZE do
# not fit one pass: /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/profile.sh <config> <tag> [extra bench args]
set -u
CFG=${1:-C3}
TAG=${2:-r01}
shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${CFG}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the resident-graph loop only: pre-reduced input, no C4 leg, no CPU baseline
BENCH="python $ROOT/bench.py --config $CFG --steps 2 --warmup 1 --cpu-seconds 0 --input dense --c4-leg off --c3-leg off $*"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
# SQ pass (8 slots): issue / wait breakdown of every kernel (quad-cycles; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)
# PMC_SMALL=1: only the HBM bytes and the L2 hit rate (big configs: every pass re-runs the whole bench)
PMC_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum")
if [ -z "${PMC_SMALL:-}" ]; then
    PMC_GROUPS+=("TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"
             "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY")
fi
for C in "${PMC_GROUPS[@]}"; do
    name=$(echo $C | tr ' ' '+')
    timeout 900 rocprofv3 --pmc $C -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1
done
# rocprofv3 writes rocpd .db files; reduce them to the small text summaries kept under profiles/
python $ROOT/tools/export_profile.py "$OUT" "$ROOT/gpurun_out/${TAG}_${CFG}" 2>&1 | tail -2
find "$OUT" -name "*.db" -size +8M -delete
cat "$ROOT/gpurun_out/${TAG}_${CFG}_kernel_stats.csv" | cut -c1-200
