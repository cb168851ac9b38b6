#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 --kernel-trace --stats of an arbitrary command, reduced to the
# per-kernel CSV kept under profiles/ (tools/export_profile.py).
# usage: tools/trace.sh <tag> <command...>      -> gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=$1
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
(cd "$ROOT" && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- "$@" > "$OUT/trace.log" 2>&1)
python "$ROOT/tools/export_profile.py" "$OUT" "$ROOT/gpurun_out/${TAG}" 2>&1 | tail -1
find "$OUT" -name "*.db" -size +8M -delete
cut -c1-220 "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" | head -24
