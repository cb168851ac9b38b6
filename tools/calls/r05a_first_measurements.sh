#!/bin/bash
# round 5, call a (prepared at the end of round 4, not yet run): the measurements the ranked next steps of DESIGN.md §8 ask for FIRST,
# in one short call (~3 GPU minutes):
#   1. tools/bittest_bench.bin  - is a changed-bit test against an LDS-resident part of the bitmap cheaper than the L2 hit? (bitmap pass, step 3)
#   2. the two GPU tests added at the end of round 4 without a GPU run (logic checked on the interpreted sources only)
#   3. tools/diff_fuzz.py --mode mixed on the MI355X (passes + record boundary)
#   4. rocprofv3 kernel trace of the LT graph: per-kernel time of the sweep passes (seed / expand / levels / node rows) = the budget of step 1
set -u
O=gpurun_out/r05a; mkdir -p $O
# 0. (round 5) the two request-size questions behind VERDICT r4 #1: 32-B gathers, and line mates gathered by NEIGHBOURING quads
timeout 120 tools/gather_bench.bin L2_4MiB,MALL_256MiB,HBM_8GiB 1 > $O/gather_request_size.txt 2>&1; echo "gather rc=$?"; cat $O/gather_request_size.txt
timeout 60 tools/bittest_bench.bin > $O/bittest_bench.txt 2>&1; echo "bittest rc=$?"; cat $O/bittest_bench.txt
timeout 200 python -m pytest tests/test_gpu.py -m gpu -q -k "very_long_reader or test_load_webgraph_from_edge_store" > $O/new_tests.log 2>&1; echo "new tests rc=$?"; tail -2 $O/new_tests.log
timeout 60 python tools/diff_fuzz.py --mode mixed --seconds 40 --seed 9 > $O/diff_fuzz_mixed_gpu.json 2> $O/diff_fuzz_mixed_gpu.err; echo "fuzz rc=$?"; cat $O/diff_fuzz_mixed_gpu.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_LT -- python $GRAFT_REPO_ROOT/bench.py --config LT --c4-leg off --end-to-end off --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/bench_LT.json 2> $GRAFT_REPO_ROOT/$O/bench_LT.err; echo "LT trace rc=$?"
find $GRAFT_REPO_ROOT/$O/prof_LT -name "*kernel_stats.csv" | head -1 | xargs -r head -25
