#!/bin/bash
# round 5, call l: after the fuzz finding of call k (a pipelined pass must always be a sweep pass): the GPU suite with the regression
# test, then the differential fuzzer on the MI355X again - mixed (passes + record boundary + reference tail + ranks) and pass mode
set -u
O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
timeout 300 python tools/diff_fuzz.py --mode mixed --seconds 150 --seed 51 > $O/diff_fuzz_mixed_gpu_seed51.json 2> $O/diff_fuzz_a.err; echo "fuzz mixed 51 rc=$?"; cat $O/diff_fuzz_mixed_gpu_seed51.json | cut -c1-300; tail -2 $O/diff_fuzz_a.err | cut -c1-300
timeout 200 python tools/diff_fuzz.py --mode records --seconds 90 --seed 77 > $O/diff_fuzz_records_gpu_seed77.json 2> $O/diff_fuzz_b.err; echo "fuzz records 77 rc=$?"; cat $O/diff_fuzz_records_gpu_seed77.json | cut -c1-300; tail -2 $O/diff_fuzz_b.err | cut -c1-300
timeout 200 python tools/diff_fuzz.py --mode passes --seconds 60 --seed 5 > $O/diff_fuzz_passes_gpu_seed5.json 2> $O/diff_fuzz_c.err; echo "fuzz passes 5 rc=$?"; cat $O/diff_fuzz_passes_gpu_seed5.json | cut -c1-300
