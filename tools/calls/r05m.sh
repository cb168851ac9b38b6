#!/bin/bash
# round 5, call m: the GPU suite + smoke on the tree of commit "profiles r05k/r05l" (known-good marker before further kernel work)
set -u
O=gpurun_out/r05m; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 200 python tools/diff_fuzz.py --mode mixed --seconds 60 --seed 909 > $O/diff_fuzz_mixed_gpu_seed909_new_switches.json 2> $O/diff_fuzz.err; echo "fuzz rc=$?"; cat $O/diff_fuzz_mixed_gpu_seed909_new_switches.json | cut -c1-300; tail -2 $O/diff_fuzz.err | cut -c1-300
