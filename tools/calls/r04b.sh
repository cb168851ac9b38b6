#!/bin/bash
# round 4, call b: which kernel over-reads under the guard-page build (16-byte guard)? + first run of the rebuilt ingest
set -u
O=gpurun_out/r04b; mkdir -p $O
export HB_LIB_PATH=$PWD/stract_amd/lib/libhyperball_guard.so
HB_GUARD_TRACE=1 timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/guard_trace_smoke.log 2>&1; echo "guard trace smoke rc=$?"; grep -c "done" $O/guard_trace_smoke.log; tail -4 $O/guard_trace_smoke.log | cut -c1-300
HB_GUARD_TRACE=1 timeout 300 python -m pytest tests/test_gpu.py -m gpu -x -q -k "gpu_ingest_equals_host_ingest" > $O/guard_trace_ingest.log 2>&1; echo "guard trace ingest rc=$?"; grep "launch" $O/guard_trace_ingest.log | tail -2 | cut -c1-300; tail -3 $O/guard_trace_ingest.log | cut -c1-300
export HB_LIB_PATH=$PWD/stract_amd/lib/libhyperball_bounds.so
timeout 600 python -m pytest tests -m gpu -q > $O/bounds_pytest_gpu.log 2>&1; echo "bounds pytest rc=$?"; tail -4 $O/bounds_pytest_gpu.log | cut -c1-300
export HB_LIB_PATH=$PWD/stract_amd/lib/libhyperball_redzone.so
timeout 600 python -m pytest tests -m gpu -q > $O/redzone_pytest_gpu.log 2>&1; echo "redzone pytest rc=$?"; tail -4 $O/redzone_pytest_gpu.log | cut -c1-300
unset HB_LIB_PATH
timeout 400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "shipped pytest rc=$?"; tail -4 $O/pytest_gpu.log | cut -c1-300
HB_TRACE_INGEST=1 timeout 300 python tools/ingest_bench.py C3 --verify --out $O/ingest_C3.json > /dev/null 2> $O/ingest_C3.err; echo "ingest C3 rc=$?"; grep "hb ingest" $O/ingest_C3.err | tail -12
