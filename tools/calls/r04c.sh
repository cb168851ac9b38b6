#!/bin/bash
# round 4, call c: planner index checks under the bounds build, guard trace again, runtime/H2D comparison, insert kernel time, store writer on the box
set -u
O=gpurun_out/r04c; mkdir -p $O
export HB_LIB_PATH=$PWD/stract_amd/lib/libhyperball_bounds.so
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/bounds_smoke.log 2>&1; echo "bounds smoke rc=$?"; tail -2 $O/bounds_smoke.log | cut -c1-300
export HB_LIB_PATH=$PWD/stract_amd/lib/libhyperball_guard.so
HB_GUARD_TRACE=1 timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/guard_trace_smoke.log 2>&1; echo "guard trace smoke rc=$?"; grep "launch" $O/guard_trace_smoke.log | tail -1 | cut -c1-200
unset HB_LIB_PATH
timeout 400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "shipped pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
timeout 300 python tools/ingest_bench.py C3 --out $O/ingest_C3_own_runtime.json > /dev/null 2> $O/ingest_C3_own_runtime.err; echo "ingest C3 (library's runtime) rc=$?"
timeout 300 python tools/ingest_bench.py C3 --torch-first --out $O/ingest_C3_torch_runtime.json > /dev/null 2> $O/ingest_C3_torch_runtime.err; echo "ingest C3 (torch's runtime) rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_ingest -- python $OLDPWD/tools/ingest_bench.py C3 > /dev/null 2> $OLDPWD/$O/prof_ingest.err; echo "rocprof rc=$?"
cd $OLDPWD
find $O/prof_ingest -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-200' > $O/ingest_kernel_stats_head.txt
mkdir -p /dev/shm/sb && HB_TRACE_STORE=1 timeout 300 python tools/store_bench.py 20000000 --dir /dev/shm/sb --check 100 > $O/store_bench_shm.json 2> $O/store_bench_shm.err; echo "store shm rc=$?"; rm -rf /dev/shm/sb
mkdir -p /tmp/sb && HB_TRACE_STORE=1 timeout 300 python tools/store_bench.py 20000000 --dir /tmp/sb > $O/store_bench_disk.json 2> $O/store_bench_disk.err; echo "store disk rc=$?"; rm -rf /tmp/sb
