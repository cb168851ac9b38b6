#!/bin/bash
# round 4, call a: is the debug allocator trustworthy? + first regression of the changed build
set -u
O=gpurun_out/r04a; mkdir -p $O
{ nproc; free -g | head -2; df -h . /tmp /dev/shm 2>/dev/null; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null; } > $O/host.txt 2>&1
timeout 120 tools/guard_selftest_copies0.bin > $O/guard_selftest_copies0.txt 2>&1; echo "selftest copies0 rc=$?"
timeout 120 tools/guard_selftest_copies1.bin > $O/guard_selftest_copies1.txt 2>&1; echo "selftest copies1 rc=$?"
timeout 120 tools/h2d_probe.bin > $O/h2d_probe.txt 2>&1; echo "h2d rc=$?"
timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
export HB_LIB_PATH=$PWD/stract_amd/lib/libhyperball_guard.so
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/guard_smoke.log 2>&1; echo "guard smoke rc=$?"; tail -1 $O/guard_smoke.log
timeout 400 python -m pytest tests/test_gpu.py -m gpu -x -q -k "c1_config_bit_exact or gpu_ingest_equals_host_ingest or device_plan_equals_host_plan" > $O/guard_tests.log 2>&1; echo "guard tests rc=$?"; tail -2 $O/guard_tests.log
export HB_LIB_PATH=$PWD/stract_amd/lib/libhyperball_redzone.so
timeout 600 python -m pytest tests -m gpu -x -q > $O/redzone_pytest_gpu.log 2>&1; echo "redzone pytest rc=$?"; tail -2 $O/redzone_pytest_gpu.log
timeout 300 python tools/record_stress.py C3,LT --rounds 2 --tag redzone --out $O/stress_redzone.json > /dev/null 2> $O/stress_redzone.err; echo "stress redzone rc=$?"
unset HB_LIB_PATH
timeout 300 python tools/record_stress.py C3 --rounds 5 --tag shipped --out $O/stress_shipped.json > /dev/null 2> $O/stress_shipped.err; echo "stress shipped rc=$?"
