#!/bin/bash
# round 5, call c: GPU suite on the side-stream snapshots + the device key index of the AMPC shard; first-run probe; AMPC throughput
# at 10 M keys; rocprofv3 kernel trace + HBM PMC passes of C4 on this tree (profiles/current_C4_pmc.json is from round 3)
set -u
O=gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
timeout 300 python tools/first_run_probe.py C3 > $O/first_run_probe_C3.json 2> $O/first_run_probe.err; echo "probe rc=$?"; cat $O/first_run_probe_C3.json
timeout 300 python tools/ampc_bench.py 10000000 1000000 > $O/ampc_bench_10M.json 2> $O/ampc_bench.err; echo "ampc rc=$?"; cat $O/ampc_bench_10M.json; tail -3 $O/ampc_bench.err
export HB_SYNTH_CACHE=/dev/shm/hb_synth_cache
PMC_SMALL=1 timeout 1500 bash tools/profile.sh C4 r05c > $O/profile_C4.log 2>&1; echo "profile rc=$?"; tail -30 $O/profile_C4.log | cut -c1-220
rm -rf /dev/shm/hb_synth_cache
ls gpurun_out | head -30
