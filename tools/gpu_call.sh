set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03p_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r03p_pytest_gpu.log | tail -2; grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r03p_pytest_gpu.log | head -12
python __graft_entry__.py smoke > gpurun_out/r03p_smoke.txt 2>&1; tail -1 gpurun_out/r03p_smoke.txt
( time python bench.py ) > gpurun_out/r03p_bench_default.json 2> gpurun_out/r03p_bench_default.err; tail -4 gpurun_out/r03p_bench_default.err; cut -c1-300 gpurun_out/r03p_bench_default.json
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r03p -o trace -- python /root/repo/bench.py --config C3 --steps 2 --warmup 1 --cpu-seconds 0 --input dense --c4-leg off > /root/repo/gpurun_out/r03p_trace.log 2>&1
python tools/export_profile.py gpurun_out/prof_r03p gpurun_out/r03p_C3 2>&1 | tail -2; find gpurun_out/prof_r03p -name "*.db" -size +8M -delete; head -12 gpurun_out/r03p_C3_kernel_stats.csv | cut -c1-150
