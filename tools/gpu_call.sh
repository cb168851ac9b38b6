set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03i_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r03i_pytest_gpu.log | tail -2; grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r03i_pytest_gpu.log | head -12
python __graft_entry__.py smoke > gpurun_out/r03i_smoke.txt 2>&1; tail -1 gpurun_out/r03i_smoke.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r03i_bench_default.json 2> gpurun_out/r03i_bench_default.err; tail -c 300 gpurun_out/r03i_bench_default.err; cut -c1-300 gpurun_out/r03i_bench_default.json
