set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03j_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r03j_pytest_gpu.log | tail -2; grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r03j_pytest_gpu.log | head -12
