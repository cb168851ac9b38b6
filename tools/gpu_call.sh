set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu.py -m gpu -x -q -k "per_pass_state or logical_ranks or full_run or unfused" > gpurun_out/r03l_pytest_subset.log 2>&1; grep -E "passed|failed" gpurun_out/r03l_pytest_subset.log | tail -2; grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r03l_pytest_subset.log | head -12
python tools/sweep.py C3 "0:0:" "0:0:0,8192" "0:0:" > gpurun_out/r03l_sweep_pipeline_C3.txt 2>&1; cut -c1-230 gpurun_out/r03l_sweep_pipeline_C3.txt
