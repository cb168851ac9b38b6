set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03f_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r03f_pytest_gpu.log | tail -2; grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r03f_pytest_gpu.log | head -8
python tools/sweep.py C3 "0:0:" "0:0:0,8192" "0:0:0,0,101,0,0,0,1000000" "0:0:0,8192,101,0,0,0,1000000" > gpurun_out/r03f_sweep_C3.txt 2>&1; cut -c1-600 gpurun_out/r03f_sweep_C3.txt
python tools/sweep.py LT "0:0:" "0:0:0,8192" > gpurun_out/r03f_sweep_LT.txt 2>&1; cut -c1-400 gpurun_out/r03f_sweep_LT.txt
