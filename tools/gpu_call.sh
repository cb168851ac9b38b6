set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03c_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r03c_pytest_gpu.log | tail -2
python tools/sweep.py C3 "0:0:" "0:0:0,1024" "0:0:0,2048" > gpurun_out/r03c_sweep_C3.txt 2>&1; cut -c1-700 gpurun_out/r03c_sweep_C3.txt
python tools/sweep.py LT "0:0:" "0:0:0,2048" > gpurun_out/r03c_sweep_LT.txt 2>&1; cut -c1-300 gpurun_out/r03c_sweep_LT.txt
python bench.py --config C5 --verify --input dense --steps 2 --warmup 1 --c4-leg off --pass-log gpurun_out/r03c_passes_C5.json > gpurun_out/r03c_bench_C5_verify.json 2> gpurun_out/r03c_bench_C5.err; tail -c 600 gpurun_out/r03c_bench_C5.err; cut -c1-900 gpurun_out/r03c_bench_C5_verify.json
