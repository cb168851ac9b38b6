set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03d_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r03d_pytest_gpu.log | tail -2; grep -E "^(FAILED|ERROR)|Error" gpurun_out/r03d_pytest_gpu.log | head -5
python __graft_entry__.py smoke > gpurun_out/r03d_smoke.txt 2>&1; tail -1 gpurun_out/r03d_smoke.txt
python tools/sweep.py C3 "0:0:" "0:0:0,256" > gpurun_out/r03d_sweep_C3.txt 2>&1; cut -c1-500 gpurun_out/r03d_sweep_C3.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r03d_bench_default.json 2> gpurun_out/r03d_bench_default.err; tail -c 800 gpurun_out/r03d_bench_default.err; cut -c1-400 gpurun_out/r03d_bench_default.json
bash tools/profile.sh C3 r03d > gpurun_out/r03d_profile.log 2>&1; tail -3 gpurun_out/r03d_profile.log | cut -c1-200
python bench.py --config LT --steps 5 --warmup 2 --c4-leg off --pass-log gpurun_out/r03d_passes_LT.json > gpurun_out/r03d_bench_LT.json 2>/dev/null; cut -c1-300 gpurun_out/r03d_bench_LT.json
