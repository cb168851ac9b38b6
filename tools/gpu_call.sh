set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03b_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r03b_pytest_gpu.log | tail -2
python tools/sweep.py C3 "0:0:" "0:0:0,256" "0:0:0,512" "0:0:0,768" "0:0:0,0,70" "0:0:0,0,90" "0:0:0,0,50,0,0,0,5" "0:0:0,0,50,0,0,0,20" "0:0:0,0,101,0,0,0,1000000" "0:0:0,512,101,0,0,0,1000000" > gpurun_out/r03b_sweep_C3.txt 2>&1; cut -c1-900 gpurun_out/r03b_sweep_C3.txt
python tools/sweep.py LT "0:0:" "0:0:0,768" "0:0:0,0,50,0,0,0,5" "0:0:0,0,50,0,0,0,20" > gpurun_out/r03b_sweep_LT.txt 2>&1; cut -c1-600 gpurun_out/r03b_sweep_LT.txt
python tools/sweep.py C4 "0:0:" "0:0:0,768" "0:0:0,0,90" > gpurun_out/r03b_sweep_C4.txt 2>&1; cut -c1-1200 gpurun_out/r03b_sweep_C4.txt
for c in HBM_8GiB HBM_8GiB_sorted_rows window_4MiB_in_8GiB window_32MiB_in_8GiB window_256MiB_in_8GiB window_1GiB_in_8GiB; do ./tools/gather_bench.bin $c; done > gpurun_out/r03b_gather_windows.txt 2>&1; cat gpurun_out/r03b_gather_windows.txt
HB_TRACE_INGEST=1 python tools/ingest_bench.py C4 --out gpurun_out/r03b_ingest_C4.json > gpurun_out/r03b_ingest_C4.log 2>&1; grep "hb ingest" gpurun_out/r03b_ingest_C4.log; tail -1 gpurun_out/r03b_ingest_C4.log | cut -c1-1200
