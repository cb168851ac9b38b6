set -x
mkdir -p gpurun_out
python tools/sweep.py LT "0:0:" "0:0:" > gpurun_out/r03o_sweep_rows_5waves_LT.txt 2>&1; cut -c1-200 gpurun_out/r03o_sweep_rows_5waves_LT.txt
python tools/sweep.py C3 "0:0:" "0:0:" > gpurun_out/r03o_sweep_rows_5waves_C3.txt 2>&1; cut -c1-200 gpurun_out/r03o_sweep_rows_5waves_C3.txt
