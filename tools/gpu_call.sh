mkdir -p gpurun_out
timeout 75 python -m pytest tests -m gpu -x -q > gpurun_out/r03zz_pytest_gpu_final.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r03zz_pytest_gpu_final.log | cut -c1-200
