mkdir -p gpurun_out
timeout 70 python -m pytest tests/test_gpu.py -m gpu -x -q -k "store_harmonic_writes or fixture" > gpurun_out/r03z_pytest_store.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r03z_pytest_store.log | cut -c1-300
