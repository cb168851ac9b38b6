#!/bin/bash
# round 6, call p: the GPU suite of the final tree under the debug builds (never shipped): red zones around every device buffer checked at the end of
# every C-ABI call, and device-side checks of every gathered index (make redzone bounds)
set -u
mkdir -p gpurun_out
T0=$(date +%s)
for B in redzone bounds; do
  HB_LIB_PATH=stract_amd/lib/libhyperball_$B.so timeout 1200 python -m pytest tests/test_gpu.py -m gpu -x -q > gpurun_out/r06p_pytest_gpu_$B.log 2>&1; echo "$B rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r06p_pytest_gpu_$B.log | cut -c1-300
done
echo "total $(( $(date +%s) - T0 )) s"
