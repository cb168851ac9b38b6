#!/bin/bash
# round 5, call q (re-entry after the container was replaced; call p's output was lost with it): the final tree - GPU suite + smoke.
# 3.7 GPU-minutes were left: both steps carry their own short timeouts.
set -u
O=gpurun_out/r05q; mkdir -p $O
timeout 140 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
timeout 45 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-300
