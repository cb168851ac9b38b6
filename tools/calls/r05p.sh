#!/bin/bash
# round 5, call p: the final tree - GPU suite, smoke, a short record-boundary line of C3 with parity and the end-to-end chain
set -u
O=gpurun_out/r05p; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py --config C3 --steps 10 --warmup 3 --c3-leg off --cpu-seconds 10 > $O/bench_C3_records_e2e.json 2> $O/bench_C3.err; echo "C3 rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05p/bench_C3_records_e2e.json").read().strip().splitlines()[-1])
print("C3 value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity_bit_exact"], d["parity"]["scope"], "e2e", json.dumps(d["detail"]["end_to_end"])[:400])
PY
