#!/bin/bash
# round 4, call j: suite with the allocator-pressure test; C4 records again; kernel trace + PMC of the final build at C3
set -u
O=gpurun_out/r04j; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "shipped pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
HB_TRACE_INGEST=1 timeout 900 python tools/ingest_bench.py C4 --out $O/ingest_C4.json > /dev/null 2> $O/ingest_C4.err; echo "ingest C4 rc=$?"
grep "hb finalize\|hb ingest\] [a-z]" $O/ingest_C4.err | cut -c1-200 | head -12
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04j/ingest_C4.json"))
print({k:d["boundary"][k] for k in ("s_append_edges","s_finalize","append_GBs","records_per_s_library","ms_ingest_reduce","ms_plan","ms_h2d_state","peak_bytes_per_record","allocator_held_peak_bytes")}, d["parity"], d["run"])
PY
PMC_SMALL=1 bash tools/profile.sh C3 r04j > $O/profile.log 2>&1; echo "profile rc=$?"; tail -12 $O/profile.log | cut -c1-200
