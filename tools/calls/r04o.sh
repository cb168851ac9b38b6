#!/bin/bash
# round 4, call o: the fence-free table protocol (8-byte agent atomics on both sides): suite, C3 and C4 through the record boundary
set -u
O=gpurun_out/r04o; mkdir -p $O
timeout 200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "shipped pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
HB_TRACE_INGEST=1 timeout 120 python tools/ingest_bench.py C3 --out $O/ingest_C3.json > /dev/null 2> $O/ingest_C3.err; echo "ingest C3 rc=$?"
HB_TRACE_INGEST=1 timeout 300 python tools/ingest_bench.py C4 --out $O/ingest_C4.json > /dev/null 2> $O/ingest_C4.err; echo "ingest C4 rc=$?"
grep "append of" $O/ingest_C4.err | tail -3 | cut -c1-200
python - <<'PY'
import json
for c in ("C3","C4"):
    try:
        d=json.load(open("gpurun_out/r04o/ingest_%s.json"%c))
        print(c, {k:d["boundary"][k] for k in ("s_append_edges","s_finalize","append_GBs","records_per_s_library","stats_ok","n","m_eff")}, d["parity"]["bit_exact"])
    except Exception as e:
        print(c, "no result:", e)
PY
