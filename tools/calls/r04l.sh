#!/bin/bash
# round 4, call l: suite (damaged-store test), C4 with the end-to-end chain after the CRC moved beside the streaming
set -u
O=gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "shipped pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
HB_TRACE_INGEST=1 timeout 1500 python bench.py --config C4 --steps 2 --warmup 1 --cpu-seconds 0 --c4-leg off > $O/bench_C4_e2e.json 2> $O/bench_C4_e2e.err; echo "bench C4 rc=$?"
grep "hb webgraph" $O/bench_C4_e2e.err | cut -c1-400
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04l/bench_C4_e2e.json").read().strip().splitlines()[-1])
    print("C4 value", d["value"], "input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s","ingest_peak_bytes_per_record")})
    print("C4 e2e", json.dumps(d["detail"].get("end_to_end")))
except Exception as e:
    print("no bench line:", e); print(open("gpurun_out/r04l/bench_C4_e2e.err").read()[-1500:])
PY
