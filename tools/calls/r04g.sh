#!/bin/bash
# round 4, call g: GPU suite incl. the multi-process exchange test; the default bench line (C3 + end to end + C4 leg with end to end)
set -u
O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "shipped pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|Error" $O/pytest_gpu.log | head -5 | cut -c1-300
HB_TRACE_INGEST=1 timeout 300 python tools/ingest_bench.py C3 --out $O/ingest_C3.json > /dev/null 2> $O/ingest_C3.err; echo "ingest C3 rc=$?"; grep "append of" $O/ingest_C3.err | sed -n '3,4p' | cut -c1-330
HB_TRACE_INGEST=1 HB_TRACE_STORE=1 timeout 1700 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
grep "hb store\|hb webgraph" $O/bench_default.err | cut -c1-330
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04g/bench_default.json").read().strip().splitlines()[-1])
    print("C3 value", d["value"], "parity", d["parity_bit_exact"], "roof", d["roofline"]["frac"], "input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s","pinned_h2d_GBs","ingest_peak_device_bytes")})
    print("C3 e2e", json.dumps(d["detail"].get("end_to_end")))
    c4=d["detail"].get("c4") or {}
    print("C4", {k:c4.get(k) for k in ("value","ms_per_step","parity_bit_exact","error")}, "input", c4.get("input"))
    print("C4 e2e", json.dumps(c4.get("end_to_end")))
except Exception as e:
    print("no bench line:", e)
PY
