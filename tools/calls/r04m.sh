#!/bin/bash
# round 4, call m: last regression of the final tree - GPU suite, smoke, the default bench line
set -u
O=gpurun_out/r04m; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "shipped pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
HB_TRACE_INGEST=1 timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
grep "hb webgraph" $O/bench_default.err | cut -c1-400
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04m/bench_default.json").read().strip().splitlines()[-1])
    print("C3 value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity_bit_exact"], "roof", d["roofline"]["frac"], "dominant", d["roofline"]["dominant_kernel"]["frac"])
    print("C3 input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s","ingest_peak_bytes_per_record")})
    print("C3 e2e", json.dumps(d["detail"].get("end_to_end")))
    c4=d["detail"].get("c4") or {}
    print("C4", {k:c4.get(k) for k in ("value","ms_per_step","parity_bit_exact","error")}, "input", {k:(c4.get("input") or {}).get(k) for k in ("s_append_edges","s_finalize","records_per_s","ingest_peak_bytes_per_record")})
    print("C4 e2e", json.dumps(c4.get("end_to_end")))
except Exception as e:
    print("no bench line:", e); print(open("gpurun_out/r04m/bench_default.err").read()[-1500:])
PY
