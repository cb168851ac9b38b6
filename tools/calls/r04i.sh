#!/bin/bash
# round 4, call i: the caching allocator: whole GPU suite, C4 through the record boundary with traces, C3 bench line
set -u
O=gpurun_out/r04i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "shipped pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
HB_TRACE_INGEST=1 HB_PLAN_TIMING=1 timeout 900 python tools/ingest_bench.py C4 --out $O/ingest_C4.json > /dev/null 2> $O/ingest_C4.err; echo "ingest C4 rc=$?"
grep "hb finalize\|hb ingest\] [a-z]\|gpu plan" $O/ingest_C4.err | cut -c1-200 | head -24
grep "append of" $O/ingest_C4.err | tail -1 | cut -c1-330
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04i/ingest_C4.json"))
print({k:d["boundary"][k] for k in ("s_append_edges","s_finalize","append_GBs","records_per_s_library","ms_ingest_reduce","ms_plan","ms_h2d_state","peak_bytes_per_record")}, d["parity"], d["run"])
PY
timeout 600 python bench.py --steps 20 --warmup 5 --c4-leg off > $O/bench_C3.json 2> $O/bench_C3.err; echo "bench C3 rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04i/bench_C3.json").read().strip().splitlines()[-1])
    print("C3 value", d["value"], "parity", d["parity_bit_exact"], "roof", d["roofline"]["frac"], "input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s","ingest_peak_device_bytes")})
    print("C3 e2e", json.dumps(d["detail"].get("end_to_end")))
except Exception as e:
    print("no bench line:", e)
PY
timeout 200 python tools/record_stress.py C3,LT --rounds 4 --tag shipped --out $O/stress_shipped.json > /dev/null 2> $O/stress_shipped.err; echo "stress shipped rc=$?"
