#!/bin/bash
# round 4, call k: the default bench line of the final build (what the driver runs), smoke, and 10 more record-input rounds
set -u
O=gpurun_out/r04k; mkdir -p $O
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 1700 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04k/bench_default.json").read().strip().splitlines()[-1])
    print("C3 value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity_bit_exact"], "roof", d["roofline"]["frac"], "dominant", d["roofline"]["dominant_kernel"]["frac"], "traffic", d["roofline"]["traffic"])
    print("C3 input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s","ingest_peak_bytes_per_record","allocator_held_peak_bytes")})
    print("C3 e2e", json.dumps(d["detail"].get("end_to_end")))
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
    c4=d["detail"].get("c4") or {}
    print("C4", {k:c4.get(k) for k in ("value","ms_per_step","parity_bit_exact","error")}, "input", {k:(c4.get("input") or {}).get(k) for k in ("s_append_edges","s_finalize","records_per_s","ingest_peak_bytes_per_record","allocator_held_peak_bytes")})
    print("C4 e2e", json.dumps(c4.get("end_to_end")))
except Exception as e:
    print("no bench line:", e); print(open("gpurun_out/r04k/bench_default.err").read()[-1500:])
PY
timeout 300 python tools/record_stress.py C3,LT --rounds 5 --tag shipped --out $O/stress_shipped.json > /dev/null 2> $O/stress_shipped.err; echo "stress shipped rc=$?"
