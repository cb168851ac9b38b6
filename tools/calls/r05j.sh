#!/bin/bash
# round 5, call j (= call g on the final tree): the driver-style default line on the (near-)final tree: C4 headline through the record boundary, oracle to
# convergence, end-to-end chain with hb_store_harmonic_results, C3 child leg; result-path and store phases traced to stderr
set -u
O=gpurun_out/r05j; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
HB_TRACE_STORE=1 HB_TRACE_INGEST=1 timeout 2000 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
grep -E "hb store|hb webgraph|hb finalize" $O/bench_default.err | head -40 | cut -c1-300
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r05j/bench_default.json").read().strip().splitlines()[-1])
    print("MAIN", d["config"]["workload"][:40], "value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity_bit_exact"], (d["parity"] or {}).get("scope"))
    print("roof", d["roofline"]["frac"], "dominant", d["roofline"]["dominant_kernel"]["frac"], "traffic", d["roofline"]["traffic"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:260])
    print("finish ms", d["detail"]["ms_finish_per_step"], "loop", d["detail"]["ms_loop_per_step"], "gpu passes", d["detail"]["ms_gpu_passes_per_step"])
    print("per pass", [(p["t"], p["mode"], p["ms"]) for p in d["roofline"]["per_pass"]])
    print("input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s")})
    print("e2e", json.dumps(d["detail"].get("end_to_end")))
    c3=d["detail"].get("c3") or {}
    print("C3", {k:c3.get(k) for k in ("value","ms_per_step","parity_bit_exact","error")}, json.dumps(c3.get("end_to_end")))
except Exception as e:
    print("no bench line:", e); print(open("gpurun_out/r05j/bench_default.err").read()[-2500:])
PY
