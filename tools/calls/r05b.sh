#!/bin/bash
# round 5, call b: GPU suite with the staged result download + ABI 5, then the new default bench line (C4 headline, oracle to
# convergence = final-list parity at 100M / 2B, C3 as the child leg)
set -u
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 1700 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r05b/bench_default.json").read().strip().splitlines()[-1])
    print("MAIN", d["config"]["workload"][:40], "value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity_bit_exact"], (d["parity"] or {}).get("scope"))
    print("roof", d["roofline"]["frac"], "dominant", d["roofline"]["dominant_kernel"]["frac"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:200])
    print("finish ms", d["detail"]["ms_finish_per_step"], "loop", d["detail"]["ms_loop_per_step"], "gpu passes", d["detail"]["ms_gpu_passes_per_step"])
    print("per pass", [(p["t"], p["mode"], p["ms"]) for p in d["roofline"]["per_pass"]])
    print("input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s")})
    print("e2e", json.dumps(d["detail"].get("end_to_end")))
    c3=d["detail"].get("c3") or {}
    print("C3", {k:c3.get(k) for k in ("value","ms_per_step","parity_bit_exact","error")})
except Exception as e:
    print("no bench line:", e); print(open("gpurun_out/r05b/bench_default.err").read()[-2500:])
PY
