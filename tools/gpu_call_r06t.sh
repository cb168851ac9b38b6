#!/bin/bash
# round 6, call t: the committed tree once more as the driver runs it: GPU suite, __graft_entry__.smoke(), `python bench.py --steps 20 --warmup 5`
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06t_pytest_gpu.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; grep -n "passed\|failed" gpurun_out/r06t_pytest_gpu.log | tail -2
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -3
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06t_bench_default.json 2> gpurun_out/r06t_bench_default.err; echo "default rc=$?"
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r06t_bench_default.json") if l.startswith("{")][-1]); det=d["detail"]
    print("C4", d["value"], "GTEPS", d["ms_per_step"], "ms; first", d.get("first_run_ms"), "loop", det.get("ms_loop_per_step"), "gpu", det.get("ms_gpu_passes_per_step"), "finish", det.get("ms_finish_per_step"), "parity", (d.get("parity") or {}).get("bit_exact"))
    print(" roofline", {k:v for k,v in d["roofline"].items() if k in ("achieved","frac","traffic")}, "dominant", {k:v for k,v in d["roofline"]["dominant_kernel"].items() if k in ("avg_launch_ms","achieved","frac","traffic","l2_hit_rate")})
    e=det.get("end_to_end"); print(" e2e C4", {k:v for k,v in (e or {}).items() if (k.startswith("s_") and k!="s_results_and_ranks")})
    c3=det.get("c3") or {}; print(" c3", c3.get("value"), c3.get("ms_per_step"), c3.get("first_run_ms"), (c3.get("parity") or {}).get("bit_exact"))
except Exception as e: print("failed", e)
PY
echo "total $(( $(date +%s) - T0 )) s"
