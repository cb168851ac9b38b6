#!/bin/bash
# round 6, call e: GPU suite on the tree with the sort-based transposition + 6-bit exchange; kernel trace of C4 (the per-kernel table of
# round 6); where hb_load's state stage goes (HB_TRACE_INGEST laps), transposition by sort vs by scatter (experiments bit 25)
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06e_pytest_gpu.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r06e_pytest_gpu.log | cut -c1-300
HB_TRACE_INGEST=1 tools/trace.sh r06e_C4 python bench.py --config C4 --steps 2 --warmup 1 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off
grep -h "hb state\|hb finalize" gpurun_out/prof_r06e_C4/trace.log | head -30
echo "--- scatter form"
HB_TRACE_INGEST=1 timeout 900 python bench.py --config C4 --steps 3 --warmup 1 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off --tune 0,33554432 > gpurun_out/r06e_bench_C4_scatter.json 2> gpurun_out/r06e_bench_C4_scatter.err; echo "rc=$?"
grep -h "hb state" gpurun_out/r06e_bench_C4_scatter.err | head -20
python - <<'PY'
import json
for v in ("scatter",):
    try:
        d=json.loads([l for l in open("gpurun_out/r06e_bench_C4_%s.json"%v) if l.startswith("{")][-1])
        print(v, d["value"], "GTEPS", d["ms_per_step"], "ms; plan", d["detail"]["ms_plan"], "state", d["detail"]["ms_h2d"])
        print(" per pass", [(p["t"],p["mode"],p["ms"],p["ms_level1_or_expand"],p["ms_node_rows"]) for p in d["roofline"]["per_pass"]])
    except Exception as e: print(v,"failed",e)
PY
echo "total $(( $(date +%s) - T0 )) s"
