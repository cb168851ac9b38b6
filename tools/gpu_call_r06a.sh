#!/bin/bash
# round 6, call a: the GPU suite on the new tree, the expansion micro-benchmark, the first-run probe, quick C3 / C4 lines
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06a_pytest_gpu.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r06a_pytest_gpu.log | cut -c1-300
timeout 120 tools/xcd_atomic_bench.bin > gpurun_out/r06a_xcd_atomic_bench.txt 2>&1; echo "xcd bench rc=$?"; cat gpurun_out/r06a_xcd_atomic_bench.txt
timeout 300 python tools/first_run_probe.py C3 > gpurun_out/r06a_first_run_probe_C3.json 2> gpurun_out/r06a_first_run_probe_C3.err; echo "probe rc=$?"; cat gpurun_out/r06a_first_run_probe_C3.json | cut -c1-1500
timeout 400 python bench.py --config C3 --steps 10 --warmup 3 --input dense --cpu-seconds 0 --end-to-end off --c3-leg off > gpurun_out/r06a_bench_C3_quick.json 2> gpurun_out/r06a_bench_C3_quick.err; echo "C3 rc=$?"
python - <<'PY'
import json
for cfg in ("C3",):
    try:
        d=json.loads([l for l in open("gpurun_out/r06a_bench_%s_quick.json"%cfg) if l.startswith("{")][-1])
        print(cfg, d["value"], "GTEPS", d["ms_per_step"], "ms first", d.get("first_run_ms"), "finish", d["detail"]["ms_finish_per_step"], d["detail"].get("ms_finish_first_run"))
        print(" per pass", [(p["t"],p["mode"],p["ms"],p["ms_level1_or_expand"],p["ms_node_rows"]) for p in d["roofline"]["per_pass"]])
    except Exception as e: print(cfg,"failed",e)
PY
HB_TRACE_RESULTS=1 timeout 900 python bench.py --config C4 --steps 5 --warmup 2 --input dense --cpu-seconds 0 --end-to-end off --c3-leg off > gpurun_out/r06a_bench_C4_quick.json 2> gpurun_out/r06a_bench_C4_quick.err; echo "C4 rc=$?"
python - <<'PY'
import json
for cfg in ("C4",):
    try:
        d=json.loads([l for l in open("gpurun_out/r06a_bench_%s_quick.json"%cfg) if l.startswith("{")][-1])
        print(cfg, d["value"], "GTEPS", d["ms_per_step"], "ms first", d.get("first_run_ms"), "finish", d["detail"]["ms_finish_per_step"], d["detail"].get("ms_finish_first_run"))
        print(" loop", d["detail"]["ms_loop_per_step"], "gpu passes", d["detail"]["ms_gpu_passes_per_step"])
        print(" per pass", [(p["t"],p["mode"],p["ms"],p["ms_level1_or_expand"],p["ms_node_rows"]) for p in d["roofline"]["per_pass"]])
    except Exception as e: print(cfg,"failed",e)
PY
tail -5 gpurun_out/r06a_bench_C4_quick.err | cut -c1-300
echo "total $(( $(date +%s) - T0 )) s"
