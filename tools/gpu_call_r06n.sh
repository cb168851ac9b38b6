#!/bin/bash
# round 6, call n: the final tree - (1) full GPU suite; (2) the C4 and C3 end-to-end chains, traced, with the reworked fst writer (implicit key
# tails); (3) the driver-style default line; (4) ranks / multi-process fuzz modes on the MI355X
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06n_pytest_gpu.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; grep -n "passed\|failed" gpurun_out/r06n_pytest_gpu.log | tail -2
for CFG in C4 C3; do
  HB_TRACE_STORE=1 HB_TRACE_INGEST=1 timeout 1500 python bench.py --config $CFG --steps 3 --warmup 1 --cpu-seconds 0 --c3-leg off --end-to-end on > gpurun_out/r06n_bench_${CFG}_e2e.json 2> gpurun_out/r06n_bench_${CFG}_e2e.err; echo "e2e $CFG rc=$?"
  python - $CFG <<'PY'
import json,sys
c=sys.argv[1]
try:
    d=json.loads([l for l in open("gpurun_out/r06n_bench_%s_e2e.json"%c) if l.startswith("{")][-1]); det=d["detail"]
    print(c, d["value"], "GTEPS", d["ms_per_step"], "ms; first", d.get("first_run_ms"))
    e=det.get("end_to_end"); print(" e2e", {k:v for k,v in (e or {}).items() if (k.startswith("s_") and k!="s_results_and_ranks") or k in ("graph_ok","same_result_as_record_leg","stores_read_back_ok")})
except Exception as e: print("failed", e)
PY
  grep "hb store\|hb webgraph" gpurun_out/r06n_bench_${CFG}_e2e.err | tail -17 | cut -c1-330
done
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06n_bench_default.json 2> gpurun_out/r06n_bench_default.err; echo "default rc=$?"
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r06n_bench_default.json") if l.startswith("{")][-1]); det=d["detail"]
    print("C4", d["value"], "GTEPS", d["ms_per_step"], "ms; first", d.get("first_run_ms"), "loop", det.get("ms_loop_per_step"), "gpu", det.get("ms_gpu_passes_per_step"), "finish", det.get("ms_finish_per_step"), "parity", (d.get("parity") or {}).get("bit_exact"))
    print(" per pass", [(p["t"],p["mode"],round(p["ms"],3)) for p in d["roofline"]["per_pass"]][:10])
    print(" roofline", {k:v for k,v in d["roofline"].items() if k in ("achieved","frac","traffic")}, "dominant", {k:v for k,v in d["roofline"]["dominant_kernel"].items() if k in ("avg_launch_ms","achieved","frac","traffic","l2_hit_rate")})
    e=det.get("end_to_end"); print(" e2e C4", {k:v for k,v in (e or {}).items() if (k.startswith("s_") and k!="s_results_and_ranks")})
    c3=det.get("c3") or {}; print(" c3", c3.get("value"), c3.get("ms_per_step"), c3.get("first_run_ms"), (c3.get("parity") or {}).get("bit_exact")); e=c3.get("detail",{}).get("end_to_end") or c3.get("end_to_end"); print(" e2e C3", {k:v for k,v in (e or {}).items() if (k.startswith("s_") and k!="s_results_and_ranks")})
except Exception as e: print("failed", e)
PY
for MODE in ranks tail; do
  HB_LIB_PATH=stract_amd/lib/libhyperball_exp.so timeout 300 python tools/diff_fuzz.py --mode $MODE --seconds 45 --seed 607 > gpurun_out/r06n_diff_fuzz_$MODE.txt 2>&1; echo "fuzz $MODE rc=$?"; tail -1 gpurun_out/r06n_diff_fuzz_$MODE.txt | cut -c1-300
done
echo "total $(( $(date +%s) - T0 )) s"
