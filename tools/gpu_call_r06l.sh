#!/bin/bash
# round 6, call l: the round's tree as the driver sees it - (1) full GPU suite; (2) `python bench.py --steps 20 --warmup 5` (C4 headline through the
# record boundary, oracle to convergence, C3 leg, both end-to-end chains - with the fst fix); (3) rocprofv3 kernel trace + HBM / L2 counters of the same
# tree at C4 (tools/profile.sh); (4) C5 (BASELINE configs[4]) through the record boundary with --verify
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06l_pytest_gpu.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; grep -n "passed\|failed" gpurun_out/r06l_pytest_gpu.log | tail -2
HB_TRACE_STORE=1 timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06l_bench_default.json 2> gpurun_out/r06l_bench_default.err; echo "default rc=$?"
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r06l_bench_default.json") if l.startswith("{")][-1]); det=d["detail"]
    print("C4", d["value"], "GTEPS", d["ms_per_step"], "ms; first", d.get("first_run_ms"), "loop", det.get("ms_loop_per_step"), "gpu", det.get("ms_gpu_passes_per_step"), "finish", det.get("ms_finish_per_step"), "parity", (d.get("parity") or {}).get("bit_exact"))
    print(" per pass", [(p["t"],p["mode"],round(p["ms"],3),round(p["ms_level1_or_expand"],3),round(p["ms_node_rows"],3)) for p in d["roofline"]["per_pass"]][:10])
    print(" roofline", {k:v for k,v in d["roofline"].items() if not isinstance(v,(list,dict))})
    print(" dominant", {k:v for k,v in d["roofline"]["dominant_kernel"].items() if k in ("avg_launch_ms","achieved","frac","traffic","l2_hit_rate")})
    print(" cpu_baseline", {k:v for k,v in (d.get("cpu_baseline") or {}).items() if k in ("value","cores","seconds")})
    print(" input", {k:v for k,v in det["input"].items() if not isinstance(v,(list,dict))})
    e=det.get("end_to_end"); print(" e2e C4", {k:v for k,v in (e or {}).items() if (k.startswith("s_") and k!="s_results_and_ranks") or k.startswith("ms_")})
    c3=det.get("c3") or {}; print(" c3", c3.get("value"), c3.get("ms_per_step"), c3.get("first_run_ms"), (c3.get("parity") or {}).get("bit_exact")); e=c3.get("end_to_end"); print(" e2e C3", {k:v for k,v in (e or {}).items() if (k.startswith("s_") and k!="s_results_and_ranks")})
except Exception as e: print("failed", e)
PY
grep "hb store" gpurun_out/r06l_bench_default.err | tail -22
PMC_SMALL=1 tools/profile.sh C4 r06l > gpurun_out/r06l_profile_C4.log 2>&1; tail -45 gpurun_out/r06l_profile_C4.log | cut -c1-150 | head -30
timeout 2400 python bench.py --config C5 --input records --verify --steps 2 --warmup 1 --end-to-end off --c3-leg off > gpurun_out/r06l_bench_C5_records_verify.json 2> gpurun_out/r06l_bench_C5.err; echo "bench C5 rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r06l_bench_C5_records_verify.json").read().strip().splitlines()[-1])
    print("C5 value", d["value"], "ms/step", d["ms_per_step"], "first", d.get("first_run_ms"), "parity", d["parity"], "roof", d["roofline"]["frac"], "input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s","ingest_peak_bytes_per_record")})
    print(" per pass", [(p["t"],p["mode"],round(p["ms"],2)) for p in d["roofline"]["per_pass"]])
except Exception as e:
    print("no bench line:", e); print(open("gpurun_out/r06l_bench_C5.err").read()[-1500:])
PY
echo "total $(( $(date +%s) - T0 )) s"
