#!/bin/bash
# round 6, call k: the compact result image (one entry per node with in-edges when results travel in stages): (1) full GPU suite;
# (2) C4 / C3 / LT timing with the HB_TRACE_RESULTS timeline; (3) C4 through the record boundary with the whole end-to-end chain traced
# (HB_TRACE_INGEST / HB_TRACE_STORE: where load and store emission spend their time); (4) first run of a fresh process at C3
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06k_pytest_gpu.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r06k_pytest_gpu.log | cut -c1-300
show() {
python - "$1" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    det=d["detail"]
    print(f, d["value"], "GTEPS", d["ms_per_step"], "ms; first", d.get("first_run_ms"), "loop", det.get("ms_loop_per_step"), "gpu", det.get("ms_gpu_passes_per_step"), "finish", det.get("ms_finish_per_step"), "parity", (d.get("parity") or {}).get("bit_exact"))
    print(" per pass", [(p["t"],p["mode"],round(p["ms"],3),round(p["ms_level1_or_expand"],3),round(p["ms_node_rows"],3)) for p in d["roofline"]["per_pass"]][:10])
    e=det.get("end_to_end")
    if e: print(" e2e", {k:e[k] for k in e if k.startswith("s_") and k != "s_results_and_ranks" or k.startswith("ms_") or k in ("graph_ok","same_result_as_record_leg","stores_read_back_ok")})
except Exception as e: print(f,"failed",e)
PY
}
for CFG in C4 C3 LT; do
  ST=5; [ $CFG != C4 ] && ST=20
  HB_TRACE_RESULTS=1 timeout 900 python bench.py --config $CFG --steps $ST --warmup 2 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off > gpurun_out/r06k_bench_$CFG.json 2> gpurun_out/r06k_bench_$CFG.err; echo "$CFG rc=$?"
  show gpurun_out/r06k_bench_$CFG.json
  grep "hb results" gpurun_out/r06k_bench_$CFG.err | grep -v "pass [0-3] returned" | tail -7 | cut -c1-160
done
HB_TRACE_STORE=1 HB_TRACE_INGEST=1 timeout 1500 python bench.py --config C4 --steps 3 --warmup 1 --cpu-seconds 0 --c3-leg off --end-to-end on > gpurun_out/r06k_bench_C4_e2e.json 2> gpurun_out/r06k_bench_C4_e2e.err; echo "e2e rc=$?"
show gpurun_out/r06k_bench_C4_e2e.json
grep "hb store\|hb webgraph\|hb state\|hb finalize\|hb ingest" gpurun_out/r06k_bench_C4_e2e.err | tail -60 | cut -c1-250
timeout 600 python tools/first_run_probe.py > gpurun_out/r06k_first_run_probe_C3.json 2> gpurun_out/r06k_first_run_probe_C3.err; echo "probe rc=$?"; cut -c1-700 gpurun_out/r06k_first_run_probe_C3.json
echo "total $(( $(date +%s) - T0 )) s"
