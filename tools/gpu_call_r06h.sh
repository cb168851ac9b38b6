#!/bin/bash
# round 6, call h: (1) full GPU suite; (2) C4 through the record boundary: pass 0 with init_level1_kernel, sweep rows in three round trips,
# id_lo from the device ingest (HB_TRACE_INGEST laps); (3) A/B: sweep rows round by round (bit 26), pass 0 level 1 generic (bit 27); (4) C3, LT
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06h_pytest_gpu.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r06h_pytest_gpu.log | cut -c1-300
show() {
python - "$1" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, d["value"], "GTEPS", d["ms_per_step"], "ms; first", d.get("first_run_ms"), "plan", d["detail"].get("ms_plan"), "state", d["detail"].get("ms_h2d"), "parity", (d.get("parity") or {}).get("bit_exact"))
    print(" per pass", [(p["t"],p["mode"],p["ms"],p["ms_level1_or_expand"],p["ms_node_rows"]) for p in d["roofline"]["per_pass"]])
except Exception as e: print(f,"failed",e)
PY
}
HB_TRACE_INGEST=1 timeout 900 python bench.py --config C4 --steps 5 --warmup 2 --cpu-seconds 0 --c3-leg off --end-to-end off > gpurun_out/r06h_bench_C4_records.json 2> gpurun_out/r06h_bench_C4_records.err; echo "rc=$?"
show gpurun_out/r06h_bench_C4_records.json
grep -h "hb state\|hb finalize" gpurun_out/r06h_bench_C4_records.err | head -16
for V in rows_round_by_round:67108864 p0_generic:134217728; do
  N=${V%%:*}; T=${V##*:}
  timeout 900 python bench.py --config C4 --steps 5 --warmup 2 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off --tune 0,$T > gpurun_out/r06h_bench_C4_$N.json 2> gpurun_out/r06h_bench_C4_$N.err; echo "rc=$?"
  show gpurun_out/r06h_bench_C4_$N.json
done
timeout 600 python bench.py --config C3 --steps 20 --warmup 3 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off > gpurun_out/r06h_bench_C3.json 2> gpurun_out/r06h_bench_C3.err; echo "rc=$?"
show gpurun_out/r06h_bench_C3.json
for V in new:0 rows_round_by_round:67108864; do
  N=${V%%:*}; T=${V##*:}
  timeout 600 python bench.py --config LT --steps 10 --warmup 2 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off --tune 0,$T > gpurun_out/r06h_bench_LT_$N.json 2> gpurun_out/r06h_bench_LT_$N.err; echo "rc=$?"
  python - gpurun_out/r06h_bench_LT_$N.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], d["value"], "GTEPS", d["ms_per_step"], "ms, passes", d["config"].get("passes_T"))
except Exception as e: print("failed",e)
PY
done
echo "total $(( $(date +%s) - T0 )) s"
