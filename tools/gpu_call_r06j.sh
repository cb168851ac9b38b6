#!/bin/bash
# round 6, call j: (1) first results snapshot when the next pass is no longer dense (this tree) vs at A <= 90 % (libhyperball_snap90.so = the tree
# before it): C4, C3, LT with the HB_TRACE_RESULTS timeline; (2) the size[] timing probe of the dense node rows (experiments build, tune[1] bit 28;
# WRONG RESULTS by design - what a 16 B / row state diet could buy at most); (3) the driver-style default line of this tree
set -u
mkdir -p gpurun_out
T0=$(date +%s)
show() {
python - "$1" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    det=d["detail"]
    print(f, d["value"], "GTEPS", d["ms_per_step"], "ms; first", d.get("first_run_ms"), "loop", det.get("ms_loop_per_step"), "gpu", det.get("ms_gpu_passes_per_step"), "finish", det.get("ms_finish_per_step"), "parity", (d.get("parity") or {}).get("bit_exact"))
    print(" per pass", [(p["t"],p["mode"],round(p["ms"],3),round(p["ms_level1_or_expand"],3),round(p["ms_node_rows"],3)) for p in d["roofline"]["per_pass"]][:10])
except Exception as e: print(f,"failed",e)
PY
}
for CFG in C4 C3 LT; do
  ST=5; [ $CFG != C4 ] && ST=20
  for V in new snap90; do
    L=""; [ $V = snap90 ] && L="stract_amd/lib/libhyperball_snap90.so"
    HB_TRACE_RESULTS=1 HB_LIB_PATH=$L timeout 900 python bench.py --config $CFG --steps $ST --warmup 2 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off > gpurun_out/r06j_bench_${CFG}_$V.json 2> gpurun_out/r06j_bench_${CFG}_$V.err; echo "$CFG $V rc=$?"
    show gpurun_out/r06j_bench_${CFG}_$V.json
    grep "hb results" gpurun_out/r06j_bench_${CFG}_$V.err | grep -v "pass [0-3] returned" | tail -9 | cut -c1-160
  done
done
for V in base:0 nosize:268435456; do
  N=${V%%:*}; T=${V##*:}
  HB_LIB_PATH=stract_amd/lib/libhyperball_exp.so timeout 900 python bench.py --config C4 --steps 5 --warmup 2 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off --tune 0,$T > gpurun_out/r06j_bench_C4_exp_$N.json 2> gpurun_out/r06j_bench_C4_exp_$N.err; echo "exp $N rc=$?"
  show gpurun_out/r06j_bench_C4_exp_$N.json
done
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06j_bench_default.json 2> gpurun_out/r06j_bench_default.err; echo "default rc=$?"
show gpurun_out/r06j_bench_default.json
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r06j_bench_default.json") if l.startswith("{")][-1])
    print("roofline", {k:v for k,v in d["roofline"].items() if not isinstance(v,(list,dict))})
    print("cpu_baseline", d.get("cpu_baseline")); print("parity", d.get("parity"))
    e=d["detail"].get("end_to_end"); print("e2e C4", {k:v for k,v in (e or {}).items() if k.startswith("s_") or k.startswith("ms_")})
    c3=d["detail"].get("c3") or {}; print("c3", c3.get("value"), c3.get("ms_per_step"), c3.get("first_run_ms")); e=c3.get("end_to_end"); print("e2e C3", {k:v for k,v in (e or {}).items() if k.startswith("s_") or k.startswith("ms_")})
except Exception as e: print("failed", e)
PY
grep "hb store\|hb load\|hb finalize" gpurun_out/r06j_bench_default.err | tail -30 | cut -c1-160
echo "total $(( $(date +%s) - T0 )) s"
