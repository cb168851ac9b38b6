#!/bin/bash
# round 6, call i: (1) full GPU suite; (2) A/B of the pass-0 node-row kernel's occupancy (4 = default, 5, 6 waves per SIMD; -DHB_P0_NODE_WAVES);
# (3) C4 kernel trace + HBM / L2 counters of every kernel on this tree (tools/profile.sh, PMC_SMALL)
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06i_pytest_gpu.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r06i_pytest_gpu.log | cut -c1-300
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
for V in w4 w5 w6; do
  L=""; [ $V != w4 ] && L="stract_amd/lib/libhyperball_$V.so"
  HB_LIB_PATH=$L timeout 900 python bench.py --config C4 --steps 5 --warmup 2 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off > gpurun_out/r06i_bench_C4_$V.json 2> gpurun_out/r06i_bench_C4_$V.err; echo "$V rc=$?"
  show gpurun_out/r06i_bench_C4_$V.json
done
PMC_SMALL=1 tools/profile.sh C4 r06i > gpurun_out/r06i_profile_C4.log 2>&1; tail -40 gpurun_out/r06i_profile_C4.log | cut -c1-180
echo "total $(( $(date +%s) - T0 )) s"
