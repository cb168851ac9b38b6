#!/bin/bash
# round 6, call g: pass 0 with the three-stage pipeline (row pointers two tiles ahead, first batch one tile ahead): node-row kernel compiled
# for 4 waves per SIMD (9 spilled VGPRs) vs 3 (none); C4 + C3 timing, parity tests first
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 300 python -m pytest tests/test_gpu.py -m gpu -x -q -k "test_per_pass_state_matches_oracle or test_c2" > gpurun_out/r06g_pytest_gpu.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; tail -2 gpurun_out/r06g_pytest_gpu.log | cut -c1-300
for V in w4 w3; do
  L=""; [ $V = w3 ] && L="stract_amd/lib/libhyperball_w3.so"
  for CFG in C3 C4; do
    ST=5; [ $CFG = C3 ] && ST=20
    HB_LIB_PATH=$L timeout 900 python bench.py --config $CFG --steps $ST --warmup 2 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off > gpurun_out/r06g_bench_${CFG}_$V.json 2> gpurun_out/r06g_bench_${CFG}_$V.err; echo "$CFG $V rc=$?"
    python - $CFG $V <<'PY'
import json,sys
c,v=sys.argv[1:3]
try:
    d=json.loads([l for l in open("gpurun_out/r06g_bench_%s_%s.json"%(c,v)) if l.startswith("{")][-1])
    print(c,v, d["value"], "GTEPS", d["ms_per_step"], "ms; first", d.get("first_run_ms"), "parity", d.get("parity",{}).get("bit_exact"))
    print(" per pass", [(p["t"],p["mode"],p["ms"],p["ms_level1_or_expand"],p["ms_node_rows"]) for p in d["roofline"]["per_pass"]][:3])
except Exception as e: print(c,v,"failed",e)
PY
  done
done
echo "total $(( $(date +%s) - T0 )) s"
