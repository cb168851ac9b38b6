#!/bin/bash
# round 6, call s: sweep passes in place, the copy-in kernel as a bitmap scan + LDS row lists (call r: it tested one bit per quad over ALL rows - 0.35 ms per launch at C4): (1) full GPU suite; (2) C4 / C3 / LT in place
# (default) vs double-buffered (experiments build, tune[1] bit 29) on the same box; (3) the fuzzer with the new switch randomised
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06s_pytest_gpu.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; grep -n "passed\|failed" gpurun_out/r06s_pytest_gpu.log | tail -2
show() {
python - "$1" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); det=d["detail"]
    print(f, d["value"], "GTEPS", d["ms_per_step"], "ms; first", d.get("first_run_ms"), "loop", det.get("ms_loop_per_step"), "gpu", det.get("ms_gpu_passes_per_step"), "finish", det.get("ms_finish_per_step"))
    print(" per pass", [(p["t"],p["mode"],round(p["ms"],3),round(p["ms_level1_or_expand"],3),round(p["ms_node_rows"],3)) for p in d["roofline"]["per_pass"]][5:12])
except Exception as e: print(f,"failed",e)
PY
}
for CFG in C4 C3 LT; do
  ST=5; [ $CFG != C4 ] && ST=20
  for V in inplace:0 double_buffered:536870912; do
    N=${V%%:*}; T=${V##*:}
    HB_LIB_PATH=stract_amd/lib/libhyperball_exp.so timeout 900 python bench.py --config $CFG --steps $ST --warmup 2 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off --tune 0,$T > gpurun_out/r06s_bench_${CFG}_$N.json 2> gpurun_out/r06s_bench_${CFG}_$N.err; echo "$CFG $N rc=$?"
    show gpurun_out/r06s_bench_${CFG}_$N.json
  done
done
timeout 600 python bench.py --config C4 --steps 5 --warmup 2 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off --verify > gpurun_out/r06s_bench_C4_product_verify.json 2> gpurun_out/r06s_bench_C4_product_verify.err; echo "C4 product verify rc=$?"
show gpurun_out/r06s_bench_C4_product_verify.json
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r06s_bench_C4_product_verify.json") if l.startswith("{")][-1]); print(" parity", d.get("parity"))
except Exception as e: print("failed", e)
PY
for MODE in passes mixed records; do
  HB_LIB_PATH=stract_amd/lib/libhyperball_exp.so timeout 300 python tools/diff_fuzz.py --mode $MODE --seconds 60 --seed 608 > gpurun_out/r06s_diff_fuzz_$MODE.txt 2>&1; echo "fuzz $MODE rc=$?"; tail -1 gpurun_out/r06s_diff_fuzz_$MODE.txt | cut -c1-300
done
echo "total $(( $(date +%s) - T0 )) s"
