#!/bin/bash
# round 6, call m: store emission with the fst built beside the blob files, booked blocks staged into large writes, parallel group scan:
# (1) store tests; (2) tools/store_bench.py at C4's result count on an unloaded host; (3) the C4 end-to-end chain, traced; (4) the differential
# fuzzer on the MI355X against this tree (compact result image, snapshot policy)
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 600 python -m pytest tests/test_store.py tests/test_gpu.py -m "gpu or not gpu" -x -q -k "store or result" > gpurun_out/r06m_pytest_store.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r06m_pytest_store.log | cut -c1-200
HB_TRACE_STORE=1 timeout 900 python tools/store_bench.py 79000000 --dir /dev/shm/hb_sb > gpurun_out/r06m_store_bench_79M.txt 2>&1; echo "store_bench rc=$?"; cat gpurun_out/r06m_store_bench_79M.txt | tail -16; rm -rf /dev/shm/hb_sb
HB_TRACE_STORE=1 timeout 1500 python bench.py --config C4 --steps 3 --warmup 1 --cpu-seconds 0 --c3-leg off --end-to-end on > gpurun_out/r06m_bench_C4_e2e.json 2> gpurun_out/r06m_bench_C4_e2e.err; echo "e2e rc=$?"
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r06m_bench_C4_e2e.json") if l.startswith("{")][-1]); det=d["detail"]
    print("C4", d["value"], "GTEPS", d["ms_per_step"], "ms; first", d.get("first_run_ms"))
    e=det.get("end_to_end"); print(" e2e C4", {k:v for k,v in (e or {}).items() if (k.startswith("s_") and k!="s_results_and_ranks") or k in ("graph_ok","same_result_as_record_leg","stores_read_back_ok")})
except Exception as e: print("failed", e)
PY
grep "hb store" gpurun_out/r06m_bench_C4_e2e.err | tail -16
for MODE in passes records mixed; do
  HB_LIB_PATH=stract_amd/lib/libhyperball_exp.so timeout 300 python tools/diff_fuzz.py --mode $MODE --seconds 60 --seed 606 > gpurun_out/r06m_diff_fuzz_$MODE.txt 2>&1; echo "fuzz $MODE rc=$?"; tail -2 gpurun_out/r06m_diff_fuzz_$MODE.txt | cut -c1-300
done
echo "total $(( $(date +%s) - T0 )) s"
