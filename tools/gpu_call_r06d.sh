#!/bin/bash
# round 6, call d: GPU suite on the pruned tree (product + experiments builds), then the C4 record + end-to-end leg alone
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06d_pytest_gpu.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; grep -n "passed\|failed" gpurun_out/r06d_pytest_gpu.log | tail -3
python -c "
from stract_amd import _lib
try:
    _lib.load()  # product
    import ctypes
    o=_lib.HbOptions(); o.struct_size=ctypes.sizeof(o); o.device=-1; o.tune[1]=0x100
    h=ctypes.c_void_p(); rc=_lib.load().hb_create(ctypes.byref(o),ctypes.byref(h)); print('product lib with an experiment switch: rc',rc,(_lib.load().hb_last_error(None) or b'')[:80])
except Exception as e: print('ERR',e)
"
HB_TRACE_STORE=1 HB_TRACE_RESULTS=1 timeout 1200 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --c3-leg off > gpurun_out/r06d_bench_C4_e2e.json 2> gpurun_out/r06d_bench_C4_e2e.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r06d_bench_C4_e2e.json") if l.startswith("{")][-1])
    print(d["value"], d["ms_per_step"], "first", d.get("first_run_ms"))
    e=d["detail"]["end_to_end"]; print({k:e[k] for k in e if k.startswith("s_") or k in ("compute_share","graph_ok","same_result_as_record_leg","stores_read_back_ok")})
except Exception as ex: print("failed",ex)
PY
grep -n "hb store\|hb results\] hb_finish" gpurun_out/r06d_bench_C4_e2e.err | tail -14
echo "total $(( $(date +%s) - T0 )) s"
