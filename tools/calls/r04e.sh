#!/bin/bash
# round 4, call e: bisect the allocation behind the guard build's failure; ingest with sleeping OpenMP workers; C3 end to end again
set -u
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python tools/guard_bisect.py --out $O/guard_bisect.json > $O/guard_bisect.log 2>&1; echo "bisect rc=$?"; tail -30 $O/guard_bisect.log | cut -c1-200
HB_TRACE_INGEST=1 timeout 300 python tools/ingest_bench.py C3 --out $O/ingest_C3.json > /dev/null 2> $O/ingest_C3.err; echo "ingest C3 rc=$?"; grep "append of" $O/ingest_C3.err | sed -n '2,3p' | cut -c1-250
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04e/ingest_C3.json"))
print({k:d["boundary"][k] for k in ("s_append_edges","s_finalize","append_GBs","records_per_s_library","ms_ingest_reduce","ms_plan","ms_h2d_state")}, d["parity"]["bit_exact"])
PY
timeout 600 python bench.py --steps 10 --warmup 3 --c4-leg off --cpu-seconds 5 > $O/bench_C3_e2e.json 2> $O/bench_C3_e2e.err; echo "bench C3 + e2e rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04e/bench_C3_e2e.json").read().strip().splitlines()[-1])
    print("value", d["value"], "parity", d["parity_bit_exact"], "input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s","pinned_h2d_GBs")})
    print("e2e", json.dumps(d["detail"].get("end_to_end")))
except Exception as e:
    print("no bench line:", e)
PY
