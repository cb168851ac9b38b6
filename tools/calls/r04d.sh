#!/bin/bash
# round 4, call d: does the guard build's wrong plan depend on what fresh memory holds?  poison build through the suite;
# non-blocking ingest; first end-to-end chain at C3; store writer with the thread cap
set -u
O=gpurun_out/r04d; mkdir -p $O
export HB_LIB_PATH=$PWD/stract_amd/lib/libhyperball_guard.so
for fill in A5 00 FF; do
  HB_GUARD_FILL=$fill timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/guard_smoke_fill$fill.log 2>&1; echo "guard smoke, fresh memory = 0x$fill: rc=$?"; tail -1 $O/guard_smoke_fill$fill.log | cut -c1-260
done
export HB_LIB_PATH=$PWD/stract_amd/lib/libhyperball_poison.so
timeout 900 python -m pytest tests -m gpu -q > $O/poison_pytest_gpu.log 2>&1; echo "poison pytest rc=$?"; grep -E "passed|failed" $O/poison_pytest_gpu.log | tail -1
unset HB_LIB_PATH
timeout 400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "shipped pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
HB_TRACE_INGEST=1 timeout 300 python tools/ingest_bench.py C3 --out $O/ingest_C3.json > /dev/null 2> $O/ingest_C3.err; echo "ingest C3 rc=$?"; grep "append of" $O/ingest_C3.err | sed -n '2,4p' | cut -c1-250
timeout 600 python bench.py --steps 10 --warmup 3 --c4-leg off --cpu-seconds 5 > $O/bench_C3_e2e.json 2> $O/bench_C3_e2e.err; echo "bench C3 + e2e rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04d/bench_C3_e2e.json").read().strip().splitlines()[-1])
    print("value", d["value"], "parity", d["parity_bit_exact"], "input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s","pinned_h2d_GBs")})
    print("e2e", json.dumps(d["detail"].get("end_to_end")))
except Exception as e:
    print("no bench line:", e)
PY
mkdir -p /tmp/sb && HB_TRACE_STORE=1 timeout 300 python tools/store_bench.py 20000000 --dir /tmp/sb --check 100 > $O/store_bench_disk.json 2> $O/store_bench_disk.err; echo "store disk rc=$?"; cat $O/store_bench_disk.err | cut -c1-120; cat $O/store_bench_disk.json; rm -rf /tmp/sb
timeout 200 python tools/record_stress.py C3 --rounds 5 --tag shipped --out $O/stress_shipped.json > /dev/null 2> $O/stress_shipped.err; echo "stress shipped rc=$?"
