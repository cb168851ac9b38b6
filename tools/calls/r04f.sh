#!/bin/bash
# round 4, call f: guard build with the runtime's own copies; scans / select / atomics on guarded memory; ingest + e2e traces
set -u
O=gpurun_out/r04f; mkdir -p $O
timeout 200 tools/guard_selftest_copies0.bin > $O/guard_selftest_copies0.txt 2>&1; echo "selftest copies0 rc=$?"; tail -6 $O/guard_selftest_copies0.txt
timeout 200 tools/guard_selftest_copies1.bin > $O/guard_selftest_copies1.txt 2>&1; echo "selftest copies1 rc=$?"; tail -6 $O/guard_selftest_copies1.txt
for v in guard_rt guard; do
  HB_LIB_PATH=$PWD/stract_amd/lib/libhyperball_$v.so HB_GUARD_STRICT_TO=0 timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/${v}_loose_smoke.log 2>&1; echo "$v, all allocations loose: rc=$?"; tail -1 $O/${v}_loose_smoke.log | cut -c1-250
  HB_LIB_PATH=$PWD/stract_amd/lib/libhyperball_$v.so timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/${v}_strict_smoke.log 2>&1; echo "$v, all strict: rc=$?"; tail -1 $O/${v}_strict_smoke.log | cut -c1-250
done
HB_TRACE_INGEST=1 timeout 300 python tools/ingest_bench.py C3 --out $O/ingest_C3.json > /dev/null 2> $O/ingest_C3.err; echo "ingest C3 rc=$?"; grep "append of" $O/ingest_C3.err | sed -n '2,3p' | cut -c1-250
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04f/ingest_C3.json"))
print({k:d["boundary"][k] for k in ("s_append_edges","s_finalize","append_GBs","records_per_s_library","ms_ingest_reduce","ms_plan","ms_h2d_state")}, d["parity"]["bit_exact"])
PY
HB_TRACE_INGEST=1 HB_TRACE_STORE=1 timeout 600 python bench.py --steps 10 --warmup 3 --c4-leg off --cpu-seconds 5 > $O/bench_C3_e2e.json 2> $O/bench_C3_e2e.err; echo "bench C3 + e2e rc=$?"
grep "hb store\|hb webgraph" $O/bench_C3_e2e.err | cut -c1-330
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04f/bench_C3_e2e.json").read().strip().splitlines()[-1])
    print("value", d["value"], "parity", d["parity_bit_exact"], "input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s","pinned_h2d_GBs")})
    print("e2e", json.dumps(d["detail"].get("end_to_end")))
except Exception as e:
    print("no bench line:", e)
PY
