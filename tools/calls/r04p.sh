#!/bin/bash
# round 4, call p (last GPU seconds of the round): the default line's C3 half on the final tree - record input, end-to-end chain with
# the harness writing its edge store with 4 segments in the making at once - and smoke()
set -u
O=gpurun_out/r04p; mkdir -p $O
timeout 35 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 95 python bench.py --c4-leg off --steps 20 --warmup 5 > $O/bench_C3.json 2> $O/bench_C3.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04p/bench_C3.json").read().strip().splitlines()[-1])
    print("C3 value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity_bit_exact"], "roof", d["roofline"]["frac"])
    print("input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s")})
    print("e2e", json.dumps(d["detail"].get("end_to_end")))
except Exception as e:
    print("no bench line:", e); print(open("gpurun_out/r04p/bench_C3.err").read()[-1500:])
PY
