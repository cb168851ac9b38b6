#!/bin/bash
# round 4, call n: parity lines of the final tree on the other configs (C2 = BASELINE configs[1], LT = long tail), verified against the oracle run to
# convergence, through the record boundary; more record-input rounds
set -u
O=gpurun_out/r04n; mkdir -p $O
timeout 300 python bench.py --config C2 --verify --steps 5 --warmup 2 --c4-leg off --end-to-end on > $O/bench_C2_verify.json 2> $O/bench_C2_verify.err; echo "bench C2 rc=$?"
timeout 400 python bench.py --config LT --verify --steps 5 --warmup 2 --c4-leg off > $O/bench_LT_verify.json 2> $O/bench_LT_verify.err; echo "bench LT rc=$?"
python - <<'PY'
import json
for c in ("C2","LT"):
    try:
        d=json.loads(open("gpurun_out/r04n/bench_%s_verify.json"%c).read().strip().splitlines()[-1])
        e=d["detail"].get("end_to_end") or {}
        print(c, "value", d["value"], "ms/step", d["ms_per_step"], "T", d["config"]["passes_T"], "parity", d["parity"]["bit_exact"], d["parity"]["scope"][:70], "records/s", d["detail"]["input"].get("records_per_s"), "e2e", {k:e.get(k) for k in ("s_total","graph_ok","same_result_as_record_leg","stores_read_back_ok")})
    except Exception as ex:
        print(c, "no line:", ex)
PY
timeout 200 python tools/record_stress.py C3,LT --rounds 5 --tag shipped --out $O/stress_shipped.json > /dev/null 2> $O/stress_shipped.err; echo "stress shipped rc=$?"
