#!/bin/bash
# round 4, call h: where does hb_finalize spend its time at C4?  C5 (312 M hosts / 5.3 B edges, 6.4 G raw records) through the record boundary, verified
set -u
O=gpurun_out/r04h; mkdir -p $O
HB_TRACE_INGEST=1 HB_PLAN_TIMING=1 timeout 900 python tools/ingest_bench.py C4 --out $O/ingest_C4.json > /dev/null 2> $O/ingest_C4.err; echo "ingest C4 rc=$?"
grep "hb finalize\|hb ingest\] [a-z]\|gpu plan" $O/ingest_C4.err | cut -c1-200 | tail -45
grep "append of" $O/ingest_C4.err | tail -2 | cut -c1-330
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04h/ingest_C4.json"))
print({k:d["boundary"][k] for k in ("s_append_edges","s_finalize","append_GBs","records_per_s_library","ms_ingest_reduce","ms_plan","ms_h2d_state","peak_bytes_per_record")}, d["parity"])
PY
HB_TRACE_INGEST=1 timeout 2400 python bench.py --config C5 --input records --verify --steps 1 --warmup 0 --end-to-end off --c4-leg off > $O/bench_C5_records.json 2> $O/bench_C5_records.err; echo "bench C5 records rc=$?"
grep "hb finalize\|hb ingest\] [a-z]" $O/bench_C5_records.err | cut -c1-200 | tail -14
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04h/bench_C5_records.json").read().strip().splitlines()[-1])
    print("C5 value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"], "input", d["detail"]["input"])
except Exception as e:
    print("no bench line:", e); print(open("gpurun_out/r04h/bench_C5_records.err").read()[-1500:])
PY
