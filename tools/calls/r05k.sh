#!/bin/bash
# round 5, call k: the final tree at the largest BASELINE config - C5 (312 M hosts / 5.3 B edges, 6.36 G raw records through the record
# boundary on ONE GPU), oracle run to convergence, final list compared - and the differential fuzzer on the MI355X (passes + record
# boundary + reference tail + ranks), both on this round's kernels (staged result download at n = 312 M, lazy store in the dense pass,
# tail pipeline, wave-uniform table insert)
set -u
O=gpurun_out/r05k; mkdir -p $O
timeout 120 python tools/diff_fuzz.py --mode mixed --seconds 60 --seed 51 > $O/diff_fuzz_mixed_gpu.json 2> $O/diff_fuzz.err; echo "fuzz rc=$?"; cat $O/diff_fuzz_mixed_gpu.json | cut -c1-300
timeout 2400 python bench.py --config C5 --input records --verify --steps 2 --warmup 1 --end-to-end off --c3-leg off > $O/bench_C5_records_verify.json 2> $O/bench_C5.err; echo "bench C5 rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r05k/bench_C5_records_verify.json").read().strip().splitlines()[-1])
    print("C5 value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"], "roof", d["roofline"]["frac"], "input", {k:d["detail"]["input"].get(k) for k in ("s_append_edges","s_finalize","records_per_s","ingest_peak_bytes_per_record")})
    print("per pass", [(p["t"], p["mode"], p["ms"]) for p in d["roofline"]["per_pass"]], "finish", d["detail"]["ms_finish_per_step"])
except Exception as e:
    print("no bench line:", e); print(open("gpurun_out/r05k/bench_C5.err").read()[-2000:])
PY
