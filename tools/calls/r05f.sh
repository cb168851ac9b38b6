#!/bin/bash
# round 5, call f: GPU suite (tail pipeline, dest changed-only with one round trip per pass, device-sorted store), LT / C3 lines with
# the pipeline on and off, C3 through the record boundary with the end-to-end chain (hb_store_harmonic_results)
set -u
O=gpurun_out/r05f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
export HB_SYNTH_CACHE=/dev/shm/hb_synth_cache
Q="--cpu-seconds 0 --input dense --c3-leg off --end-to-end off"
for cfg in LT C3; do
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 $Q > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "$cfg rc=$?"
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 $Q --tune 0,1048576 > $O/bench_${cfg}_no_tail_pipeline.json 2> $O/bench_${cfg}_b.err; echo "$cfg (bit 20) rc=$?"
done
unset HB_SYNTH_CACHE
rm -rf /dev/shm/hb_synth_cache
HB_TRACE_STORE=1 timeout 900 python bench.py --config C3 --steps 5 --warmup 2 --c3-leg off > $O/bench_C3_records_e2e.json 2> $O/bench_C3_records_e2e.err; echo "C3 records+e2e rc=$?"
grep "hb store" $O/bench_C3_records_e2e.err | head -12
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05f/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], "loop", d["detail"]["ms_loop_per_step"], "gpu", d["detail"]["ms_gpu_passes_per_step"], "finish", d["detail"]["ms_finish_per_step"], "parity", d["parity_bit_exact"])
        if d["detail"].get("end_to_end"): print("   e2e", json.dumps(d["detail"]["end_to_end"]))
    except Exception as e:
        print(f, "no line", e)
PY
