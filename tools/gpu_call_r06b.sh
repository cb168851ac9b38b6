#!/bin/bash
# round 6, call b: where the first run of a fresh process spends its time; lean pass 0 vs the full init kernel (A/B); CRC forms on this host
set -u
mkdir -p gpurun_out
T0=$(date +%s)
HB_TRACE_RESULTS=1 HB_TRACE_INIT=1 timeout 300 python tools/first_run_probe.py C3 > gpurun_out/r06b_first_run_probe_C3.json 2> gpurun_out/r06b_first_run_probe_C3.err; echo "probe rc=$?"; cut -c1-900 gpurun_out/r06b_first_run_probe_C3.json; head -40 gpurun_out/r06b_first_run_probe_C3.err
timeout 120 python tools/crc_bench.py 4 > gpurun_out/r06b_crc_bench.json 2>&1; cat gpurun_out/r06b_crc_bench.json
for V in lean fullinit; do
  T=""; [ $V = fullinit ] && T="--tune 0,8388608"
  timeout 600 python bench.py --config C4 --steps 5 --warmup 2 --input dense --cpu-seconds 0 --end-to-end off --c3-leg off $T > gpurun_out/r06b_bench_C4_$V.json 2> gpurun_out/r06b_bench_C4_$V.err; echo "C4 $V rc=$?"
  python - $V <<'PY'
import json,sys
v=sys.argv[1]
try:
    d=json.loads([l for l in open("gpurun_out/r06b_bench_C4_%s.json"%v) if l.startswith("{")][-1])
    print(v, d["value"], "GTEPS", d["ms_per_step"], "ms first", d.get("first_run_ms"), "finish", d["detail"]["ms_finish_per_step"], "loop", d["detail"]["ms_loop_per_step"], "gpu", d["detail"]["ms_gpu_passes_per_step"])
    print(" per pass", [(p["t"],p["mode"],p["ms"],p["ms_level1_or_expand"],p["ms_node_rows"]) for p in d["roofline"]["per_pass"]][:3])
except Exception as e: print(v,"failed",e)
PY
done
for V in lean fullinit; do
  T=""; [ $V = fullinit ] && T="--tune 0,8388608"
  timeout 300 python bench.py --config C3 --steps 10 --warmup 3 --input dense --cpu-seconds 0 --end-to-end off --c3-leg off $T > gpurun_out/r06b_bench_C3_$V.json 2> gpurun_out/r06b_bench_C3_$V.err; echo "C3 $V rc=$?"
  python - $V <<'PY'
import json,sys
v=sys.argv[1]
try:
    d=json.loads([l for l in open("gpurun_out/r06b_bench_C3_%s.json"%v) if l.startswith("{")][-1])
    print(v, d["value"], "GTEPS", d["ms_per_step"], "ms first", d.get("first_run_ms"), "finish", d["detail"]["ms_finish_per_step"], "loop", d["detail"]["ms_loop_per_step"], "gpu", d["detail"]["ms_gpu_passes_per_step"])
    print(" per pass", [(p["t"],p["mode"],p["ms"],p["ms_level1_or_expand"],p["ms_node_rows"]) for p in d["roofline"]["per_pass"]][:3])
except Exception as e: print(v,"failed",e)
PY
done
echo "total $(( $(date +%s) - T0 )) s"
