#!/bin/bash
# round 6, call f: pass 0 without the src[beg] loads (PassParams::virt_rows): C4 timing; C3 kernel trace + HBM / L2 / SQ counters of every kernel
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 300 python -m pytest tests/test_gpu.py -m gpu -x -q -k "test_per_pass_state_matches_oracle or test_c2" > gpurun_out/r06f_pytest_gpu.log 2>&1; echo "pytest rc=$? after $(( $(date +%s) - T0 )) s"; tail -2 gpurun_out/r06f_pytest_gpu.log | cut -c1-300
timeout 900 python bench.py --config C4 --steps 5 --warmup 2 --cpu-seconds 0 --input dense --c3-leg off --end-to-end off > gpurun_out/r06f_bench_C4.json 2> gpurun_out/r06f_bench_C4.err; echo "rc=$?"
python - <<'PY'
import json
for v in ("C4",):
    try:
        d=json.loads([l for l in open("gpurun_out/r06f_bench_%s.json"%v) if l.startswith("{")][-1])
        print(v, d["value"], "GTEPS", d["ms_per_step"], "ms; first", d.get("first_run_ms"), "plan", d["detail"]["ms_plan"], "state", d["detail"]["ms_h2d"])
        print(" per pass", [(p["t"],p["mode"],p["ms"],p["ms_level1_or_expand"],p["ms_node_rows"]) for p in d["roofline"]["per_pass"]])
    except Exception as e: print(v,"failed",e)
PY
tools/profile.sh C3 r06f > gpurun_out/r06f_profile_C3.log 2>&1; tail -30 gpurun_out/r06f_profile_C3.log | cut -c1-180
echo "total $(( $(date +%s) - T0 )) s"
