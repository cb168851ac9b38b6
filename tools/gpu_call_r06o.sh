#!/bin/bash
# round 6, call o: rocprofv3 kernel trace + ALL counter groups (HBM bytes, L2 hit / miss, request sizes, SQ issue / wait breakdown) of the final tree
# at C4 with the export that leaves the load's zero-row warm-up launches out of the class averages
set -u
mkdir -p gpurun_out
T0=$(date +%s)
tools/profile.sh C4 r06o > gpurun_out/r06o_profile_C4.log 2>&1; tail -60 gpurun_out/r06o_profile_C4.log | cut -c1-210 | head -40
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r06o_C4_pmc.json"))
    for k,v in d.items():
        if "SQ_WAVES" in v and v["SQ_WAVES"]["max"] > 1000:
            w=v["SQ_WAVES"]["max"]; 
            print(k[:64], "waves", int(w), "VALU insts/wave", round(v["SQ_INSTS_VALU"]["max"]/w,1), "busy", {c:round(v[c]["max"]/v["SQ_WAVE_CYCLES"]["max"],3) for c in ("SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_INST_CYCLES_VMEM") if c in v})
except Exception as e: print("failed", e)
PY
echo "total $(( $(date +%s) - T0 )) s"
