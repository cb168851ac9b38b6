#!/bin/bash
# round 5, call h: the planner's two cut knobs at C4 on this round's kernels - slice width (tune[3]: log2 counters per slice, default
# 16 = 4 MiB = one XCD's L2) and the minimum number of sources a slice cut needs (tune[4], default 8).  C4's level-1 launch has 92 M
# chunk rows for 1.67 G hub edges (18 sources per chunk on average): fewer, fuller chunks trade L2 hits for partial-row traffic.
set -u
O=gpurun_out/r05h; mkdir -p $O
export HB_SYNTH_CACHE=/dev/shm/hb_synth_cache
timeout 1500 python tools/sweep.py C4 "0:0:" "0:0:0,0,0,0,12" "0:0:0,0,0,0,16" "0:0:0,0,0,0,24" "0:0:0,0,0,0,6" "0:0:0,0,0,15,8" "0:0:0,0,0,17,16" "0:0:0,0,0,15,12" > $O/sweep_C4_cut_knobs.txt 2> $O/sweep.err; echo "rc=$?"
rm -rf /dev/shm/hb_synth_cache
python - <<'PY'
import json
for line in open("gpurun_out/r05h/sweep_C4_cut_knobs.txt"):
    try:
        d=json.loads(line)
        print(d["spec"], "loop", d["ms_loop"], "dense", d["dense_ms_gpu"], "node", d["dense_ms_main"], "vrows", d["virtual_rows"], "L1", d["ms_level1_or_expand"][1:3], "plan", d["ms_plan"], d["same_result"])
    except Exception as e:
        print(line[:200])
PY
