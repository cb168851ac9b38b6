#!/bin/bash
# round 5, call i: tune[4] (minimum sources per slice cut) finer at C4 and at C3 / LT - is 12 better than 8 everywhere?
set -u
O=gpurun_out/r05i; mkdir -p $O
export HB_SYNTH_CACHE=/dev/shm/hb_synth_cache
timeout 900 python tools/sweep.py C3 "0:0:" "0:0:0,0,0,0,10" "0:0:0,0,0,0,12" "0:0:0,0,0,0,16" "0:0:" > $O/sweep_C3_minc.txt 2> $O/sweep_C3.err; echo "C3 rc=$?"
timeout 900 python tools/sweep.py LT "0:0:" "0:0:0,0,0,0,12" > $O/sweep_LT_minc.txt 2> $O/sweep_LT.err; echo "LT rc=$?"
timeout 1500 python tools/sweep.py C4 "0:0:" "0:0:0,0,0,0,10" "0:0:0,0,0,0,12" "0:0:0,0,0,0,14" "0:0:" > $O/sweep_C4_minc.txt 2> $O/sweep_C4.err; echo "C4 rc=$?"
rm -rf /dev/shm/hb_synth_cache
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05i/sweep_*_minc.txt")):
    print(f)
    for line in open(f):
        try:
            d=json.loads(line)
            print("  ", d["spec"], "loop", d["ms_loop"], "dense", d["dense_ms_gpu"], "node", d["dense_ms_main"], "front", d["front_ms_sum"], "sparse", d["sparse_ms_sum"], "vrows", d["virtual_rows"], d["same_result"])
        except Exception as e:
            print(line[:200])
PY
