#!/bin/bash
# round 5, call e: where does the first hb_run of a fresh context lose 0.4 s at C4 (end-to-end leg s_run 0.64 vs 0.21 steady)?
set -u
O=gpurun_out/r05e; mkdir -p $O
export HB_SYNTH_CACHE=/dev/shm/hb_synth_cache
timeout 900 python tools/first_run_probe_big.py C4 > $O/first_run_C4_staged.txt 2>&1; echo "rc=$?"; cat $O/first_run_C4_staged.txt | cut -c1-200
timeout 900 python tools/first_run_probe_big.py C4 0x4000 > $O/first_run_C4_staging_off.txt 2>&1; echo "rc=$?"; grep context $O/first_run_C4_staging_off.txt
rm -rf /dev/shm/hb_synth_cache
