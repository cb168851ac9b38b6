#!/bin/bash
# round 6, call u: a longer fuzz campaign of the committed tree on the MI355X (every mode, fresh seed)
set -u
mkdir -p gpurun_out
T0=$(date +%s)
for MODE in passes records mixed ranks tail; do
  HB_LIB_PATH=stract_amd/lib/libhyperball_exp.so timeout 400 python tools/diff_fuzz.py --mode $MODE --seconds 100 --seed 6099 > gpurun_out/r06u_diff_fuzz_$MODE.txt 2>&1; echo "fuzz $MODE rc=$?"; tail -1 gpurun_out/r06u_diff_fuzz_$MODE.txt | cut -c1-300
done
echo "total $(( $(date +%s) - T0 )) s"
