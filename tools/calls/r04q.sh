#!/bin/bash
# round 4, call q (the last GPU minute): differential fuzzing of the pass driver against the oracle ON THE MI355X - random graphs x random
# layout / mode knobs, compared after every pass (tools/diff_fuzz.py; the same tool runs on the interpreted device sources without a GPU)
set -u
O=gpurun_out/r04q; mkdir -p $O
timeout 58 python tools/diff_fuzz.py --seconds 42 --seed 5 > $O/diff_fuzz_gpu.json 2> $O/diff_fuzz_gpu.err; echo "fuzz rc=$?"
cat $O/diff_fuzz_gpu.json; tail -3 $O/diff_fuzz_gpu.err
