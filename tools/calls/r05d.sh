#!/bin/bash
# round 5, call d: GPU suite + quick lines (resident loop only) of C4 / C3 / LT with the streaming cheap-rows kernel, the lazy store of
# the fused dense pass and the side-stream snapshots; A/B of the two switches at C4 (tune[1] bit 18 = cheap rows in the row kernel)
set -u
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
export HB_SYNTH_CACHE=/dev/shm/hb_synth_cache
Q="--cpu-seconds 0 --input dense --c3-leg off --end-to-end off"
for cfg in C4 C3 LT; do
  timeout 600 python bench.py --config $cfg --steps 5 --warmup 2 $Q > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "$cfg rc=$?"
done
timeout 600 python bench.py --config C4 --steps 5 --warmup 2 $Q --tune 0,262144 > $O/bench_C4_cheap_in_row_kernel.json 2> $O/bench_C4_b.err; echo "C4 (bit 18) rc=$?"
timeout 600 python bench.py --config C4 --steps 5 --warmup 2 $Q --tune 0,16384 > $O/bench_C4_staging_off.json 2> $O/bench_C4_c.err; echo "C4 (bit 14) rc=$?"
rm -rf /dev/shm/hb_synth_cache
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05d/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], "loop", d["detail"]["ms_loop_per_step"], "gpu", d["detail"]["ms_gpu_passes_per_step"], "finish", d["detail"]["ms_finish_per_step"])
        print("   ", [(p["t"], p["mode"], p["ms"], p["ms_level1_or_expand"], p["ms_node_rows"]) for p in d["roofline"]["per_pass"]][:12])
    except Exception as e:
        print(f, "no line", e)
PY
