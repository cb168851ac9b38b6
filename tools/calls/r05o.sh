#!/bin/bash
# round 5, call o (call n again after the entry fixes: the collect kernel clears the stale bits, the pipeline hands over): the single-workgroup tail kernel on the MI355X - GPU suite, differential fuzzer (its switches randomised), LT / C3
# lines with the kernel on (default) and off (tune[1] bit 21)
set -u
O=gpurun_out/r05o; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -5 | cut -c1-300
timeout 200 python tools/diff_fuzz.py --mode mixed --seconds 75 --seed 313 > $O/diff_fuzz_mixed_gpu_seed313.json 2> $O/diff_fuzz_a.err; echo "fuzz mixed rc=$?"; cat $O/diff_fuzz_mixed_gpu_seed313.json | cut -c1-300; tail -2 $O/diff_fuzz_a.err | cut -c1-400
timeout 200 python tools/diff_fuzz.py --mode passes --seconds 45 --seed 314 > $O/diff_fuzz_passes_gpu_seed314.json 2> $O/diff_fuzz_b.err; echo "fuzz passes rc=$?"; cat $O/diff_fuzz_passes_gpu_seed314.json | cut -c1-300; tail -2 $O/diff_fuzz_b.err | cut -c1-400
export HB_SYNTH_CACHE=/dev/shm/hb_synth_cache
Q="--cpu-seconds 0 --input dense --c3-leg off --end-to-end off"
for cfg in LT C3; do
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 $Q > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "$cfg rc=$?"
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 $Q --tune 0,2097152 > $O/bench_${cfg}_tail_kernel_off.json 2> $O/bench_${cfg}_b.err; echo "$cfg (bit 21) rc=$?"
done
timeout 600 python bench.py --config LT --steps 3 --warmup 1 --verify --input dense --c3-leg off --end-to-end off > $O/bench_LT_verify.json 2> $O/bench_LT_v.err; echo "LT verify rc=$?"
rm -rf /dev/shm/hb_synth_cache
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05o/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        pp=d["roofline"]["per_pass"]
        print(f.split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], "loop", d["detail"]["ms_loop_per_step"], "gpu", d["detail"]["ms_gpu_passes_per_step"], "parity", d["parity_bit_exact"], "modes", "".join(str(p["mode"]) for p in pp))
        print("    tail ms", [p["ms"] for p in pp if p["mode"] in (2,4)][-8:])
    except Exception as e:
        print(f, "no line", e)
PY
