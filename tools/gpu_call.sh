mkdir -p gpurun_out
timeout 70 python bench.py --steps 1 --warmup 0 --c4-leg off --cpu-seconds 4 > gpurun_out/r03x_poison_bench_C3.json 2> gpurun_out/r03x_poison_bench_C3.err; echo "rc=$? C3"; grep -v amdgpu.ids gpurun_out/r03x_poison_bench_C3.err | tail -3 | cut -c1-300
timeout 60 python bench.py --config LT --steps 1 --warmup 0 --c4-leg off --cpu-seconds 6 > gpurun_out/r03x_poison_bench_LT.json 2> gpurun_out/r03x_poison_bench_LT.err; echo "rc=$? LT"; grep -v amdgpu.ids gpurun_out/r03x_poison_bench_LT.err | tail -3 | cut -c1-300
