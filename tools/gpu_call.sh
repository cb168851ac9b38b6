mkdir -p gpurun_out
timeout 80 python bench.py --steps 3 --warmup 1 --c4-leg off --cpu-seconds 2 > gpurun_out/r03y_bench_C3_supervised.json 2> gpurun_out/r03y_bench_C3.err; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r03y_bench_C3.err | tail -3 | cut -c1-200; cut -c1-160 gpurun_out/r03y_bench_C3_supervised.json; wc -l gpurun_out/r03y_bench_C3_supervised.json
