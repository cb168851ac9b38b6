set -x
mkdir -p gpurun_out
( time python bench.py --config C4 --verify --steps 2 --warmup 1 --c4-leg off --pass-log gpurun_out/r03s_passes_C4.json ) > gpurun_out/r03s_bench_C4_records_verify.json 2> gpurun_out/r03s_bench_C4.err; tail -5 gpurun_out/r03s_bench_C4.err; cut -c1-800 gpurun_out/r03s_bench_C4_records_verify.json
