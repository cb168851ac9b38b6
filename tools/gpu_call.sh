set -x
mkdir -p gpurun_out
( time python bench.py --config C5 --verify --input dense --steps 1 --warmup 0 --c4-leg off --tune 0,0,101,0,0,0,1000000 --pass-log gpurun_out/r03q_passes_C5_frontier_always.json ) > gpurun_out/r03q_bench_C5_frontier_always_verify.json 2> gpurun_out/r03q_bench_C5.err; tail -5 gpurun_out/r03q_bench_C5.err; cut -c1-900 gpurun_out/r03q_bench_C5_frontier_always_verify.json
