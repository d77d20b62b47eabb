set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -2 gpurun_out/bench_final.err
cp gpurun_out/bench_full.json gpurun_out/bench_full_final.json
