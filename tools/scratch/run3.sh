timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/collect_profiles.sh r01e longbackref_256x4MiB 2>&1 | tail -20
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json
