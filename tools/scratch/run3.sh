timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/collect_profiles.sh r01f longbackref_256x4MiB 2>&1 | tail -12
bash tools/collect_profiles.sh r01f_c4 highentropy_256x4MiB 2>&1 | tail -12
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json
timeout 900 python tools/fuzz_campaign.py 1 40 2>&1 | tail -2
