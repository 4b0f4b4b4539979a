timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 600 python tools/fuzz_big.py 11 2>&1 | tail -2
for wl in longbackref_256x4MiB highentropy_256x4MiB alice29x1024; do timeout 300 python bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:50], d['value'])"; done
