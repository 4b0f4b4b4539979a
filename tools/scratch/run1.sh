timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for wl in longbackref_256x4MiB highentropy_256x4MiB; do timeout 300 python bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'])"; done
