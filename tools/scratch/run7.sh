timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 600 python tools/fuzz_big.py 31 2>&1 | tail -1
for wl in highentropy_256x4MiB highentropy_1024x1MiB longbackref_256x4MiB; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['roofline']['kernel_ms'])"
done
