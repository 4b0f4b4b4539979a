timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for wl in longbackref_256x4MiB alice29x1024 highentropy_256x4MiB; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['roofline']['kernel_ms'])"
done
