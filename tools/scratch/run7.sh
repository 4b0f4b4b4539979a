timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
timeout 600 python tools/fuzz_big.py 41 2>&1 | tail -1
export BROTLI_BENCH_UNIQUE=1
for wl in highentropy_1x4MiB highentropy_1x16MiB longbackref_1x4MiB longbackref_1x16MiB; do
  timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['roofline']['kernel_ms'])"
done
unset BROTLI_BENCH_UNIQUE
for wl in highentropy_256x4MiB longbackref_256x4MiB; do
  timeout 600 python bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['roofline']['kernel_ms'])"
done
