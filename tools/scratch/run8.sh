export BROTLI_BENCH_UNIQUE=1
for wl in longbackref_1x64MiB highentropy_1x64MiB; do
  timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['roofline']['kernel_ms'], d['config']['workload'])"
done
