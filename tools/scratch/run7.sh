timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 600 python tools/fuzz_big.py 15 2>&1 | tail -1
for lib in rust-brotli-decompressor_amd/libbrotli_decompressor.so tools/scratch/lib_0e3a0ec.so; do
for wl in longbackref_256x4MiB alice29x1024; do
  BROTLI_AMD_LIB=$PWD/$lib timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $wl', d['value'], d['roofline']['kernel_ms'])"
done; done
