for p in 0 1 2 3 4 5 6 7; do lib=tools/scratch/lib_pad$p.so
 for wl in longbackref_256x4MiB alice29x1024; do
  BROTLI_AMD_LIB=$PWD/$lib timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pad$p $wl', d['value'], d['roofline']['kernel_ms'])"
 done
done
