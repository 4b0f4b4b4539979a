BROTLI_AMD_LIB=$PWD/tools/scratch/lib_profspec.so timeout 300 python tools/prof_workload.py high_entropy 64 2>&1 | grep "spec rounds" | head -1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/spill_report.py 2>&1 | tail -3
for wl in alice29x1024 longbackref_1024x1MiB longbackref_4096x256KiB; do
  timeout 300 python bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['config'].get('second_pass_streams'))"
done
