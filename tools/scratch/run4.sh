for wl in alice29x1024 longbackref_1024x1MiB longbackref_2048x512KiB longbackref_4096x256KiB longbackref_16384x64KiB highentropy_4096x256KiB highentropy_256x4MiB; do
  timeout 300 python bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['config'].get('second_pass_streams'))"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
