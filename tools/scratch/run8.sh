for f in reducetostream.map.compressed metablock_reset.compressed plrabn12.txt.compressed plrabn12.txt.bro lcet10.txt.compressed random_then_unicode.compressed mapsdatazrh.compressed zeros.compressed quickfox_repeated.compressed compressed_repeated.compressed alice29.txt.bro asyoulik.txt.compressed; do
  timeout 300 python bench.py --workload "fixture:${f}x1024" --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['roofline']['kernel_ms'], d['config'].get('second_pass_streams'))" 2>&1 | tail -1
done
