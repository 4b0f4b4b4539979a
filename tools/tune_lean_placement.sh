#!/bin/bash
# The lean loop's speed depends on where its loops lie relative to the 32-byte instruction-fetch lines (C3: up to 6 %
# between the eight 4-byte placements), so BROTLI_AMD_LEAN_PAD_NEVER / _CTX in brotli_kernels.hip are measured, not
# chosen, and have to be measured again after every edit of lean_commands.
#   tools/tune_lean_placement.sh build     (here: eight libraries under tools/scratch/, ~12 min)
#   gpurun -- 'bash tools/tune_lean_placement.sh run'   (on the GPU box: C3 and alice29 x 1024 with each)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
PKG=$REPO/rust-brotli-decompressor_amd
mkdir -p "$REPO/tools/scratch"
if [ "${1:-}" = build ]; then
  for pad in 0 1 2 3 4 5 6 7; do
    touch "$PKG/csrc/brotli_kernels.hip"
    make -s -C "$PKG" EXTRA="-DBROTLI_AMD_LEAN_PAD_NEVER=$pad -DBROTLI_AMD_LEAN_PAD_CTX=$pad" 2>&1 | grep -i error
    cp "$PKG/libbrotli_decompressor.so" "$REPO/tools/scratch/lib_pad$pad.so"
  done
  touch "$PKG/csrc/brotli_kernels.hip"; make -s -C "$PKG" 2>&1 | grep -i error
  echo "built: tools/scratch/lib_pad[0-7].so (the in-tree library is back at the defaults)"
else
  for pad in 0 1 2 3 4 5 6 7; do
    for wl in longbackref_256x4MiB alice29x1024; do
      BROTLI_AMD_LIB=$REPO/tools/scratch/lib_pad$pad.so timeout 300 python "$REPO/bench.py" --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pad $pad $wl %.1f MB/s' % d['value'])"
    done
  done
fi
