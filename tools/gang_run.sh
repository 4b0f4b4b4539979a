#!/bin/bash
# The gang legs' measurements on the GPU box: tools/gang_run.sh <tag>  ->  gpurun_out/gang_<tag>/...
# (rocprofv3 summaries and PMC passes of one stream of 1 GiB / 64 MiB and of 32 x 4 MiB, the gangs' own counters from a
# -DBROTLI_AMD_GANG_STATS build if tools/scratch/lib_gangstats.so is there, gangs against one block a stream on the same box)
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
O=gpurun_out/gang_$TAG
mkdir -p $O
for WL in longbackref_1x1024MiB longbackref_1x64MiB longbackref_32x4MiB longbackrefmix_200; do
  timeout 900 tools/collect_profiles.sh ${TAG}_$WL $WL > $O/collect_$WL.log 2>&1
  cd "$REPO"
done
if [ -f tools/scratch/lib_gangstats.so ]; then
  for WL in longbackref_1x4MiB longbackref_1x64MiB longbackref_1x1024MiB surveymix_1x4MiB; do
    echo "== $WL (the first stream's gang: counters of a -DBROTLI_AMD_GANG_STATS build, bench.py --workload $WL --steps 1 --warmup 0)"
    BROTLI_AMD_GANG_STATS=1 BROTLI_AMD_LIB=tools/scratch/lib_gangstats.so timeout 300 python bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-extra 2>&1 | grep "^gang of"
  done > $O/gang_stats.txt 2>&1
fi
{
  for WL in longbackref_1x4MiB longbackref_1x64MiB longbackref_1x1024MiB longbackref_8x4MiB longbackref_32x4MiB longbackref_64x4MiB longbackref_128x4MiB surveymix_1x4MiB surveymix_8x4MiB longbackrefq9_8x4MiB highentropy_8x4MiB recompressed:lcet10.txt.compressedq5x8 alice29x8; do
    for G in 8 0; do
      BROTLI_AMD_GANG=$G timeout 300 python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL', 'BROTLI_AMD_GANG=$G', 'blocks a stream', d['config']['blocks_per_stream'], d['value'], 'MB/s', d['ms_per_step'], 'ms', 'engine share', d['engine_commands_share'])"
    done
  done
  for WL in longbackrefmix_200 longbackrefmix_256; do   # (a pool of blocks: one 64 MiB stream among 199 / 255 of 1 MiB)
    for P in 1 0; do
      BROTLI_AMD_POOL=$P timeout 300 python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL', 'BROTLI_AMD_POOL=$P', 'a pool' if d['config']['pool_of_blocks'] else 'no pool', d['value'], 'MB/s', d['ms_per_step'], 'ms', 'engine share', d['engine_commands_share'])"
    done
  done
} > $O/gang_ab.txt 2>&1
cat $O/gang_ab.txt
